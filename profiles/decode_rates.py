"""Decode rates of tum.DecodePool on this host, without the GPU path: one decoder in this process against pools of several
decoder processes, with the ring pageable and page-locked.  (profiles/r06_call7.sh)"""
import os
import sys
import time

sys.path.insert(0, ".")
import numpy as np

from revo_amd import synth, tum
from revo_amd.settings import ImgPyramidSettings

if __name__ == "__main__":
    s3 = ImgPyramidSettings()
    d = "/tmp/tumsynth_%d" % os.getpid()
    seq = synth.make_sequence(5, s3, 64, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(0.5), 0], workers=8)
    tum.write_synthetic_dataset(d, seq)
    rows = tum.read_associate(d + "/associate.txt")
    print("cpus: affinity %d, cgroup cpu.max %s" % (len(os.sched_getaffinity(0)), open("/sys/fs/cgroup/cpu.max").read().strip() if os.path.exists("/sys/fs/cgroup/cpu.max") else "?"))
    t0 = time.perf_counter()
    for r in rows[:32]:
        tum.load_frame(d, r[1], r[3])
    print("one decoder in this process: %.0f frames/s" % (32 / (time.perf_counter() - t0)))
    for pin in (False, True):
        for w in (2, 4, 8, 12, 14, 16):
            with tum.DecodePool(d, rows * 4, 640, 480, workers=w, pin=pin) as pool:
                pool.warm()
                t0 = time.perf_counter()
                k = sum(1 for _ in pool)
                print("pool of %2d decoders, ring %s: %.0f frames/s" % (w, "page-locked" if pool.pinned else "pageable", k / (time.perf_counter() - t0)))
