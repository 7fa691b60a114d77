"""The host-buffer batch path alone (revo_track_pairs_submit / _wait, 32 pairs, three jobs in flight): frames/s and PCIe
rate for u16 and f32 depth.  REVO_H2D_CHUNK=<bytes> splits every plane copy (experiment)."""
import sys
import time
from collections import deque
import numpy as np
import torch
sys.path.insert(0, ".")
from revo_amd import api, synth
from revo_amd.settings import ImgPyramidSettings, TrackerSettings
s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
cam = api.CameraPyr(s)
api.TrackerNew(TrackerSettings(), s, cam)
n = 32
base = [synth.make_pair(i, s) for i in range(4)]
rendered = [base[i % 4] for i in range(n)]
for tag, scale in (("u16", 5000.0), ("f32", None)):
    fr = []
    for r in rendered:
        one = []
        for key in ("ref", "curr"):
            d = np.clip(r[key][1] * 5000.0, 0, 65535).astype(np.uint16) if scale else r[key][1]
            one.append((torch.from_numpy(np.ascontiguousarray(r[key][0])).pin_memory().numpy(),
                        torch.from_numpy(np.ascontiguousarray(d)).pin_memory().numpy()))
        fr.append(tuple(one))
    hb = api.HostBatchTracker(cam, depth_scale_factor=scale)
    for j in [hb.submit(fr) for _ in range(3)]:
        hb.wait(j)
    jobs, steps = deque(), 30
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        jobs.append(hb.submit(fr))
        if len(jobs) == 3:
            hb.wait(jobs.popleft())
    while jobs:
        hb.wait(jobs.popleft())
    dt = time.perf_counter() - t0
    nbytes = n * 2 * 640 * 480 * (3 + (2 if scale else 4))
    print("%s: %.0f frames/s, %.2f ms per job, %.1f GB/s" % (tag, n * steps / dt, dt / steps * 1e3, nbytes * steps / dt / 1e9))
    del hb
