"""Diagnostic: where do the resident gate's census counter and the host's count of enqueued tracker workgroups part?
usage: python profiles/census_probe.py [pinned|pageable] [thread|nothread]"""
import ctypes as C
import sys
import numpy as np
sys.path.insert(0, ".")
import torch
from revo_amd import api, synth, vo, _lib
from revo_amd.settings import ImgPyramidSettings, TrackerSettings

pinned = "pinned" in sys.argv
thread = "nothread" not in sys.argv
L = _lib.lib()
L.revo_debug_census_.argtypes = [C.c_int, C.POINTER(C.c_uint)]


def census(tag):
    torch.cuda.synchronize()
    o = (C.c_uint * 3)()
    rc = L.revo_debug_census_(0, o)
    print("%-40s rc %d census %d timeouts %d enqueued %d  diff %d" % (tag, rc, o[0], o[1], o[2], o[2] - o[0]), flush=True)


s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
cam = api.CameraPyr(s, device=0)
api.TrackerNew(TrackerSettings(), s, cam)
seq = synth.make_sequence(7, s, 20, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(1.0), 0])


def pin(x):
    return torch.from_numpy(np.ascontiguousarray(x)).pin_memory().numpy() if pinned else x


frames = [(pin(f[0]), pin(f[1]), f[2]) for f in seq]
for r in range(3):
    d = vo.REVO(s, cameraPyr=cam)
    d.run(frames, io_thread=thread)
    census("after VO run %d (pinned=%s thread=%s) kf %d" % (r, pinned, thread, d.nKeyFrames))
    del d
n = 8
pairs = [synth.make_pair(i, s) for i in range(n)]
bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).cuda()
pipe = api.Pipeline(cam, n)
print(pipe.info())
census("after pipeline create")
out = torch.zeros(n * 96, dtype=torch.uint8, device="cuda")
import time
for k in range(3):
    t0 = time.perf_counter()
    for _ in range(4):
        pipe.submit(bgr.data_ptr(), dep.data_ptr(), out.data_ptr())
    pipe.drain()
    census("after 4 pipelined steps (%.1f ms)" % ((time.perf_counter() - t0) * 1e3))
