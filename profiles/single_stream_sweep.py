"""Single sequential stream (the bench's 60-frame seeded sweep) under tracker knobs: best-of-3 frames/s,
keyframes and ATE per setting.  Usage: python profiles/single_stream_sweep.py [KSPEC:CLUSTER_ONE:REDUNDANT_ONE ...]"""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from revo_amd import api, synth, vo
from revo_amd.settings import ImgPyramidSettings

s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
n = 60
seq = synth.make_sequence(7, s, n, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(1.0), 0])
frames = [(f[0], f[1], f[2]) for f in seq]
configs = sys.argv[1:] or ["4:16:1024"]
for cfg in configs:
    k, cl, red = cfg.split(":")
    os.environ["REVO_TRACK_KSPEC"] = k
    os.environ["REVO_TRACK_CLUSTER_ONE"] = cl
    os.environ["REVO_TRACK_REDUNDANT_ONE"] = red
    cam = api.CameraPyr(s)
    vo.REVO(s, cameraPyr=cam).run(frames[:6])
    runs = []
    for _ in range(3):
        drv = vo.REVO(s, cameraPyr=cam)
        t0 = time.perf_counter()
        drv.run(frames)
        runs.append(n / (time.perf_counter() - t0))
    ate = synth.ate_rmse([p[1] for p in drv.poses], [f[3] for f in seq])
    print("kspec %-5s cluster %2s redundant<= %5s : %7.0f frames/s (runs %s)  keyframes %d  ATE %.3f mm"
          % (k, cl, red, max(runs), " ".join("%.0f" % r for r in runs), drv.nKeyFrames, ate * 1e3), flush=True)
    del drv, cam
