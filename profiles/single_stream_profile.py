"""Where a frame of the sequential stream (BASELINE configs[1]) spends its time.
Usage: python profiles/single_stream_profile.py [n_frames] [levels]   (GPU box)
Prints host wall-clock per call (submit = host copy + enqueue of the async build; track_next =
trackFrames + vote + bookkeeping) -- run it under `rocprofv3 --kernel-trace --stats` for the
device side."""
import sys
import time

import numpy as np

sys.path.insert(0, ".")
from revo_amd import api, synth, vo  # noqa: E402
from revo_amd.settings import ImgPyramidSettings  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 60
levels = int(sys.argv[2]) if len(sys.argv) > 2 else 4
s = ImgPyramidSettings(pyr_min_lvl=levels - 1)
frames = synth.make_sequence(7, s, n, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(1.0), 0])
cam = api.CameraPyr(s)
vo.REVO(s, cameraPyr=cam).run([f[:3] for f in frames[:6]])  # warm-up
for mode in ("io_thread", "lookahead", "push"):
    drv = vo.REVO(s, cameraPyr=cam)
    t_sub, t_trk = [], []
    t0 = time.perf_counter()
    if mode == "io_thread":
        drv.run([f[:3] for f in frames], io_thread=True)
        t_sub, t_trk = [0.0], [0.0]
    elif mode == "lookahead":
        a = time.perf_counter(); drv.submit(*frames[0][:3]); t_sub.append(time.perf_counter() - a)
        for f in frames[1:]:
            a = time.perf_counter(); drv.submit(*f[:3]); b = time.perf_counter(); drv.track_next(); c = time.perf_counter()
            t_sub.append(b - a); t_trk.append(c - b)
        a = time.perf_counter(); drv.track_next(); t_trk.append(time.perf_counter() - a)
    else:
        for f in frames:
            a = time.perf_counter(); drv.submit(*f[:3]); b = time.perf_counter(); drv.track_next(); c = time.perf_counter()
            t_sub.append(b - a); t_trk.append(c - b)
    dt = time.perf_counter() - t0
    est = [p[1] for p in drv.poses]
    print("%s: %.0f frames/s, %d keyframes; submit %.3f ms (median %.3f), track_next %.3f ms (median %.3f, max %.3f); ATE vs GT %.4f m"
          % (mode, n / dt, drv.nKeyFrames, 1e3 * np.mean(t_sub), 1e3 * np.median(t_sub), 1e3 * np.mean(t_trk),
             1e3 * np.median(t_trk), 1e3 * np.max(t_trk), synth.ate_rmse(est, [f[3] for f in frames])))
