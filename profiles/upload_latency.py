"""Latency of one in-place frame upload + build enqueue (revo_pyramid_create with rows in page-locked memory returns when the
device-side clone exists): copy KERNEL (REVO_H2D_KERNEL=1) against hipMemcpyAsync (=0), 640x480, BGR8 + f32 depth = 2.15 MB.
  python profiles/upload_latency.py [calls=400]            (GPU box; the knobs are environment variables)"""
import os
import sys
import time

import numpy as np
import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from revo_amd import api, synth  # noqa: E402
from revo_amd.settings import ImgPyramidSettings  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 400
s = ImgPyramidSettings(pyr_min_lvl=3)
bgr, dep = synth.make_sequence(3, s, 1, max_t=0.01, max_rot_deg=0.4)[0][:2]
pb = torch.from_numpy(np.ascontiguousarray(bgr)).pin_memory().numpy()
pd = torch.from_numpy(np.ascontiguousarray(dep, np.float32)).pin_memory().numpy()
cam = api.CameraPyr(s)
keep = [api.ImgPyramidRGBD(s, cam, pb, pd, 0.0) for _ in range(8)]
del keep
torch.cuda.synchronize()
t = []
for i in range(n):
    t0 = time.perf_counter()
    p = api.ImgPyramidRGBD(s, cam, pb, pd, float(i))
    t.append(time.perf_counter() - t0)
    del p
    time.sleep(0.0003)  # (let the build drain: the call's own latency is what is measured)
t = np.array(t) * 1e6
print("REVO_H2D_KERNEL=%s REVO_UPLOAD_BLOCKS=%s: create call median %.1f us, 10 %% %.1f, 90 %% %.1f, max %.1f  (%.1f GB/s if it were all copy)"
      % (os.environ.get("REVO_H2D_KERNEL", "1"), os.environ.get("REVO_UPLOAD_BLOCKS", "256"), np.median(t), np.percentile(t, 10),
         np.percentile(t, 90), t.max(), (pb.nbytes + pd.nbytes) / np.median(t) / 1e3))
