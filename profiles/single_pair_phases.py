"""Phase split of k_track for ONE pair (the sequential path): run with the normal library for the
evaluation counts, and with a -DREVO_TRACK_PROFILE build (evals[] then carries phase cycles / 16)."""
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from revo_amd import api, synth
from revo_amd.settings import ImgPyramidSettings, TrackerSettings
s = ImgPyramidSettings(pyr_min_lvl=3)
fr = synth.make_sequence(7, s, 3, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(1.0), 0])
cam = api.CameraPyr(s)
trk = api.TrackerNew(TrackerSettings(), s, cam)
a = api.ImgPyramidRGBD(s, cam, fr[0][0], fr[0][1]); a.makeKeyframe()
b = api.ImgPyramidRGBD(s, cam, fr[1][0], fr[1][1])
for _ in range(3):
    trk.trackFrames(np.eye(3), np.zeros(3), a, b)
t0 = time.perf_counter()
for _ in range(20):
    trk.trackFrames(np.eye(3), np.zeros(3), a, b)
dt = (time.perf_counter() - t0) / 20
ev = list(trk.last_evals)
print("trackFrames wall %.1f us; evals/phases %s" % (dt * 1e6, ev))
