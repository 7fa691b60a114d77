"""Phase split of k_track for ONE pair (the sequential path).  Run with a -DREVO_TRACK_PROFILE build of the
library (REVO_HIP_SO=...): the record's evals[] then carry cycles/16 of: [0] evaluation loop, [1] barrier 1,
[2] LDS sum + cluster exchange, [3] barrier 2, [4] decision (solve, exp, candidates), and [5] the number of passes."""
import os
import sys
import time
import numpy as np
sys.path.insert(0, ".")
from revo_amd import api, synth
from revo_amd.settings import ImgPyramidSettings, TrackerSettings
s = ImgPyramidSettings(pyr_min_lvl=3)
s.hist_patch[3] = 0
fr = synth.make_sequence(7, s, 3, max_t=0.01, max_rot_deg=0.4, bias=[0.004, 0, 0, 0, np.deg2rad(1.0), 0])
KS = [int(x) for x in os.environ.get('PH_KSPEC', '1,2244').split(',')]
CLS = [int(x) for x in os.environ.get('PH_CLUSTER', '8,16').split(',')]
for kspec in KS:
    for cl in CLS:
        os.environ["REVO_TRACK_KSPEC"] = str(kspec)
        os.environ["REVO_TRACK_CLUSTER_ONE"] = str(cl)
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
        ev = [int(x) for x in trk.last_evals]
        if os.environ.get("REVO_HIP_SO"):
            import ctypes as C
            from revo_amd import _lib
            buf = (C.c_float * 13)()
            _lib.lib().revo_debug_track_profile_(cam._h, buf)
            pr = [x / 2400.0 for x in buf]
            npass = max(1.0, buf[12]); ngen = max(1.0, buf[10])
            print("    per pass us: eval[pose+F %.2f | E %.2f | reduce %.2f]  B1 %.2f  sum+exch %.2f  decision %.2f [logic %.2f | solve %.2f | exp+write %.2f per gen pass, %d gen passes]  B2 %.2f"
                  % (pr[6] / npass, pr[7] / npass, (pr[0] - pr[6] - pr[7]) / npass, pr[1] / npass, pr[2] / npass, pr[4] / npass,
                     pr[8] / ngen, pr[9] / ngen, (pr[4] - pr[8] - pr[9]) / ngen, int(ngen), pr[5] / npass))
            us = [x * 16 / 2400.0 for x in ev[:5]]  # cycles at ~2.4 GHz
            n = max(1, ev[5])
            print("kspec %d cluster %2d: trackFrames wall %6.1f us, %3d passes; per pass us: eval %.2f  barrier1 %.2f  sum+exchange %.2f  "
                  "barrier2 %.2f  decision %.2f (sum %.2f)" % (kspec, cl, dt * 1e6, n, us[0] / n, us[1] / n, us[2] / n, us[3] / n, us[4] / n,
                                                              sum(us) / n))
        else:
            print("kspec %d cluster %2d: trackFrames wall %6.1f us, evals %s" % (kspec, cl, dt * 1e6, ev))
