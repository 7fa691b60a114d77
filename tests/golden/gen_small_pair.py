"""Generates tests/golden/small_pair.npz: a 160x120 synthetic frame-pair (inputs) and the oracle's
outputs on it (regression fixture: detects drift of the CPU restatement; it is NOT derived from the
reference, which cannot be built here -- see DESIGN.md "Oracle")."""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle import ro  # noqa: E402
from revo_amd import synth  # noqa: E402
from revo_amd.settings import ImgPyramidSettings, PLANE_DT, PLANE_EDGES, PLANE_EDGES3D  # noqa: E402

s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
pair = synth.make_pair(42, s)
ref = ro.Pyramid(s, *pair["ref"])
cur = ro.Pyramid(s, *pair["curr"])
ref.makeKeyframe()
r = ro.Tracker(s).trackFrames(ref, cur, np.eye(3), np.zeros(3))
out = dict(ref_bgr=pair["ref"][0], ref_depth=pair["ref"][1], cur_bgr=pair["curr"][0], cur_depth=pair["curr"][1],
           T_ref_curr=pair["T_ref_curr"], R=r["R"], T=r["T"], err=np.float32(r["err"]), evals=r["evals"],
           good=np.int32(r["info"].good_pts_edges), bad=np.int32(r["info"].bad_pts_edges))
for lvl in range(3):
    out["edges%d" % lvl] = np.packbits(cur.read(PLANE_EDGES, lvl) > 0)
    out["npts%d" % lvl] = np.int32(cur.read(PLANE_EDGES3D, lvl).shape[0])
    out["dt_sum%d" % lvl] = np.float64(ref.read(PLANE_DT, lvl).astype(np.float64).sum())
np.savez_compressed(os.path.join(os.path.dirname(os.path.abspath(__file__)), "small_pair.npz"), **out)
print("wrote small_pair.npz", r["evals"], r["err"])
