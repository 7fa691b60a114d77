"""Tracker soak at the bench geometry (640x480, 4 levels): N seeded pairs through batches of 32 (the bench's cluster
size) vs the oracle pair by pair -- pose differences, evaluation counts, good/bad counts.  Not collected by pytest:
`python tests/tools/soak_gpu_tracker.py [n_pairs] [first_seed]` on a GPU box."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402
from oracle import ro  # noqa: E402
from revo_amd import api, synth  # noqa: E402
from revo_amd.settings import ImgPyramidSettings, OptimizerSettings, TrackerSettings  # noqa: E402


rot_angle = synth.rot_angle  # (the skew part of the relative rotation: arccos of the trace loses everything below 3e-4 rad in float)


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 128
    seed0 = int(sys.argv[2]) if len(sys.argv) > 2 else 1000
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bt = api.BatchTracker(cam, 32)
    ot = ro.Tracker(s, OptimizerSettings(), TrackerSettings())
    drot, dtr, ev_same, worst = [], [], 0, []
    for b0 in range(0, n, 32):
        pairs = synth.make_pairs(range(seed0 + b0, seed0 + b0 + 32), s)  # rendered on several host cores
        bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
        dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).cuda()
        d_res = torch.zeros(32 * 96, dtype=torch.uint8, device="cuda")
        bt.track(bgr.data_ptr(), dep.data_ptr(), d_res.data_ptr())
        bt.sync()
        res = api.results_from_buffer(d_res.cpu().numpy().tobytes(), 32)
        for i, p in enumerate(pairs):
            if b0 + i >= n:
                break
            o_ref, o_cur = ro.Pyramid(s, *p["ref"]), ro.Pyramid(s, *p["curr"])
            o_ref.makeKeyframe()
            r_o = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
            dr, dt = rot_angle(res[i]["R"], r_o["R"]), float(np.linalg.norm(res[i]["T"] - r_o["T"]))
            drot.append(dr); dtr.append(dt)
            same_ev = list(res[i]["evals"][:4]) == list(r_o["evals"][:4])
            ev_same += same_ev
            if dr > 1e-4 or dt > 1e-4 or res[i]["flags"] & (2 | 4 | 8):
                worst.append((seed0 + b0 + i, dr, dt, res[i]["flags"], list(res[i]["evals"][:4]), list(r_o["evals"][:4])))
    drot, dtr = np.array(drot), np.array(dtr)
    print("pairs %d: within 1e-5: %d, within 1e-4: %d; max %.2e rad %.2e m; median %.2e rad %.2e m; identical evaluation counts: %d"
          % (len(drot), int(((drot < 1e-5) & (dtr < 1e-5)).sum()), int(((drot < 1e-4) & (dtr < 1e-4)).sum()), drot.max(), dtr.max(),
             np.median(drot), np.median(dtr), ev_same))
    for w in worst:
        print("  outside 1e-4: seed %d  %.2e rad %.2e m flags %d evals gpu %s oracle %s" % w)
    return 0


if __name__ == "__main__":
    sys.exit(main())
