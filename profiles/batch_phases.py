"""Phase split of k_track for the bench batch (32 pairs in flight).  Needs a -DREVO_TRACK_PROFILE build of the
library (REVO_HIP_SO=profiles/build/librevo_hip_prof.so): the records then carry cycle counters instead of poses
(revo_track.hip, end of k_track)."""
import multiprocessing as mp
import os
import sys
import numpy as np
sys.path.insert(0, ".")
import bench
NP = int(os.environ.get("PH_PAIRS", "32"))
if __name__ == "__main__":
    from revo_amd.settings import ImgPyramidSettings, TrackerSettings
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    with mp.get_context("fork").Pool(16) as pool:
        rend = pool.map(bench.render_pair, [(sd, 640, 480, 4) for sd in range(NP)])
    import torch
    from revo_amd import api
    dev = torch.device("cuda", 0)
    bgr = torch.from_numpy(np.stack([r[k] for r in rend for k in (0, 2)])).to(dev)
    dep = torch.from_numpy(np.stack([r[k] for r in rend for k in (1, 3)])).to(dev)
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bt = api.BatchTracker(cam, NP)
    res = torch.zeros(NP * 96, dtype=torch.uint8, device=dev)
    st = torch.cuda.Stream()
    bt.build(bgr.data_ptr(), dep.data_ptr(), stream=st.cuda_stream)
    for _ in range(3):
        bt.track_only(res.data_ptr(), stream=st.cuda_stream)
    torch.cuda.synchronize()
    ms = bt.time_tracker(res.data_ptr(), reps=10, stream=st.cuda_stream)
    if os.environ.get("PH_OVERLAP"):  # the phase split NEXT TO a build of another batch (like the bench's steady state)
        bt2 = api.BatchTracker(cam, NP)
        sb = torch.cuda.Stream()
        for _ in range(3):
            bt2.build(bgr.data_ptr(), dep.data_ptr(), stream=sb.cuda_stream, borrow_depth=True)
            bt.track_only(res.data_ptr(), stream=st.cuda_stream)
            torch.cuda.synchronize()
        print("(the numbers below: k_track running next to a build of 64 frames)")
    a = np.frombuffer(res.cpu().numpy().tobytes(), np.float32).reshape(NP, 24)
    ai = np.frombuffer(res.cpu().numpy().tobytes(), np.int32).reshape(NP, 24)
    us = 1.0 / 2400.0
    npts = np.array([[len(bt.frame(2 * i, s).return3DEdges(l)) if hasattr(bt.frame(2 * i, s), "return3DEdges") else 0 for l in range(4)] for i in range(min(NP, 4))])
    print("k_track alone %.1f us for %d pairs; edge points per level (first pairs): %s" % (ms * 1e3, NP, npts.tolist()))
    lvl_us = a[:, 0:4] * us
    lvl_n = a[:, 4:8]
    tot = ai[:, 16:21] * 16 * us  # eval, barrier1, sum+exchange, (unused), decision
    b2 = None
    print("per pair: total in passes %.1f us (min %.1f max %.1f), passes %.1f" % (lvl_us.sum(1).mean(), lvl_us.sum(1).min(), lvl_us.sum(1).max(), ai[:, 21].mean()))
    for l in range(4):
        print("  level %d: %.1f passes, %.1f us per pass, %.1f us in total" % (l, lvl_n[:, l].mean(), (lvl_us[:, l] / np.maximum(lvl_n[:, l], 1)).mean(), lvl_us[:, l].mean()))
    print("  phase totals per pair (us): eval %.1f [F %.1f | E %.1f]  barrier1 %.1f  sum+exchange %.1f  decision %.1f [logic %.1f | solve %.1f]"
          % (tot[:, 0].mean(), a[:, 8].mean() * us, a[:, 9].mean() * us, tot[:, 1].mean(), tot[:, 2].mean(), tot[:, 4].mean(), a[:, 10].mean() * us, a[:, 11].mean() * us))
    try:  # the full per-level table (profile builds of round 3 on)
        import ctypes
        from revo_amd import _lib
        buf = (ctypes.c_float * (64 * NP))()
        if _lib.lib().revo_debug_batch_profile_(buf, NP) == 0:
            t = np.array(buf, np.float32).reshape(NP, 64)
            L = 6
            print("  per level and pass (us):  eval | barrier+sums+exchange | decision")
            for l in range(4):
                n = np.maximum(t[:, 12 + L + l], 1)
                print("    level %d: %5.2f | %5.2f | %5.2f" % (l, (t[:, 12 + 2 * L + l] / n).mean() * us, (t[:, 12 + 3 * L + l] / n).mean() * us,
                                                          (t[:, 12 + 4 * L + l] / n).mean() * us))
    except Exception as e:  # noqa: BLE001
        print("  (no per-level phase table: %s)" % e)
    if os.environ.get("PH_DUMP"):  # one line per pair: where the launch's tail comes from
        print("pair  total_us passes | us per level 0..3 | passes per level 0..3")
        for i in np.argsort(-lvl_us.sum(1)):
            print("%4d  %7.1f %5d  | %s | %s" % (i, lvl_us[i].sum(), ai[i, 21], " ".join("%6.1f" % v for v in lvl_us[i]),
                                              " ".join("%3d" % v for v in lvl_n[i])))
