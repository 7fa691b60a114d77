"""GPU: the keyframes' distance transforms of a batch may run on the build's stream (REVO_EDT_DEFER=0) or be left to the
batch's first consumer (the default): the tracker launch on ITS stream, revo_batch_prepare, or an accessor.  Whoever runs
them, the DT planes (keyframe.cpp:43-58 / imgpyramidrgbd.cpp:241) and the tracker records are the same bits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from revo_amd import synth  # noqa: E402
from revo_amd.settings import ImgPyramidSettings, TrackerSettings  # noqa: E402


def test_deferred_keyframe_edt_gives_the_same_planes_and_records_whoever_runs_it(monkeypatch):
    import torch
    from revo_amd import api
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    n_pairs, levels = 2, 3
    pairs = [synth.make_pair(70 + i, s) for i in range(n_pairs)]
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bgr = np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])
    dep = np.stack([p[k][1] for p in pairs for k in ("ref", "curr")]).astype(np.float32)
    d_bgr, d_dep = torch.from_numpy(bgr).cuda(), torch.from_numpy(dep).cuda()
    side = torch.cuda.Stream()

    def dt_planes(bt):
        return [bt.frame(2 * i, s).returnDistTransform(lvl) for i in range(n_pairs) for lvl in range(levels)]

    def run(defer, first):
        monkeypatch.setenv("REVO_EDT_DEFER", defer)  # read by the library at every build
        bt = api.BatchTracker(cam, n_pairs)
        rec = torch.zeros(n_pairs * 96, dtype=torch.uint8, device="cuda")
        bt.build(d_bgr.data_ptr(), d_dep.data_ptr())
        planes = None
        if first == "accessor":        # an accessor comes before any tracker launch
            planes = dt_planes(bt)
        elif first == "prepare":       # the caller runs the pending part itself, on another stream than the build's
            bt.prepare(stream=side.cuda_stream)  # (ordered behind the build by the library: the set's "built" event)
            side.synchronize()
        bt.track_only(rec.data_ptr())
        bt.sync()
        if planes is None:
            planes = dt_planes(bt)
        return rec.cpu().numpy().tobytes(), planes

    ref_rec, ref_planes = run("0", "tracker")
    assert any(np.isfinite(p).all() and p.max() > 1.0 for p in ref_planes)  # real distance transforms, not empty planes
    for first in ("tracker", "accessor", "prepare"):
        rec, planes = run("1", first)
        assert rec == ref_rec, "tracker records differ (deferred EDT first run by the %s)" % first
        for a, b in zip(planes, ref_planes):
            assert a.shape == b.shape and np.array_equal(a, b), "DT plane differs (deferred EDT first run by the %s)" % first
