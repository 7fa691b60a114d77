"""revo_track_pairs_* (host-buffer batches with the H2D transfer inside): same bits as the
device-pointer batch, for float32 and raw uint16 depth, padded rows, initial poses, and pipelined submits."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from revo_amd import synth  # noqa: E402
from revo_amd.settings import ImgPyramidSettings, TrackerSettings  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from revo_amd import api as A
    return A


def _same(a, b):
    return (np.array_equal(a["R"], b["R"]) and np.array_equal(a["T"], b["T"]) and a["err"] == b["err"]
            and a["evals"].tolist() == b["evals"].tolist() and a["flags"] == b["flags"] and a["good"] == b["good"])


def test_host_buffer_batch_equals_device_pointer_batch(api):
    import torch
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    n = 5
    pairs = [synth.make_pair(900 + i, s) for i in range(n)]
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    prior = synth.se3_exp([0.004, -0.002, 0.003, 0.002, 0.003, -0.001])
    init = [(prior[:3, :3], prior[:3, 3]) if i % 2 else (np.eye(3), np.zeros(3)) for i in range(n)]
    bt = api.BatchTracker(cam, n)
    d_res = torch.zeros(n * 96, dtype=torch.uint8, device="cuda")
    for u16 in (False, True):
        if u16:
            deps = [[np.clip(p[k][1] * 5000.0, 0, 65535).astype(np.uint16) for k in ("ref", "curr")] for p in pairs]
        else:
            deps = [[p[k][1] for k in ("ref", "curr")] for p in pairs]
        d_bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
        d_dep = torch.from_numpy(np.stack([d for dd in deps for d in dd])).cuda()
        if u16:
            bt.build_u16(d_bgr.data_ptr(), d_dep.data_ptr(), 5000.0)
        else:
            bt.build(d_bgr.data_ptr(), d_dep.data_ptr())
        bt.track_only(d_res.data_ptr(), init_RT=api.pack_init_RT([r for r, _ in init], [t for _, t in init]))
        bt.sync()
        want = api.results_from_buffer(d_res.cpu().numpy().tobytes(), n)
        # host side: rows padded like a cv::Mat with a step (views into wider arrays), one pair page-locked
        host = []
        for i, p in enumerate(pairs):
            fr = []
            for k, name in enumerate(("ref", "curr")):
                b = np.zeros((s.height, s.width + 7, 3), np.uint8)[:, :s.width]
                b[...] = p[name][0]
                d = np.zeros((s.height, s.width + 5), deps[i][k].dtype)[:, :s.width]
                d[...] = deps[i][k]
                if i == 0:
                    b = torch.from_numpy(np.ascontiguousarray(b)).pin_memory().numpy()
                    d = torch.from_numpy(np.ascontiguousarray(d)).pin_memory().numpy()
                fr.append((b, d))
            host.append(tuple(fr))
        hb = api.HostBatchTracker(cam, depth_scale_factor=5000.0 if u16 else None)
        got = hb.track(host, init_RT=init)
        assert all(_same(a, b) for a, b in zip(got, want)), ("u16" if u16 else "f32")
        # three jobs in flight, a fourth is refused, results independent of the pipelining
        jobs = [hb.submit(host, init_RT=init) for _ in range(3)]
        with pytest.raises(api.RevoError) as e:
            hb.submit(host, init_RT=init)
        assert e.value.code == -5
        for j in jobs:
            assert all(_same(a, b) for a, b in zip(hb.wait(j), want))
        errs = [synth.pose_error(r["R"], r["T"], p["T_ref_curr"]) for r, p in zip(got, pairs)]
        assert max(e[0] for e in errs) < 5e-3 and max(e[1] for e in errs) < 5e-3
        # round 6: frames that lie back to back in two page-locked slabs (one per plane type) travel in one copy per slab;
        # partly adjacent layouts (pair 2 elsewhere, pair 3's frames swapped in memory) split into several runs: same records
        slab_c = torch.empty((2 * n + 2, s.height, s.width, 3), dtype=torch.uint8).pin_memory().numpy()
        slab_d = torch.from_numpy(np.zeros((2 * n + 2, s.height, s.width), deps[0][0].dtype).view(np.int16 if u16 else np.float32)).pin_memory().numpy().view(deps[0][0].dtype)
        for layout in ("adjacent", "broken"):
            slot = list(range(2 * n))
            if layout == "broken":
                slot[4], slot[5] = 2 * n, 2 * n + 1      # pair 2 lives behind the others
                slot[6], slot[7] = 7, 6                  # pair 3: current frame in front of the reference frame
            slabbed = []
            for i, p in enumerate(pairs):
                fr = []
                for k, name in enumerate(("ref", "curr")):
                    slab_c[slot[2 * i + k]] = p[name][0]
                    slab_d[slot[2 * i + k]] = deps[i][k]
                    fr.append((slab_c[slot[2 * i + k]], slab_d[slot[2 * i + k]]))
                slabbed.append(tuple(fr))
            assert all(_same(a, b) for a, b in zip(hb.track(slabbed, init_RT=init), want)), (layout, u16)


def test_host_buffer_batch_argument_errors(api):
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    p = synth.make_pair(3, s)
    hb = api.HostBatchTracker(cam)
    with pytest.raises(ValueError):
        hb.track([((p["ref"][0], p["ref"][1].astype(np.float64)), p["curr"])])
    hb16 = api.HostBatchTracker(cam, depth_scale_factor=0.0)
    raw = np.zeros((s.height, s.width), np.uint16)
    with pytest.raises(api.RevoError):
        hb16.track([((p["ref"][0], raw), (p["curr"][0], raw))])
