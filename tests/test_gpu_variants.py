"""GPU parity of the code paths that are NOT the default at the test sizes (VERDICT r03 "what's weak" 2-3): they used to be
checked by hand-run tools only.

* the banded hysteresis (k_hyst_band / k_hyst_seam / k_hyst_out) forced at sizes where one workgroup per level would do,
  with three band sizes -- bit-exact vs the oracle on the Canny edge cases;
* a 128-pair slice of tests/tools/soak_gpu_tracker.py at the bench geometry: the DISTRIBUTION behind the stated tracker
  tolerance (>= 97 % of the pairs within 1e-5 rad / 1e-5 m of the faithful oracle, <= 2 % outside 1e-4, all within 5e-4;
  DESIGN section 4) -- and the same pairs against the oracle with DOUBLE sums, whose accept / reject sequence the device follows.
"""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# share of pairs on which the device takes exactly the accept / reject sequence of the oracle with double sums (measured: 98 of
# 128 = 77 %, against 77 = 60 % for the float-sequential oracle; 127 of 128 poses within 1e-6: profiles/r06_parity_double_error_sums.txt)
SAME_COUNTS_VS_DOUBLE_ORACLE = 0.70

from revo_amd import synth  # noqa: E402
from revo_amd.settings import ImgPyramidSettings, OptimizerSettings, TrackerSettings  # noqa: E402

from test_gpu_parity import _edge_cases, compare_pyramid  # noqa: E402


@pytest.fixture(scope="module")
def api():
    from revo_amd import api as A
    return A


@pytest.fixture(scope="module")
def ro():
    from oracle import ro as R
    return R


@pytest.mark.parametrize("size", [(320, 240), (640, 480)])
@pytest.mark.parametrize("band_words", [600, 1200, 2400])
def test_banded_hysteresis_forced_bit_exact(api, ro, monkeypatch, size, band_words):
    """REVO_HYST_BANDED=1 makes every level take the banded path (several workgroups per level and frame + the exact seam
    pass), REVO_HYST_BAND_WORDS sets the band height: 2 to 16 bands per level here.  The library reads both when the
    context is created.  Everything downstream of the edges (histogram, fill-in, lists, DT) rides along."""
    w, h = size
    monkeypatch.setenv("REVO_HYST_BANDED", "1")
    monkeypatch.setenv("REVO_HYST_BAND_WORDS", str(band_words))
    s = ImgPyramidSettings.scaled(w, h, 3, hist_patch=(20, 10, 5, 0, 0, 0) if w == 640 else (10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    for name, bgr, depth in _edge_cases(s):
        gp = api.ImgPyramidRGBD(s, cam, bgr, depth)
        op = ro.Pyramid(s, bgr, depth)
        compare_pyramid("band%d_%dx%d_%s" % (band_words, w, h, name), gp, op, s, False)


def test_banded_and_single_hysteresis_agree_on_a_batch(api, monkeypatch):
    """The same 8 frames through a batch build with the banded path forced and with the single workgroup forced: identical
    edge planes, histograms and tracker-ordered lists (the two paths never run in the same context otherwise)."""
    import torch
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    pairs = [synth.make_pair(900 + i, s) for i in range(4)]
    bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
    dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).cuda()
    recs = {}
    for force in ("1", "0"):
        monkeypatch.setenv("REVO_HYST_BANDED", force)
        cam = api.CameraPyr(s)
        api.TrackerNew(TrackerSettings(), s, cam)
        bt = api.BatchTracker(cam, 4)
        res = torch.zeros(4 * 96, dtype=torch.uint8, device="cuda")
        bt.track(bgr.data_ptr(), dep.data_ptr(), res.data_ptr())
        bt.sync()
        planes = []
        for f in range(8):
            v = bt.frame(f, s)
            for lvl in range(4):
                planes.append(v.returnEdges(lvl).copy())
                planes.append(v.edges3DTiled(lvl).copy())
        recs[force] = (res.cpu().numpy().tobytes(), planes)
    assert recs["1"][0] == recs["0"][0], "tracker records differ between the two hysteresis paths"
    for a, b in zip(recs["1"][1], recs["0"][1]):
        assert np.array_equal(a, b)


@pytest.mark.parametrize("size,band_words", [((640, 480), 2400), ((320, 240), 600)])
@pytest.mark.parametrize("heavy_runs", [1, 400])
def test_mixed_hysteresis_bit_exact(api, ro, monkeypatch, size, band_words, heavy_runs):
    """Round 6: levels that fit one workgroup take the MIXED hysteresis -- frames whose level 0 has at least
    REVO_HYST_HEAVY_RUNS weak runs are closed by band workgroups inside the same launch, the others by one workgroup
    (k_hyst_mixed + the seam / output passes of the heavy frames).  Threshold 1 sends every frame with a weak pixel through the
    bands, 400 splits the edge cases between the two paths: bit-exact vs the oracle either way (imgpyramidrgbd.cpp:184)."""
    w, h = size
    monkeypatch.setenv("REVO_HYST_HEAVY_RUNS", str(heavy_runs))
    monkeypatch.setenv("REVO_HYST_BAND_WORDS", str(band_words))
    s = ImgPyramidSettings.scaled(w, h, 3, hist_patch=(20, 10, 5, 0, 0, 0) if w == 640 else (10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    for name, bgr, depth in _edge_cases(s):
        gp = api.ImgPyramidRGBD(s, cam, bgr, depth)
        op = ro.Pyramid(s, bgr, depth)
        compare_pyramid("mixed%d_%dx%d_%s" % (heavy_runs, w, h, name), gp, op, s, False)


def test_mixed_and_single_hysteresis_agree_on_the_bench_frames(api, monkeypatch):
    """The first 8 bench pairs (16 frames of 640x480 x 4 levels, among them the low-contrast frames with ~10 000 weak runs that
    make the single-workgroup launch as long as it is) through a batch as shipped (one workgroup per frame), with every frame
    forced through the bands, and with the mix at 4000 runs (only the low-contrast frames banded): identical edge planes,
    histograms, tile-ordered lists and tracker records."""
    import torch
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    pairs = synth.make_pairs(range(8), s)
    bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
    dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).cuda()
    recs = {}
    for heavy in ("default", "1", "4000"):  # default = off (one workgroup per frame); 4000 = the low-contrast frames through the bands
        if heavy == "default":
            monkeypatch.delenv("REVO_HYST_HEAVY_RUNS", raising=False)
        else:
            monkeypatch.setenv("REVO_HYST_HEAVY_RUNS", heavy)
        cam = api.CameraPyr(s)
        api.TrackerNew(TrackerSettings(), s, cam)
        bt = api.BatchTracker(cam, 8)
        res = torch.zeros(8 * 96, dtype=torch.uint8, device="cuda")
        bt.track(bgr.data_ptr(), dep.data_ptr(), res.data_ptr())
        bt.sync()
        planes = []
        for f in range(16):
            v = bt.frame(f, s)
            for lvl in range(4):
                planes.append(v.returnEdges(lvl).copy())
                planes.append(v.returnHist(lvl).copy() if s.hist_patch[lvl] > 0 else np.zeros(1))
                planes.append(v.edges3DTiled(lvl).copy())
        recs[heavy] = (res.cpu().numpy().tobytes(), planes)
    for other in ("1", "4000"):
        assert recs["default"][0] == recs[other][0], "tracker records differ (REVO_HYST_HEAVY_RUNS=%s)" % other
        for a, b in zip(recs["default"][1], recs[other][1]):
            assert np.array_equal(a, b)


@pytest.mark.parametrize("size,levels", [((640, 480), 4), ((320, 240), 3), ((160, 120), 3)])
def test_staged_edge_depths_give_the_same_lists(api, ro, monkeypatch, size, levels):
    """Round 6: in a pipelined batch the depth half of pyrDown runs behind Canny, so it also stages the
    depths of the level's EDGE pixels in list order (k_edge_prefix + k_pyrdown<.., STAGE>), and k_pts_tiles reads those instead
    of gathering from the depth planes.  Same points, same order, same bits as with REVO_STAGE_EDGE_DEPTHS=0 -- on the Canny /
    depth edge cases (NaN, inf, holes, dense noise, fill-in) and on rendered pairs -- and the reference-ordered accessor list
    still equals the oracle's (imgpyramidrgbd.cpp:199-226)."""
    import torch
    w, h = size
    hp = (20, 10, 5, 0, 0, 0) if w == 640 else ((10, 5, 0, 0, 0, 0) if w == 320 else (5, 0, 0, 0, 0, 0))
    s = ImgPyramidSettings.scaled(w, h, levels, hist_patch=hp)
    frames = [(bgr, depth) for _, bgr, depth in _edge_cases(s)]
    for i in range(2):
        p = synth.make_pair(40 + i, s)
        frames += [p["ref"], p["curr"]]
    if len(frames) % 2:
        frames.append(frames[0])
    n = len(frames) // 2
    bgr = torch.from_numpy(np.stack([f[0] for f in frames])).cuda()
    dep = torch.from_numpy(np.stack([f[1] for f in frames]).astype(np.float32)).cuda()
    got = {}
    for stage in ("1", "0"):
        monkeypatch.setenv("REVO_STAGE_EDGE_DEPTHS", stage)
        cam = api.CameraPyr(s)
        api.TrackerNew(TrackerSettings(), s, cam)
        bt = api.BatchTracker(cam, n)
        res = torch.zeros(n * 96, dtype=torch.uint8, device="cuda")
        bt.track(bgr.data_ptr(), dep.data_ptr(), res.data_ptr())
        bt.sync()
        lists = [bt.frame(f, s).edges3DTiled(lvl).copy() for f in range(2 * n) for lvl in range(levels)]
        got[stage] = (res.cpu().numpy().tobytes(), lists)
        if stage == "1":  # the accessor's list (built on demand from the planes) against the oracle, frame by frame
            for f in (0, 1, 2 * n - 1):
                o = ro.Pyramid(s, frames[f][0], frames[f][1])
                for lvl in range(levels):
                    assert np.array_equal(bt.frame(f, s).return3DEdges(lvl), o.read(6, lvl), equal_nan=True)
    assert sum(len(x) for x in got["1"][1]) > 1000
    for a, b in zip(got["1"][1], got["0"][1]):
        assert a.shape == b.shape and np.array_equal(a.view(np.uint32), b.view(np.uint32)), "tile-ordered list differs"
    assert got["1"][0] == got["0"][0], "tracker records differ"


def test_tracker_tolerance_distribution_128_pairs(api, ro, capsys):
    """The soak tool's statement as a test: 128 seeded 640x480 / 4-level pairs through bench-sized batches (32 pairs, the
    bench's cluster shape) against the oracle, pair by pair -- TWICE:

    (a) against the FAITHFUL oracle (float sums accumulated sequentially in the reference's list order, LGSX.h:392-398,
        optimizer.cpp:129-133).  Two faithful implementations of this LM may stop at different points inside its convergence
        slack (borderline `error < lastErr` / `> 0.999` decisions on sums of ~1e4 float terms, optimizer.cpp:273-278), so the
        tolerance is a distribution: >= 97 % of the pairs within 1e-5 rad / 1e-5 m, <= 2 % outside 1e-4, none above 5e-4.
    (b) against the same oracle with its sums accumulated in DOUBLE (ro_set_accum_double: same algorithm, same order, the
        rounding noise of the sequential float sums removed).  The device sums per thread, folds by butterflies and finishes in
        double, i.e. it is close to the exact sums: if the float noise of the reference's own sums is what flips the borderline
        decisions, the device must follow THIS oracle's accept / reject sequence -- its per-level evaluation counts -- more
        often than (a)'s, its poses must agree to the last digits wherever the counts agree, and almost all poses must agree
        to 1e-6 (a flipped borderline decision late in a level usually ends at the same pose: measured 127 of 128 within 1e-6
        while 98 of 128 count sequences are identical.  Round 6: the error sums the decisions compare are carried in double on the
        device too, and the share stayed at 77 %: what flips the remaining decisions is the last bit of the 27 normal-equation
        entries the candidate POSE is solved from -- the oracle against itself with +-1 ulp on A, b shows the same 77 %
        (tests/test_oracle_tracker.py::test_one_ulp_on_the_normal_equations_changes_the_lm_sequence)).
    tests/test_oracle_tracker.py shows the CPU-only half of the argument: the oracle against ITSELF (float vs double sums)
    disagrees exactly like (a)."""
    import torch
    n, seed0 = 128, 1000
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bt = api.BatchTracker(cam, 32)
    ot = ro.Tracker(s, OptimizerSettings(), TrackerSettings())
    L = ro.lib()
    drot, dtr, same_evals, flagged = [], [], 0, 0
    drot_d, dtr_d, same_d = [], [], []
    try:
        for b0 in range(0, n, 32):
            pairs = synth.make_pairs(range(seed0 + b0, seed0 + b0 + 32), s)  # rendered on several host cores
            bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
            dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).cuda()
            d_res = torch.zeros(32 * 96, dtype=torch.uint8, device="cuda")
            bt.track(bgr.data_ptr(), dep.data_ptr(), d_res.data_ptr())
            bt.sync()
            res = api.results_from_buffer(d_res.cpu().numpy().tobytes(), 32)
            for i, p in enumerate(pairs):
                o_ref, o_cur = ro.Pyramid(s, *p["ref"]), ro.Pyramid(s, *p["curr"])
                o_ref.makeKeyframe()
                L.ro_set_accum_double(0)
                r_o = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
                drot.append(synth.rot_angle(res[i]["R"], r_o["R"]))
                dtr.append(float(np.linalg.norm(res[i]["T"] - r_o["T"])))
                same_evals += list(res[i]["evals"][:4]) == list(r_o["evals"][:4])
                flagged += bool(res[i]["flags"] & (2 | 4 | 8))
                L.ro_set_accum_double(1)
                r_d = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
                drot_d.append(synth.rot_angle(res[i]["R"], r_d["R"]))
                dtr_d.append(float(np.linalg.norm(res[i]["T"] - r_d["T"])))
                same_d.append(list(res[i]["evals"][:4]) == list(r_d["evals"][:4]))
    finally:
        L.ro_set_accum_double(0)
    drot, dtr = np.array(drot), np.array(dtr)
    drot_d, dtr_d, same_d = np.array(drot_d), np.array(dtr_d), np.array(same_d)
    in5 = int(((drot < 1e-5) & (dtr < 1e-5)).sum())
    out4 = int(((drot >= 1e-4) | (dtr >= 1e-4)).sum())
    in5_d = int(((drot_d < 1e-5) & (dtr_d < 1e-5)).sum())
    in6_d = int(((drot_d < 1e-6) & (dtr_d < 1e-6)).sum())
    out4_d = int(((drot_d >= 1e-4) | (dtr_d >= 1e-4)).sum())
    worst_same = (float(drot_d[same_d].max()), float(dtr_d[same_d].max())) if same_d.any() else (0.0, 0.0)
    with capsys.disabled():
        print("\n[soak slice] %d pairs vs the faithful (float-sequential) oracle: within 1e-5: %d, outside 1e-4: %d, max %.2e rad %.2e m, "
              "median %.2e rad %.2e m, identical evaluation counts: %d (%.0f %%)"
              % (n, in5, out4, drot.max(), dtr.max(), np.median(drot), np.median(dtr), same_evals, 100.0 * same_evals / n))
        print("[soak slice] %d pairs vs the double-accumulating oracle: identical evaluation counts: %d (%.0f %%), within 1e-6: %d, "
              "within 1e-5: %d, outside 1e-4: %d, max %.2e rad %.2e m, median %.2e rad %.2e m; pairs with identical counts: max %.2e rad %.2e m"
              % (n, int(same_d.sum()), 100.0 * same_d.mean(), in6_d, in5_d, out4_d, drot_d.max(), dtr_d.max(), np.median(drot_d),
                 np.median(dtr_d), worst_same[0], worst_same[1]))
    assert flagged == 0
    assert in5 >= 0.97 * n, "only %d of %d pairs within 1e-5" % (in5, n)
    assert out4 <= 0.02 * n, "%d of %d pairs outside 1e-4" % (out4, n)
    assert drot.max() < 5e-4 and dtr.max() < 5e-4  # (round 6: was 5e-3; measured 8.0e-5 rad / 1.05e-4 m)
    # (b): the device follows the well-rounded reference
    assert same_d.mean() >= SAME_COUNTS_VS_DOUBLE_ORACLE, "identical evaluation counts vs the double-accumulating oracle: %d of %d" % (same_d.sum(), n)
    assert same_d.sum() > same_evals, "the device does not follow the double-accumulating oracle more often than the float one"
    assert worst_same[0] < 1e-6 and worst_same[1] < 1e-6, "same accept/reject sequence but different poses: %r" % (worst_same,)
    assert in5_d >= 0.97 * n and out4_d <= 0.02 * n and drot_d.max() < 5e-4 and dtr_d.max() < 5e-4  # (measured 1.25e-4 rad / 3.2e-4 m)
    # ... and to SIX digits on all but the borderline pairs: the device IS the well-rounded reference (measured 127 of 128)
    assert in6_d >= 0.97 * n, "only %d of %d pairs within 1e-6 of the double-accumulating oracle" % (in6_d, n)
