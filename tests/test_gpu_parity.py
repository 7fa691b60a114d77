"""GPU parity: HIP path (through the C ABI) vs the CPU oracle on identical inputs.

Bar: bit-exact for every integer / byte / index stage and for the per-element
float stages that keep the reference's operation order (gray, pyrDown, depth
subsample, Canny, histogram, fill-in, ordered 3-D edge list, exact EDT, gradient
table); stated SE(3) tolerance for the LM tracker, whose long float sums cannot
be ordered like the reference's sequential loop:
  per evaluation  |err - err_ref| <= 1e-4 * err_ref, A/b relative 1e-4
  per pair        rotation within 1e-4 rad, translation within 1e-4 m of the oracle
"""
import os

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from revo_amd import synth  # noqa: E402
from revo_amd.settings import (ImgPyramidSettings, OptimizerSettings, TrackerSettings, PLANE_GRAY,  # noqa: E402
                               PLANE_DEPTH, PLANE_EDGES, PLANE_EDGES_ORIG, PLANE_DT, PLANE_GRADTABLE,
                               PLANE_EDGES3D, PLANE_HIST)

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OUT = os.path.join(ROOT, "gpurun_out")

ROT_TOL = 1e-4  # rad
TRANS_TOL = 1e-4  # m


def dump(name, **arrays):
    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, "fail_%s.npz" % name), **arrays)


def assert_same(name, a, b):
    a, b = np.asarray(a), np.asarray(b)
    if a.shape != b.shape or not np.array_equal(a, b, equal_nan=True):
        dump(name, gpu=a, oracle=b)
        n = (a != b).sum() if a.shape == b.shape else -1
        raise AssertionError("%s differs: shapes %s %s, %d mismatching elements" % (name, a.shape, b.shape, n))


@pytest.fixture(scope="module")
def api():
    from revo_amd import api as A
    return A


@pytest.fixture(scope="module")
def ro():
    from oracle import ro as R
    return R


def tum_settings(levels=3):
    return ImgPyramidSettings(pyr_min_lvl=levels - 1)


@pytest.fixture(scope="module")
def pair640():
    s = tum_settings(3)
    return s, synth.make_pair(3, s)


def assert_tiled_list(tag, gp, op, lvl):
    """The list the build writes for the tracker: the reference's points, bit for bit, ordered by 32x32-pixel tile (raster
    order of tiles, row-major inside a tile)."""
    ref = op.read(PLANE_EDGES3D, lvl)
    til = gp.edges3DTiled(lvl)
    assert til.shape == ref.shape, "%s: %s vs %s points" % (tag, til.shape, ref.shape)
    if not len(ref):
        return
    K = gp.returnK(lvl).astype(np.float64)
    x = np.rint(til[:, 0].astype(np.float64) * K[0, 0] / til[:, 2] + K[0, 2]).astype(np.int64)
    y = np.rint(til[:, 1].astype(np.float64) * K[1, 1] / til[:, 2] + K[1, 2]).astype(np.int64)
    key = ((y // 32) << 40) | ((x // 32) << 28) | (y << 14) | x
    assert np.all(np.diff(key) > 0), "%s: not in tile order (or a pixel twice)" % tag
    # the reference's order is x outer, y inner: re-sorting the tiled list by (x, y) must give the reference's list exactly
    back = til[np.lexsort((y, x))]
    assert_same(tag, back, ref)


def compare_pyramid(tag, gp, op, s, keyframe):
    for lvl in range(s.nLevels()):
        assert_same("%s_gray%d" % (tag, lvl), gp._read(PLANE_GRAY, lvl), op.read(PLANE_GRAY, lvl))
        assert_same("%s_depth%d" % (tag, lvl), gp._read(PLANE_DEPTH, lvl), op.read(PLANE_DEPTH, lvl))
        assert_same("%s_edgesorig%d" % (tag, lvl), gp._read(PLANE_EDGES_ORIG, lvl), op.read(PLANE_EDGES_ORIG, lvl))
        assert_same("%s_edges%d" % (tag, lvl), gp._read(PLANE_EDGES, lvl), op.read(PLANE_EDGES, lvl))
        if s.hist_patch[lvl] > 0:
            assert_same("%s_hist%d" % (tag, lvl), gp._read(PLANE_HIST, lvl), op.read(PLANE_HIST, lvl))
        assert_same("%s_pts%d" % (tag, lvl), gp.return3DEdges(lvl), op.read(PLANE_EDGES3D, lvl))
        assert_tiled_list("%s_tiled%d" % (tag, lvl), gp, op, lvl)
        if keyframe:
            assert_same("%s_dt%d" % (tag, lvl), gp.returnDistTransform(lvl), op.read(PLANE_DT, lvl))
            assert_same("%s_table%d" % (tag, lvl), gp.returnOptimizationStructure(lvl), op.read(PLANE_GRADTABLE, lvl))


def test_pyramid_and_keyframe_bit_exact_640(api, ro, pair640):
    s, pair = pair640
    cam = api.CameraPyr(s)
    for tag, (bgr, depth) in (("ref", pair["ref"]), ("curr", pair["curr"])):
        gp = api.ImgPyramidRGBD(s, cam, bgr, depth, 1.0)
        op = ro.Pyramid(s, bgr, depth, 1.0)
        gp.makeKeyframe()
        op.makeKeyframe()
        compare_pyramid("p640_" + tag, gp, op, s, True)
        assert gp.returnTimestamp() == 1.0 and gp.isKeyframe()
        n0 = gp.return3DEdges(0).shape[0]
        assert 15000 < n0 < 40000  # the design's edge budget (iowrapperRGBD.cpp:121-122)


@pytest.mark.parametrize("levels,hist", [(4, (20, 10, 5, 0, 0, 0)), (2, (20, 10, 0, 0, 0, 0)), (1, (20, 0, 0, 0, 0, 0)),
                                         (6, (20, 10, 5, 0, 0, 0))])
def test_pyramid_other_level_counts(api, ro, levels, hist):
    """1 level (no pyramid at all) up to REVO_MAX_LEVELS = 6 (at 1280x960: 40x30 at the top; every level must
    hold a multiple of 16 pixels, which a 6-level 640x480 pyramid does not): pyramid + keyframe bit-exact and
    the coarse-to-fine tracker within tolerance."""
    s = (ImgPyramidSettings.scaled(1280, 960, 6, hist_patch=hist) if levels == 6
         else ImgPyramidSettings(pyr_min_lvl=levels - 1, hist_patch=hist))
    pair = synth.make_pair(5, s, max_t=0.01, max_rot_deg=0.3) if levels == 1 else synth.make_pair(5, s)
    bgr, depth = pair["ref"]
    cam = api.CameraPyr(s)
    gp = api.ImgPyramidRGBD(s, cam, bgr, depth)
    op = ro.Pyramid(s, bgr, depth)
    gp.makeKeyframe()
    op.makeKeyframe()
    compare_pyramid("lv%d" % levels, gp, op, s, True)
    ts = TrackerSettings(histogram_level=min(2, levels - 1))
    gt = api.TrackerNew(ts, s, cam)
    ot = ro.Tracker(s, OptimizerSettings(), ts)
    gc = api.ImgPyramidRGBD(s, cam, *pair["curr"])
    oc = ro.Pyramid(s, *pair["curr"])
    st_g, R_g, T_g, err_g = gt.trackFrames(np.eye(3), np.zeros(3), gp, gc)
    r_o = ot.trackFrames(op, oc, np.eye(3), np.zeros(3))
    assert rot_angle(R_g, r_o["R"]) < ROT_TOL and np.linalg.norm(T_g - r_o["T"]) < TRANS_TOL, (levels, T_g, r_o["T"])


def _edge_cases(s):
    h, w = s.height, s.width
    rng = np.random.default_rng(99)
    flat = np.full((h, w, 3), 90, np.uint8)
    d = np.full((h, w), 1.5, np.float32)
    yield "flat", flat, d  # no edges anywhere -> DT sentinel, empty lists
    noise = rng.integers(0, 256, (h, w, 3), dtype=np.uint8)  # dense edges, huge components
    dn = rng.uniform(0.0, 6.0, (h, w)).astype(np.float32)
    dn[rng.uniform(0, 1, (h, w)) < 0.2] = 0.0
    dn[5, 5] = np.nan
    dn[6, 6] = np.inf
    yield "noise", noise, dn
    one = flat.copy()
    one[:, w // 2:] = 200  # one vertical step crossing every tile row
    one[h // 3, :] = 10  # and one horizontal line crossing every tile column
    yield "cross", one, d
    sparse = flat.copy()  # edges confined to a corner: low tile coverage -> fill-in path
    sparse[: h // 6, : w // 6] = rng.integers(0, 2, (h // 6, w // 6, 1), dtype=np.uint8) * 160 + 40
    yield "sparse", sparse, d
    import scipy.ndimage as ndi
    sm = ndi.gaussian_filter(rng.uniform(0, 255, (h, w)), 3.0)
    sm = np.clip((sm - sm.mean()) * 14 + 128, 0, 255).astype(np.uint8)  # long curvy weak/strong chains
    yield "smooth", np.repeat(sm[..., None], 3, 2), d


def test_pyramid_edge_cases_320(api, ro):
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    for name, bgr, depth in _edge_cases(s):
        gp = api.ImgPyramidRGBD(s, cam, bgr, depth)
        op = ro.Pyramid(s, bgr, depth)
        gp.makeKeyframe()
        op.makeKeyframe()
        compare_pyramid("edge_" + name, gp, op, s, True)
    # the cross case must actually have exercised fillInEdges at level 1
    bgr, depth = [c for c in _edge_cases(s) if c[0] == "cross"][0][1:]
    op = ro.Pyramid(s, bgr, depth)
    assert not np.array_equal(op.read(PLANE_EDGES, 1), op.read(PLANE_EDGES_ORIG, 1))


def test_dense_edges_640_take_the_unstaged_compaction_path(api, ro):
    """Dense noise at full size: a 64-column strip of the ordered compaction then holds more points
    than its LDS staging area (4096), so the direct-store path of k_compact_walk runs; the tile-local
    union-find sees huge components.  Bit-exact like everything else."""
    s = tum_settings(3)
    name, bgr, depth = [c for c in _edge_cases(s) if c[0] == "noise"][0]
    cam = api.CameraPyr(s)
    gp = api.ImgPyramidRGBD(s, cam, bgr, depth)
    op = ro.Pyramid(s, bgr, depth)
    gp.makeKeyframe()
    op.makeKeyframe()
    compare_pyramid("dense640", gp, op, s, True)
    assert gp.return3DEdges(0).shape[0] > 10 * 4096  # > 4096 per strip on average (10 strips)


@pytest.mark.parametrize("w,h,levels", [(64, 48, 2), (136, 88, 2), (200, 120, 2), (328, 248, 2), (72, 40, 1), (1120, 624, 3),
                                        (1280, 1024, 3), (1920, 1080, 4)])
def test_unusual_sizes_bit_exact(api, ro, w, h, levels):
    """Widths that are not multiples of the 64-pixel tiles / strips, heights that are not multiples of the
    16-row tiles or 32-row chunks, on smooth random images with depth holes: every plane bit-exact.  1280 x 1024 and
    1920 x 1080 (round 4): level 0's edge bitmap no longer fits one workgroup's LDS (banded hysteresis only), and 1080 rows
    need 34 row chunks / 64 row groups in the column walks (the reference takes any size: camerapyr.h:98-103)."""
    import scipy.ndimage as ndi
    s = ImgPyramidSettings.scaled(w, h, levels, hist_patch=(0, 0, 0, 0, 0, 0))
    r = np.random.default_rng(w * 1000 + h)
    g = ndi.gaussian_filter(r.uniform(0, 255, (h, w, 3)), (2.0, 2.0, 0))
    bgr = np.clip((g - g.mean()) * 9 + 128, 0, 255).astype(np.uint8)
    depth = r.uniform(0.05, 5.5, (h, w)).astype(np.float32)
    depth[r.uniform(0, 1, (h, w)) < 0.15] = 0.0
    cam = api.CameraPyr(s)
    gp = api.ImgPyramidRGBD(s, cam, bgr, depth)
    op = ro.Pyramid(s, bgr, depth)
    gp.makeKeyframe()
    op.makeKeyframe()
    compare_pyramid("size_%dx%d" % (w, h), gp, op, s, True)
    assert gp.return3DEdges(0).shape[0] > 0
    for dense in (False, True):
        assert_same("size_pcl", gp.generateColoredPcl(levels - 1, dense), op.generateColoredPcl(levels - 1, dense))


def test_large_level_last_resort_hysteresis(api, ro):
    """Dense noise at 1280 x 1024: the bands' weak runs exceed their label space, so the (level, frame) falls through to the
    whole-level flood fill -- whose edge bitmap (164 KB) no longer fits a CU's LDS and lives in the scratch plane
    (k_hyst<false, true>).  Edges, lists and DT bit-exact like everywhere else."""
    s = ImgPyramidSettings.scaled(1280, 1024, 3, hist_patch=(20, 10, 5, 0, 0, 0))
    rng = np.random.default_rng(5)
    bgr = rng.integers(0, 256, (1024, 1280, 3), dtype=np.uint8)
    depth = rng.uniform(0.0, 6.0, (1024, 1280)).astype(np.float32)
    depth[rng.uniform(0, 1, depth.shape) < 0.2] = 0.0
    cam = api.CameraPyr(s)
    gp = api.ImgPyramidRGBD(s, cam, bgr, depth)
    op = ro.Pyramid(s, bgr, depth)
    gp.makeKeyframe()
    op.makeKeyframe()
    compare_pyramid("noise1280x1024", gp, op, s, True)


def test_u16_depth_entry_point(api, ro):
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    bgr, depth = synth.make_pair(1, s)["ref"]
    raw = np.clip(depth * 5000.0, 0, 65535).astype(np.uint16)
    cam = api.CameraPyr(s)
    gp = api.ImgPyramidRGBD(s, cam, bgr, raw, depth_scale_factor=5000.0)
    op = ro.Pyramid(s, bgr, ro.u16_to_depth(raw, 5000.0))
    compare_pyramid("u16", gp, op, s, False)


def _setup_pair(api, ro, s, pair, trk_settings=None):
    cam = api.CameraPyr(s)
    g_ref = api.ImgPyramidRGBD(s, cam, *pair["ref"])
    g_cur = api.ImgPyramidRGBD(s, cam, *pair["curr"])
    g_ref.makeKeyframe()
    o_ref = ro.Pyramid(s, *pair["ref"])
    o_cur = ro.Pyramid(s, *pair["curr"])
    o_ref.makeKeyframe()
    ts = trk_settings or TrackerSettings()
    gt = api.TrackerNew(ts, s, cam)
    ot = ro.Tracker(s, OptimizerSettings(), ts)
    return cam, g_ref, g_cur, o_ref, o_cur, gt, ot


rot_angle = synth.rot_angle


def test_residual_and_normal_equations_parity(api, ro, pair640):
    s, pair = pair640
    cam, g_ref, g_cur, o_ref, o_cur, gt, ot = _setup_pair(api, ro, s, pair)
    gt_T = pair["T_ref_curr"]
    poses = [(np.eye(3), np.zeros(3)), (gt_T[:3, :3], gt_T[:3, 3]),
             (synth.se3_exp([0.02, -0.01, 0.03, 0.01, -0.02, 0.015])[:3, :3], np.array([0.02, -0.01, 0.03]))]
    ro.lib().ro_set_accum_double(1)  # compare against the well-rounded sums, then the faithful ones
    try:
        for lvl in range(3):
            for k, (R, T) in enumerate(poses):
                e_g, info_g, A_g, b_g = gt.mOptimizer.evalAt(g_ref, g_cur, R, T, lvl)
                e_o, info_o, A_o, b_o = ot.eval(o_ref, o_cur, R, T, lvl)
                assert info_g.good_pts_edges == info_o.good_pts_edges, (lvl, k)
                assert info_g.bad_pts_edges == info_o.bad_pts_edges, (lvl, k)
                assert abs(e_g - e_o) <= 1e-5 * abs(e_o), (lvl, k, e_g, e_o)
                scale = np.sqrt(np.outer(np.diag(A_o), np.diag(A_o)))
                assert np.all(np.abs(A_g - A_o) <= 1e-4 * scale), (lvl, k)
                assert np.all(np.abs(b_g - b_o) <= 1e-4 * np.sqrt(np.diag(A_o)) * max(1.0, np.sqrt(e_o))), (lvl, k)
    finally:
        ro.lib().ro_set_accum_double(0)
    for lvl in range(3):  # faithful float sums: the SURVEY 8a per-evaluation tolerance
        e_g, info_g, _, _ = gt.mOptimizer.evalAt(g_ref, g_cur, *poses[0], lvl)
        e_o, info_o, _, _ = ot.eval(o_ref, o_cur, *poses[0], lvl)
        assert abs(e_g - e_o) <= 1e-4 * abs(e_o)


def test_track_level_and_track_frames_parity(api, ro, pair640):
    s, pair = pair640
    cam, g_ref, g_cur, o_ref, o_cur, gt, ot = _setup_pair(api, ro, s, pair)
    # one level from identity (Optimizer::trackFrames)
    for lvl in (2, 1):
        e_g, R_g, T_g = gt.mOptimizer.trackFrames(g_ref, g_cur, np.eye(3), np.zeros(3), lvl)
        R_o, T_o, e_o, info_o, ev_o, ab = ot.track_level(o_ref, o_cur, np.eye(3), np.zeros(3), lvl)
        assert rot_angle(R_g, R_o) < ROT_TOL and np.linalg.norm(T_g - T_o) < TRANS_TOL, (lvl, T_g, T_o)
    # full coarse-to-fine (TrackerNew::trackFrames)
    st_g, R_g, T_g, err_g = gt.trackFrames(np.eye(3), np.zeros(3), g_ref, g_cur)
    r_o = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
    dr, dt = rot_angle(R_g, r_o["R"]), float(np.linalg.norm(T_g - r_o["T"]))
    gr, gtr = synth.pose_error(R_g, T_g, pair["T_ref_curr"])
    print("GPU vs oracle: %.2e rad %.2e m; GPU vs ground truth: %.2e rad %.2e m; evals gpu %s oracle %s"
          % (dr, dt, gr, gtr, gt.last_evals.tolist(), r_o["evals"].tolist()))
    assert dr < ROT_TOL and dt < TRANS_TOL
    assert st_g == r_o["status"]
    assert abs(err_g - r_o["err"]) <= 2e-3 * r_o["err"]
    assert gr < 2e-3 and gtr < 2e-3
    # deterministic: same call, same bits
    st2, R2, T2, err2 = gt.trackFrames(np.eye(3), np.zeros(3), g_ref, g_cur)
    assert np.array_equal(R2, R_g) and np.array_equal(T2, T_g) and err2 == err_g


def test_non_default_damping_schedule_parity(api, ro, pair640):
    """optimizer.cpp:288,303: lambda *= successFac / lambda *= std::pow(failFac, incTry) with factors that
    are not powers of two and a non-zero initial lambda, from a deliberately poor prior so that rejected
    steps and retries (incTry > 1) occur.  The kernel evaluates the power as an integer power in
    double-double; the oracle calls libm pow like the reference."""
    s, pair = pair640
    os_ = OptimizerSettings()
    os_.lambda_success_fac = 0.6
    os_.lambda_fail_fac = 1.7
    for i in range(6):
        os_.lambda_initial[i] = 0.35
    ts = TrackerSettings(check_init_values=0)
    ts.optimizerSettings = os_
    cam = api.CameraPyr(s)
    g_ref = api.ImgPyramidRGBD(s, cam, *pair["ref"])
    g_cur = api.ImgPyramidRGBD(s, cam, *pair["curr"])
    g_ref.makeKeyframe()
    o_ref = ro.Pyramid(s, *pair["ref"])
    o_cur = ro.Pyramid(s, *pair["curr"])
    o_ref.makeKeyframe()
    gt = api.TrackerNew(ts, s, cam)
    ot = ro.Tracker(s, os_, ts)
    prior = synth.se3_exp([0.03, -0.02, 0.02, 0.01, 0.03, -0.01])
    st_g, R_g, T_g, err_g = gt.trackFrames(prior[:3, :3], prior[:3, 3], g_ref, g_cur)
    r_o = ot.trackFrames(o_ref, o_cur, prior[:3, :3], prior[:3, 3])
    print("evals gpu %s oracle %s" % (gt.last_evals.tolist(), r_o["evals"].tolist()))
    assert sum(r_o["evals"]) > 3 * s.nLevels()  # the schedule is exercised
    assert rot_angle(R_g, r_o["R"]) < ROT_TOL and np.linalg.norm(T_g - r_o["T"]) < TRANS_TOL
    # same schedule: the two sides count evaluations slightly differently (bookkeeping of the level's first
    # evaluation), never by more than one per level
    assert all(abs(a - b) <= 1 for a, b in zip(gt.last_evals.tolist(), r_o["evals"].tolist()))


def test_init_check_resets_bad_prior(api, ro, pair640):
    s, pair = pair640
    cam, g_ref, g_cur, o_ref, o_cur, gt, ot = _setup_pair(api, ro, s, pair)
    bad = synth.se3_exp([0.25, 0.1, 0.0, 0.0, 0.12, 0.0])  # far-off prior: identity must win (tracker.cpp:277-282)
    st_g, R_g, T_g, err_g = gt.trackFrames(bad[:3, :3], bad[:3, 3], g_ref, g_cur)
    r_o = ot.trackFrames(o_ref, o_cur, bad[:3, :3], bad[:3, 3])
    assert r_o["flags"] & 1
    assert rot_angle(R_g, r_o["R"]) < ROT_TOL and np.linalg.norm(T_g - r_o["T"]) < TRANS_TOL


def test_assess_tracking_quality_parity(api, ro, pair640):
    s, pair = pair640
    cam, g_ref, g_cur, o_ref, o_cur, gt, ot = _setup_pair(api, ro, s, pair)
    assert gt.assessTrackingQuality(np.eye(4), g_cur) == 0  # empty past list -> OK (tracker.cpp:121)
    poses = [np.eye(4), synth.se3_exp([0.01, 0, 0.02, 0, 0.01, 0]), pair["T_ref_curr"], synth.se3_exp([0.3, 0, 0, 0, 0.2, 0])]
    for k, P in enumerate(poses):
        src_g, src_o = (g_ref, o_ref) if k % 2 == 0 else (g_cur, o_cur)
        gt.addOldPclAndPose(src_g, 2, P, float(k))
        ot.addOldPclAndPose(src_o, 2, P, float(k))
        st_g, h_g, o_g = gt.assessTrackingQuality(pair["T_ref_curr"], g_cur, return_hist=True)
        st_o, h_o, o_o = ot.assessTrackingQuality(pair["T_ref_curr"], o_cur)
        assert np.array_equal(h_g, h_o) and np.array_equal(o_g, o_o), (k, h_g, h_o, o_g, o_o)
        assert st_g == st_o
    assert gt.pastSize() == 4
    gt.clearUpPastLists()
    ot.clearUpPastLists()
    assert gt.pastSize() == ot.past_size() == 3
    st_g, h_g, o_g = gt.assessTrackingQuality(pair["T_ref_curr"], g_cur, return_hist=True)
    st_o, h_o, o_o = ot.assessTrackingQuality(pair["T_ref_curr"], o_cur)
    assert np.array_equal(h_g, h_o) and np.array_equal(o_g, o_o) and st_g == st_o


def test_colored_pcl_bit_exact(api, ro, pair640):
    """ImgPyramidRGBD::generateColoredPcl (imgpyramidrgbd.cpp:279-327): the keyframe cloud for the viewer /
    PLY export, sparse (edges) and dense, on every level; also after keyframe promotion and for a
    second pyramid (shared scratch), and its error paths."""
    import ctypes as C
    from revo_amd import _lib
    s, pair = pair640
    cam = api.CameraPyr(s)
    for tag in ("ref", "curr"):
        bgr, depth = pair[tag]
        gp = api.ImgPyramidRGBD(s, cam, bgr, depth, 1.0)
        op = ro.Pyramid(s, bgr, depth, 1.0)
        if tag == "curr":
            gp.makeKeyframe()
        for lvl in range(s.nLevels()):
            for dense in (False, True):
                g = gp.generateColoredPcl(lvl, dense)
                assert_same("clrpcl_%s_%d_%d" % (tag, lvl, dense), g, op.generateColoredPcl(lvl, dense))
        assert np.array_equal(gp.generateColoredPcl(1, False)[:, :4], gp.return3DEdges(1))
        compare_pyramid("clrpcl_after_" + tag, gp, op, s, False)  # the export leaves the pyramid untouched
    n = C.c_size_t()
    buf = np.empty((10, 8), np.float32)
    L = _lib.lib()
    assert L.revo_pyramid_colored_pcl(gp._h, 0, 1, None, 0, C.byref(n)) == 0 and n.value > 200000  # count only
    assert L.revo_pyramid_colored_pcl(gp._h, 0, 0, buf.ctypes.data_as(_lib.f32p), 10, C.byref(n)) == -5  # capacity
    assert n.value == len(gp.return3DEdges(0))
    with pytest.raises(api.RevoError) as e:
        gp.generateColoredPcl(5, False)
    assert e.value.code == -6


def test_error_behaviour(api, pair640):
    s, pair = pair640
    cam = api.CameraPyr(s)
    a = api.ImgPyramidRGBD(s, cam, *pair["ref"])
    b = api.ImgPyramidRGBD(s, cam, *pair["curr"])
    trk = api.TrackerNew(TrackerSettings(), s, cam)
    with pytest.raises(api.RevoError) as e:  # imgpyramidrgbd.h:113-116 "optimizationStructure not built!"
        trk.trackFrames(np.eye(3), np.zeros(3), a, b)
    assert e.value.code == -3
    with pytest.raises(api.RevoError):
        a.returnDistTransform(0)
    a.makeKeyframe()
    Rbad = np.eye(3)
    Rbad[0, 1] = 0.01
    # tracker.cpp:314 runs the init check first: a bad prior that loses to identity is reset and tracked
    st, R, T, err = trk.trackFrames(Rbad, np.zeros(3), a, b)
    assert np.isfinite(err)
    trk2 = api.TrackerNew(TrackerSettings(check_init_values=0), s, cam)
    with pytest.raises(api.RevoError) as e:  # Sophus SO3(R) ENSURE -> abort() in the reference
        trk2.trackFrames(Rbad, np.zeros(3), a, b)
    assert e.value.code == -4
    with pytest.raises(api.RevoError) as e:  # Optimizer::trackFrames builds SE3f(R,T) directly
        trk2.mOptimizer.trackFrames(a, b, Rbad, np.zeros(3), 1)
    assert e.value.code == -4
    with pytest.raises(api.RevoError):
        a.returnGray(7)
    with pytest.raises(ValueError):
        api.ImgPyramidRGBD(s, cam, np.zeros((10, 10, 3), np.uint8), np.zeros((10, 10), np.float32))
    with pytest.raises(api.RevoError):
        api.CameraPyr(ImgPyramidSettings(width=641))


def test_empty_edge_list_is_safe(api, ro):
    """N = 0: the reference divides by zero (optimizer.cpp:190, LGSX.h:323-325); both sides must
    terminate and report the same counts."""
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    flat = np.full((120, 160, 3), 128, np.uint8)
    d = np.full((120, 160), 2.0, np.float32)
    pair = synth.make_pair(2, s)
    cam = api.CameraPyr(s)
    g_ref = api.ImgPyramidRGBD(s, cam, *pair["ref"])
    g_ref.makeKeyframe()
    g_cur = api.ImgPyramidRGBD(s, cam, flat, d)
    trk = api.TrackerNew(TrackerSettings(), s, cam)
    st, R, T, err = trk.trackFrames(np.eye(3), np.zeros(3), g_ref, g_cur)
    assert trk.last_info.good_pts_edges == 0 and trk.last_info.bad_pts_edges == 0
    assert np.isnan(err) and np.allclose(R, np.eye(3)) and np.all(T == 0)


def test_batch_u16_depth_matches_the_u16_single_frame_path(api, ro):
    """revo_batch_build_u16: raw 16-bit depth (the reference's on-disk format, iowrapperRGBD.cpp:326-327)
    converted inside the batched build == the single-frame u16 entry point == the oracle on the converted
    depth, and the tracker result equals the float-depth batch fed with that conversion."""
    import torch
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    n_pairs = 2
    pairs = [synth.make_pair(40 + i, s) for i in range(n_pairs)]
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bgr = np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])
    raw = np.stack([np.clip(p[k][1] * 5000.0, 0, 65535).astype(np.uint16) for p in pairs for k in ("ref", "curr")])
    dep = np.stack([ro.u16_to_depth(r, 5000.0) for r in raw])
    d_bgr, d_raw, d_dep = torch.from_numpy(bgr).cuda(), torch.from_numpy(raw).cuda(), torch.from_numpy(dep).cuda()
    res = []
    for use_u16 in (True, False):
        bt = api.BatchTracker(cam, n_pairs)
        d_res = torch.zeros(n_pairs * 96, dtype=torch.uint8, device="cuda")
        if use_u16:
            bt.build_u16(d_bgr.data_ptr(), d_raw.data_ptr(), 5000.0)
        else:
            bt.build(d_bgr.data_ptr(), d_dep.data_ptr())
        bt.track_only(d_res.data_ptr())
        bt.sync()
        res.append(d_res.cpu().numpy().tobytes())
        if use_u16:
            for f in range(2 * n_pairs):
                view = bt.frame(f, s)
                single = api.ImgPyramidRGBD(s, cam, bgr[f], raw[f], depth_scale_factor=5000.0)
                op = ro.Pyramid(s, bgr[f], dep[f])
                for lvl in range(3):
                    assert_same("bu16_depth", view.returnDepth(lvl), single.returnDepth(lvl))
                    assert_same("bu16_depth_o", view.returnDepth(lvl), op.read(PLANE_DEPTH, lvl))
                    assert_same("bu16_pts", view.return3DEdges(lvl), op.read(PLANE_EDGES3D, lvl))
    assert res[0] == res[1]


def test_batch_build_with_borrowed_depth_equals_the_copying_build(api, ro):
    """revo_batch_build_borrow: level 0 of the depth pyramid is the caller's buffer (no copy).  Every plane, the
    edge lists and the tracker records equal the copying build's, the level-0 accessor returns the input, and a
    copying build afterwards goes back to the batch's own plane."""
    import torch
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    n_pairs = 2
    pairs = [synth.make_pair(60 + i, s) for i in range(n_pairs)]
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bgr = np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])
    dep = np.stack([p[k][1] for p in pairs for k in ("ref", "curr")]).astype(np.float32)
    d_bgr, d_dep = torch.from_numpy(bgr).cuda(), torch.from_numpy(dep).cuda()
    bt_copy, bt_borrow = api.BatchTracker(cam, n_pairs), api.BatchTracker(cam, n_pairs)
    r_copy = torch.zeros(n_pairs * 96, dtype=torch.uint8, device="cuda")
    r_borrow = torch.zeros(n_pairs * 96, dtype=torch.uint8, device="cuda")
    bt_copy.build(d_bgr.data_ptr(), d_dep.data_ptr())
    bt_borrow.build(d_bgr.data_ptr(), d_dep.data_ptr(), borrow_depth=True)
    bt_copy.track_only(r_copy.data_ptr()); bt_borrow.track_only(r_borrow.data_ptr())
    bt_copy.sync(); bt_borrow.sync()
    assert r_copy.cpu().numpy().tobytes() == r_borrow.cpu().numpy().tobytes()
    for f in range(2 * n_pairs):
        a, b = bt_copy.frame(f, s), bt_borrow.frame(f, s)
        for lvl in range(3):
            assert_same("borrow_depth", b.returnDepth(lvl), a.returnDepth(lvl))
            assert_same("borrow_pts", b.return3DEdges(lvl), a.return3DEdges(lvl))
            assert_same("borrow_edges", b.returnEdges(lvl), a.returnEdges(lvl))
        assert_same("borrow_is_input", b.returnDepth(0), dep[f])
    # the borrowed plane is the caller's memory: changing it shows through; a copying build detaches again
    d_dep2 = d_dep.clone()
    bt_borrow.build(d_bgr.data_ptr(), d_dep2.data_ptr(), borrow_depth=True)
    bt_borrow.sync()
    d_dep2[0, 0, 0] = 123.0
    torch.cuda.synchronize()
    assert bt_borrow.frame(0, s).returnDepth(0)[0, 0] == np.float32(123.0)
    bt_borrow.build(d_bgr.data_ptr(), d_dep.data_ptr())
    bt_borrow.sync()
    d_dep2[0, 0, 1] = 77.0
    torch.cuda.synchronize()
    assert_same("copy_again", bt_borrow.frame(0, s).returnDepth(0), dep[0])


def test_many_pairs_pose_parity_statistics(api, ro):
    """48 seeded pairs through one batch vs the oracle pair by pair.  Two faithful implementations of this LM can
    stop at different points inside its 0.999 convergence slack when a decision is borderline, so the bar is
    statistical: almost all pairs agree far below the 1e-4 rad / 1e-4 m tolerance, none is off by more than the
    slack allows (a few mm), and both are equally close to ground truth."""
    import torch
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    n = 48
    pairs = [synth.make_pair(500 + i, s) for i in range(n)]
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bgr = np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])
    dep = np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])
    bt = api.BatchTracker(cam, n)
    d_res = torch.zeros(n * 96, dtype=torch.uint8, device="cuda")
    d_bgr, d_dep = torch.from_numpy(bgr).cuda(), torch.from_numpy(dep).cuda()  # must outlive the asynchronous launch
    bt.track(d_bgr.data_ptr(), d_dep.data_ptr(), d_res.data_ptr())
    bt.sync()
    res = api.results_from_buffer(d_res.cpu().numpy().tobytes(), n)
    ot = ro.Tracker(s, OptimizerSettings(), TrackerSettings())
    drot, dtr, eg, eo = [], [], [], []
    for i, p in enumerate(pairs):
        o_ref, o_cur = ro.Pyramid(s, *p["ref"]), ro.Pyramid(s, *p["curr"])
        o_ref.makeKeyframe()
        r_o = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
        drot.append(rot_angle(res[i]["R"], r_o["R"]))
        dtr.append(float(np.linalg.norm(res[i]["T"] - r_o["T"])))
        eg.append(synth.pose_error(res[i]["R"], res[i]["T"], p["T_ref_curr"])[1])
        eo.append(synth.pose_error(r_o["R"], r_o["T"], p["T_ref_curr"])[1])
        assert res[i]["flags"] & (2 | 8) == 0
    drot, dtr = np.array(drot), np.array(dtr)
    tight = (drot < 1e-5) & (dtr < 1e-5)
    print("pairs within 1e-5: %d/%d; max diff %.2e rad %.2e m; median GT error gpu %.2e oracle %.2e m"
          % (tight.sum(), n, drot.max(), dtr.max(), np.median(eg), np.median(eo)))
    assert tight.sum() >= int(0.9 * n)
    # the stated tolerance, with an explicit allowance: at most 1 of the 48 pairs may sit inside the LM's 0.999
    # convergence slack (a borderline accept/stop decision flips), and that one by less than 5e-4
    # (round 6: was 2 pairs / 5e-3; measured: none outside, max 1.1e-5 rad / 2.3e-5 m)
    outside = ~((drot < ROT_TOL) & (dtr < TRANS_TOL))
    assert outside.sum() <= 1, (np.nonzero(outside)[0].tolist(), drot[outside], dtr[outside])
    assert drot.max() < 5e-4 and dtr.max() < 5e-4
    assert abs(np.median(eg) - np.median(eo)) < 1e-4


def test_batch_matches_single_and_full_size_properties(api, ro, monkeypatch):
    import torch
    # same partition of the point lists in the batch and in the single-pair launch (8 workgroups per pair, same
    # small-level threshold): then the two paths must agree bit for bit
    monkeypatch.setenv("REVO_TRACK_CLUSTER_ONE", "8")
    monkeypatch.setenv("REVO_TRACK_CLUSTER", "8")
    monkeypatch.setenv("REVO_TRACK_REDUNDANT_ONE", "400")
    monkeypatch.setenv("REVO_TRACK_REDUNDANT_BATCH", "400")
    s = tum_settings(4)
    s.hist_patch[3] = 0
    n_pairs = 4
    pairs = [synth.make_pair(100 + i, s) for i in range(n_pairs)]
    cam = api.CameraPyr(s)
    trk = api.TrackerNew(TrackerSettings(), s, cam)
    bgr = np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])
    dep = np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])
    d_bgr = torch.from_numpy(bgr).cuda()
    d_dep = torch.from_numpy(dep).cuda()
    d_res = torch.zeros(n_pairs * 96, dtype=torch.uint8, device="cuda")
    bt = api.BatchTracker(cam, n_pairs)
    bt.track(d_bgr.data_ptr(), d_dep.data_ptr(), d_res.data_ptr())
    bt.sync()
    res = api.results_from_buffer(d_res.cpu().numpy().tobytes(), n_pairs)
    for i, p in enumerate(pairs):
        g_ref = api.ImgPyramidRGBD(s, cam, *p["ref"])
        g_cur = api.ImgPyramidRGBD(s, cam, *p["curr"])
        g_ref.makeKeyframe()
        st, R, T, err = trk.trackFrames(np.eye(3), np.zeros(3), g_ref, g_cur)
        assert np.array_equal(res[i]["R"], R) and np.array_equal(res[i]["T"], T) and res[i]["err"] == err
        assert res[i]["status"] == st and np.array_equal(res[i]["evals"], trk.last_evals)
        # batch planes == single-frame planes, bit for bit
        view = bt.frame(2 * i + 1, s)
        for lvl in range(4):
            assert np.array_equal(view.return3DEdges(lvl), g_cur.return3DEdges(lvl))
            assert np.array_equal(view.returnEdges(lvl), g_cur.returnEdges(lvl))
        kf = bt.frame(2 * i, s)
        assert np.array_equal(kf.returnOptimizationStructure(0), g_ref.returnOptimizationStructure(0))
        # size-independent properties at the full 640x480 size
        dt0 = kf.returnDistTransform(0)
        e0 = kf.returnEdges(0)
        assert np.all(dt0[e0 > 0] == 0) and np.all(dt0[e0 == 0] >= 1)
        gy, gx = np.abs(np.diff(dt0, axis=0)), np.abs(np.diff(dt0, axis=1))
        assert gy.max() <= 1.0 + 1e-6 and gx.max() <= 1.0 + 1e-6  # EDT is 1-Lipschitz
        pts = view.return3DEdges(0)
        u = pts[:, 0] / pts[:, 2] * s.fx + s.cx
        v = pts[:, 1] / pts[:, 2] * s.fy + s.cy
        lin = np.rint(u).astype(np.int64) * s.height + np.rint(v).astype(np.int64)
        assert np.all(np.diff(lin) > 0)  # strictly column-major order, no duplicates
        er, et = synth.pose_error(res[i]["R"], res[i]["T"], p["T_ref_curr"])
        assert er < 2e-3 and et < 3e-3, (i, er, et)
    # oracle parity for the 4-level configuration on pair 0
    o_ref = ro.Pyramid(s, *pairs[0]["ref"])
    o_cur = ro.Pyramid(s, *pairs[0]["curr"])
    o_ref.makeKeyframe()
    r_o = ro.Tracker(s).trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
    assert rot_angle(res[0]["R"], r_o["R"]) < ROT_TOL and np.linalg.norm(res[0]["T"] - r_o["T"]) < TRANS_TOL


def test_1280x960_five_levels(api, ro):
    """BASELINE configs[3]: 1280x960, 5-level pyramid (levels >= 3 without histogram: the reference's
    distPatchSizes has 3 entries)."""
    s = ImgPyramidSettings.scaled(1280, 960, 5, hist_patch=(20, 10, 5, 0, 0, 0))
    pair = synth.make_pair(21, s)
    cam, g_ref, g_cur, o_ref, o_cur, gt, ot = _setup_pair(api, ro, s, pair)
    compare_pyramid("big_ref", g_ref, o_ref, s, True)
    compare_pyramid("big_cur", g_cur, o_cur, s, False)
    st_g, R_g, T_g, err_g = gt.trackFrames(np.eye(3), np.zeros(3), g_ref, g_cur)
    r_o = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
    dr, dt = rot_angle(R_g, r_o["R"]), float(np.linalg.norm(T_g - r_o["T"]))
    print("1280x960: GPU vs oracle %.2e rad %.2e m, evals gpu %s oracle %s, N0 %d"
          % (dr, dt, gt.last_evals.tolist(), r_o["evals"].tolist(), g_cur.return3DEdges(0).shape[0]))
    assert dr < ROT_TOL and dt < TRANS_TOL and st_g == r_o["status"]
    er, et = synth.pose_error(R_g, T_g, pair["T_ref_curr"])
    assert er < 2e-3 and et < 3e-3


def test_1280x960_five_levels_batch_of_eight_pairs(api, ro):
    """BASELINE configs[3] at its bench shape: EIGHT 1280x960 / 5-level pairs through one batch and through
    the pipeline handle -- every pair's pose against the oracle, the pyramids of two frames bit for bit (the banded hysteresis
    is the default at this size), and the pipelined records identical to the batch's."""
    import torch
    s = ImgPyramidSettings.scaled(1280, 960, 5, hist_patch=(20, 10, 5, 0, 0, 0))
    n = 8
    pairs = synth.make_pairs(range(2100, 2100 + n), s)
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
    dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).cuda()
    bt = api.BatchTracker(cam, n)
    d_res = torch.zeros(n * 96, dtype=torch.uint8, device="cuda")
    bt.track(bgr.data_ptr(), dep.data_ptr(), d_res.data_ptr())
    bt.sync()
    raw = d_res.cpu().numpy().tobytes()
    res = api.results_from_buffer(raw, n)
    ot = ro.Tracker(s)
    outside = 0
    for i, p in enumerate(pairs):
        o_ref, o_cur = ro.Pyramid(s, *p["ref"]), ro.Pyramid(s, *p["curr"])
        o_ref.makeKeyframe()
        if i < 1:
            compare_pyramid("big8_ref%d" % i, bt.frame(2 * i, s), o_ref, s, True)
            compare_pyramid("big8_cur%d" % i, bt.frame(2 * i + 1, s), o_cur, s, False)
        r_o = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
        assert res[i]["flags"] & (2 | 4 | 8) == 0
        dr, dt = rot_angle(res[i]["R"], r_o["R"]), float(np.linalg.norm(res[i]["T"] - r_o["T"]))
        outside += not (dr < ROT_TOL and dt < TRANS_TOL)
        assert dr < 5e-3 and dt < 5e-3, (i, dr, dt)
        er, et = synth.pose_error(res[i]["R"], res[i]["T"], p["T_ref_curr"])
        assert er < 3e-3 and et < 5e-3, (i, er, et)
    assert outside <= 1, "%d of %d pairs outside 1e-4 rad / 1e-4 m" % (outside, n)
    pipe = api.Pipeline(cam, n, depth=3)
    outs = [torch.zeros(n * 96, dtype=torch.uint8, device="cuda") for _ in range(5)]
    for o in outs:
        pipe.submit(bgr.data_ptr(), dep.data_ptr(), o.data_ptr())
    pipe.drain()
    for o in outs:
        assert o.cpu().numpy().tobytes() == raw
    pipe.close()


def test_pool_reuse_and_async_build_stress(api, ro):
    """Pyramids are built asynchronously on a second stream into pooled, recycled device sets.  Interleave
    create / track / destroy so sets are reused while earlier work is still in flight: every result must
    equal a fresh computation, bit for bit."""
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    pairs = [synth.make_pair(200 + i, s) for i in range(3)]
    cam = api.CameraPyr(s)
    trk = api.TrackerNew(TrackerSettings(), s, cam)
    ref_results = []
    for p in pairs:  # fresh, sequential
        a = api.ImgPyramidRGBD(s, cam, *p["ref"])
        b = api.ImgPyramidRGBD(s, cam, *p["curr"])
        a.makeKeyframe()
        ref_results.append((trk.trackFrames(np.eye(3), np.zeros(3), a, b), b.return3DEdges(0), a.returnDistTransform(1)))
        del a, b
    rng = np.random.default_rng(0)
    live = []
    for it in range(40):
        k = int(rng.integers(0, 3))
        p = pairs[k]
        a = api.ImgPyramidRGBD(s, cam, *p["ref"])   # these two builds are queued back to back,
        b = api.ImgPyramidRGBD(s, cam, *p["curr"])  # the tracker below waits on their events
        a.makeKeyframe()
        got = trk.trackFrames(np.eye(3), np.zeros(3), a, b)
        exp = ref_results[k][0]
        assert got[0] == exp[0] and np.array_equal(got[1], exp[1]) and np.array_equal(got[2], exp[2]) and got[3] == exp[3], it
        if it % 7 == 0:
            assert np.array_equal(b.return3DEdges(0), ref_results[k][1])
            assert np.array_equal(a.returnDistTransform(1), ref_results[k][2])
        live.append((a, b))
        if len(live) > 3:  # destroy out of order -> recycled sets
            live.pop(int(rng.integers(0, len(live))))


@pytest.mark.parametrize("n_pairs", [1, 3, 9])
def test_batch_sizes_not_multiple_of_eight(api, n_pairs):
    """The tracker grid is padded to groups of 8 pairs (XCD-affine mapping): other sizes must still work."""
    import torch
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    pairs = [synth.make_pair(300 + (i % 3), s) for i in range(n_pairs)]
    cam = api.CameraPyr(s)
    trk = api.TrackerNew(TrackerSettings(), s, cam)
    bgr = np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])
    dep = np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])
    d_bgr, d_dep = torch.from_numpy(bgr).cuda(), torch.from_numpy(dep).cuda()
    d_res = torch.zeros(n_pairs * 96, dtype=torch.uint8, device="cuda")
    bt = api.BatchTracker(cam, n_pairs)
    init = api.pack_init_RT([np.eye(3)] * n_pairs, [np.zeros(3)] * n_pairs)
    for _ in range(2):  # twice: the mailbox / descriptors are re-initialised per launch
        bt.track(d_bgr.data_ptr(), d_dep.data_ptr(), d_res.data_ptr(), init_RT=init)
        bt.sync()
    res = api.results_from_buffer(d_res.cpu().numpy().tobytes(), n_pairs)
    single = {}
    for i, p in enumerate(pairs):
        key = 300 + (i % 3)
        if key not in single:
            a = api.ImgPyramidRGBD(s, cam, *p["ref"])
            b = api.ImgPyramidRGBD(s, cam, *p["curr"])
            a.makeKeyframe()
            single[key] = trk.trackFrames(np.eye(3), np.zeros(3), a, b)
        st, R, T, err = single[key]
        # the cluster size differs between the 1-pair and the n-pair launch: same algorithm, sums grouped
        # differently -> equal within the SE(3) tolerance, not bitwise
        assert synth.rot_angle(res[i]["R"], R) < ROT_TOL and np.linalg.norm(res[i]["T"] - T) < TRANS_TOL
        assert res[i]["flags"] == 0
