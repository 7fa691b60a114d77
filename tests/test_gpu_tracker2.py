"""GPU parity of the round-2 tracker: the device LDL^T against the oracle's Eigen restatement bit for bit
(well-conditioned, rank-deficient and degenerate systems), speculative LM candidates leave every result
untouched, ill-conditioned frame-pairs, and the exact bench configuration (640x480, 4 levels, 32 pairs in
one batch) against the oracle pair by pair."""
import os
from contextlib import contextmanager

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from revo_amd import synth  # noqa: E402
from revo_amd.settings import ImgPyramidSettings, OptimizerSettings, TrackerSettings, PLANE_DT  # noqa: E402

ROT_TOL = 1e-4   # rad
TRANS_TOL = 1e-4  # m
rot_angle = synth.rot_angle


@pytest.fixture(scope="module")
def api():
    from revo_amd import api as A
    return A


@pytest.fixture(scope="module")
def ro():
    from oracle import ro as R
    return R


@contextmanager
def env(**kv):
    old = {k: os.environ.get(k) for k in kv}
    os.environ.update({k: str(v) for k, v in kv.items()})
    try:
        yield
    finally:
        for k, v in old.items():
            if v is None:
                os.environ.pop(k, None)
            else:
                os.environ[k] = v


def _systems(rng):
    """(A, b, lambda) triples: J^T W J of random Jacobians at the scales the tracker sees, plus the
    degenerate shapes Eigen's LDLT treats specially."""
    out = []
    scales = np.array([300, 300, 120, 350, 350, 90], np.float64)
    for k in range(160):
        n = int(rng.integers(3, 400))
        J = rng.normal(size=(n, 6)) * scales * rng.uniform(0.2, 3.0)
        w = rng.uniform(0.1, 1.0, n)
        A = (J * w[:, None]).T @ J / n
        b = (J * w[:, None]).T @ rng.normal(size=n) / n
        out.append((A, b, [0.0, 0.2, 0.4, 1.6, 12.8, 204.8, 6553.6, 4.2e5, 5.4e7][k % 9]))
    for k in range(40):  # rank-deficient: fewer constraints than unknowns, repeated / zero columns
        n = int(rng.integers(1, 6))
        J = rng.normal(size=(n, 6)) * scales
        if k % 3 == 0:
            J[:, rng.integers(0, 6)] = 0.0
        if k % 3 == 1:
            J[:, 4] = J[:, 3]
        A = J.T @ J / n
        out.append((A, J.T @ rng.normal(size=n) / n, [0.0, 0.2, 3.0][k % 3]))
    Z = np.zeros((6, 6))
    out.append((Z, np.ones(6), 0.0))                      # whole diagonal zero: Eigen leaves the matrix alone
    out.append((Z, np.zeros(6), 0.2))
    D = np.diag([4.0, 4.0, 1.0, 9.0, 9.0, 0.0])           # ties on the diagonal + a zero pivot
    out.append((D, np.arange(1, 7, dtype=np.float64), 0.0))
    out.append((np.eye(6) * 1e-30, np.ones(6) * 1e-30, 0.0))   # tiny but valid pivots
    out.append((np.eye(6) * 1e30, np.ones(6), 1e8))
    E = np.eye(6); E[0, 1] = E[1, 0] = 1.0                 # singular 2x2 block (pivot becomes exactly 0)
    out.append((E, np.ones(6), 0.0))
    N = np.full((6, 6), np.nan)                            # 0 points -> 0/0
    out.append((N, np.full(6, np.nan), 0.0))
    return out


def test_device_ldlt_matches_the_eigen_restatement_bit_for_bit(api, ro):
    """optimizer.cpp:258-262: A(i,i) *= 1 + lambda; inc = A.ldlt().solve(b).  The kernel's row-parallel float
    LDL^T (pivot order from the damped diagonal, left-looking updates, pseudo-inverse of D) must reproduce the
    oracle's serial restatement of Eigen's algorithm exactly -- also where pivoting and zero pivots matter."""
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    sys_ = _systems(np.random.default_rng(11))
    A = np.stack([np.asarray(a, np.float32) for a, _, _ in sys_])
    A = np.stack([(a + a.T) * np.float32(0.5) if np.isfinite(a).all() else a for a in A])  # exactly symmetric in float
    b = np.stack([np.asarray(v, np.float32) for _, v, _ in sys_])
    lam = np.array([l for _, _, l in sys_], np.float32)
    x_g = api.solve6(cam, A, b, lam)
    n_pivoted = 0
    for i in range(len(sys_)):
        Ad = A[i].copy()
        Ad[np.arange(6), np.arange(6)] *= np.float32(1.0) + lam[i]
        x_o = ro.ldlt6_solve(Ad, b[i])   # the oracle takes the damped matrix (column-major == row-major: symmetric)
        assert np.array_equal(x_g[i], x_o, equal_nan=True), (i, x_g[i], x_o)
        d = np.abs(np.diag(Ad))
        n_pivoted += int(np.isfinite(d).all() and np.argmax(d) != 0)
    assert n_pivoted > 50  # the pivoting path is what was tested


def test_device_quaternion_conversions_against_the_sophus_goldens(api, ro):
    """Sophus::SE3f(R, T) at every level entry and SO3::matrix() at every level exit (optimizer.cpp:241,308): with
    zero LM iterations per level the tracker returns matrix(SO3(R_in)) applied once per level -- the device's
    Eigen Quaternionf(Matrix3f) / toRotationMatrix restatements, fed with the rotation matrices of the reference's
    sympy So3 (tests/golden/sophus_se3_golden.json) and compared with the oracle's on the same input."""
    import json
    gold = json.load(open(os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden", "sophus_se3_golden.json")))
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    pair = synth.make_pair(2, s)
    os0 = OptimizerSettings()
    for i in range(6):
        os0.max_its_per_lvl[i] = 0
    ts = TrackerSettings(check_init_values=0)
    cam, g_ref, g_cur, gt = _pair_objects(api, ro, s, pair, ts, os0)
    o_ref, o_cur = ro.Pyramid(s, *pair["ref"]), ro.Pyramid(s, *pair["curr"])
    o_ref.makeKeyframe()
    ot = ro.Tracker(s, os0, ts)
    T0 = np.array([0.01, -0.02, 0.03], np.float32)
    for case in gold["so3"]:
        R_in = np.array(case["matrix"]).astype(np.float32)
        st, R_g, T_g, _ = gt.trackFrames(R_in, T0, g_ref, g_cur)
        r_o = ot.trackFrames(o_ref, o_cur, R_in, T0)
        assert np.array_equal(R_g, r_o["R"]) and np.array_equal(T_g, r_o["T"]), case["omega"]  # same float operations
        assert np.allclose(R_g, np.array(case["matrix"]), atol=5e-6), case["omega"]            # and the golden matrix
        assert gt.last_evals.tolist()[:3] == [1, 1, 1]


def _pair_objects(api, ro, s, pair, ts=None, os_=None):
    ts = ts or TrackerSettings()
    if os_ is not None:
        ts.optimizerSettings = os_
    cam = api.CameraPyr(s)
    g_ref = api.ImgPyramidRGBD(s, cam, *pair["ref"])
    g_cur = api.ImgPyramidRGBD(s, cam, *pair["curr"])
    g_ref.makeKeyframe()
    gt = api.TrackerNew(ts, s, cam)
    return cam, g_ref, g_cur, gt


def test_speculation_depth_and_cluster_shape_leave_the_lm_sequence_untouched(api, ro):
    """One pass evaluates the next LM candidate in full and up to 3 retries error-only; the decision consumes
    them in the reference's order (optimizer.cpp:258-304).  Same poses, same error, same evaluation counts
    for 1, 2 and 4 candidates per pass -- bit for bit, from a good and from a poor prior, default and
    non-default damping schedule."""
    s = ImgPyramidSettings(pyr_min_lvl=3)
    s.hist_patch[3] = 0
    pair = synth.make_pair(3, s)
    os2 = OptimizerSettings()
    os2.lambda_success_fac, os2.lambda_fail_fac = 0.6, 1.7
    for i in range(6):
        os2.lambda_initial[i] = 0.35
    prior = synth.se3_exp([0.03, -0.02, 0.02, 0.01, 0.03, -0.01])
    cases = [(None, np.eye(3), np.zeros(3), 1), (None, prior[:3, :3], prior[:3, 3], 0), (os2, prior[:3, :3], prior[:3, 3], 0)]
    for os_, R0, T0, chk in cases:
        got = {}
        for k in (1, 2, 4):
            with env(REVO_TRACK_KSPEC=k):
                cam, g_ref, g_cur, gt = _pair_objects(api, ro, s, pair, TrackerSettings(check_init_values=chk), os_)
                st, R, T, err = gt.trackFrames(R0, T0, g_ref, g_cur)
                got[k] = (st, R, T, err, gt.last_evals.tolist())
        for k in (2, 4):
            assert got[k][0] == got[1][0] and np.array_equal(got[k][1], got[1][1]) and np.array_equal(got[k][2], got[1][2])
            assert got[k][3] == got[1][3] and got[k][4] == got[1][4], (k, got[k][4], got[1][4])
        assert sum(got[1][4]) > 12
    # the reference's sequence: evaluation counts equal the oracle's for the default schedule from identity
    o_ref, o_cur = ro.Pyramid(s, *pair["ref"]), ro.Pyramid(s, *pair["curr"])
    o_ref.makeKeyframe()
    r_o = ro.Tracker(s).trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
    cam, g_ref, g_cur, gt = _pair_objects(api, ro, s, pair)
    st, R, T, err = gt.trackFrames(np.eye(3), np.zeros(3), g_ref, g_cur)
    print("evals gpu %s oracle %s" % (gt.last_evals.tolist(), r_o["evals"].tolist()))
    assert rot_angle(R, r_o["R"]) < ROT_TOL and np.linalg.norm(T - r_o["T"]) < TRANS_TOL


def _ill_conditioned_inputs(s):
    """(name, ref, curr): scenes whose normal equations are badly conditioned or rank deficient."""
    h, w = s.height, s.width
    rng = np.random.default_rng(5)
    out = []
    # (a) a single fronto-parallel textured plane: translation along / rotation about the axes of the plane are
    #     nearly interchangeable
    def plane(shift):
        yy, xx = np.mgrid[0:h, 0:w]
        g = (((xx + shift) // 23 + yy // 19) % 2) * 140 + 50
        bgr = np.stack([g, g, g], -1).astype(np.uint8)
        return bgr, np.full((h, w), 2.0, np.float32)
    out.append(("plane", plane(0), plane(2)))
    # (b) very few edges: one small square on a flat background (N of a few hundred at level 0, ~0 at level 3)
    def square(dx):
        g = np.full((h, w), 90, np.uint8)
        g[200:240, 300 + dx:340 + dx] = 200
        bgr = np.stack([g, g, g], -1)
        d = np.full((h, w), 1.5, np.float32)
        return bgr, d
    out.append(("few_edges", square(0), square(1)))
    # (c) only vertical lines: motion along them is unobservable (aperture problem -> singular direction)
    def bars(dx):
        g = np.full((h, w), 60, np.uint8)
        for x0 in range(40, w - 40, 64):
            g[:, x0 + dx:x0 + dx + 12] = 190
        bgr = np.stack([g, g, g], -1)
        d = (1.2 + 0.001 * np.arange(w, dtype=np.float32))[None, :].repeat(h, 0)
        return bgr, d.astype(np.float32)
    out.append(("bars", bars(0), bars(1)))
    return out


def test_ill_conditioned_pairs_follow_the_oracle(api, ro):
    """Few edges, a single plane, the aperture problem: Eigen's pivoting / pseudo-inverse of D decide the step
    there (optimizer.cpp:258-262).  The device solve is the oracle's bit for bit on the oracle's own normal
    equations of these pairs, and the whole tracker lands where the oracle lands."""
    s = ImgPyramidSettings(pyr_min_lvl=3)
    s.hist_patch[3] = 0
    for name, ref, cur in _ill_conditioned_inputs(s):
        pair = {"ref": ref, "curr": cur}
        cam, g_ref, g_cur, gt = _pair_objects(api, ro, s, pair)
        o_ref, o_cur = ro.Pyramid(s, *ref), ro.Pyramid(s, *cur)
        o_ref.makeKeyframe()
        ot = ro.Tracker(s)
        conds = []
        for lvl in range(4):
            e_o, info_o, A_o, b_o = ot.eval(o_ref, o_cur, np.eye(3), np.zeros(3), lvl)
            if info_o.good_pts_edges == 0:
                continue
            A_o = np.asarray(A_o, np.float32)
            lams = np.array([0.0, 0.2, 1.6, 204.8], np.float32)
            x_g = api.solve6(cam, np.stack([A_o] * 4), np.stack([-np.asarray(b_o, np.float32)] * 4), lams)
            for k, lam in enumerate(lams):
                Ad = A_o.copy()
                Ad[np.arange(6), np.arange(6)] *= np.float32(1.0) + lam
                assert np.array_equal(x_g[k], ro.ldlt6_solve(Ad, -np.asarray(b_o, np.float32)), equal_nan=True), (name, lvl, lam)
            ev = np.linalg.eigvalsh(np.asarray(A_o, np.float64))
            conds.append(ev[-1] / max(ev[0], 1e-300))
        st_g, R_g, T_g, err_g = gt.trackFrames(np.eye(3), np.zeros(3), g_ref, g_cur)
        r_o = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
        dr, dt = rot_angle(R_g, r_o["R"]), float(np.linalg.norm(T_g - r_o["T"]))
        print("%s: cond(A) up to %.1e, GPU vs oracle %.2e rad %.2e m, err %.5f / %.5f, evals %s / %s"
              % (name, max(conds), dr, dt, err_g, r_o["err"], gt.last_evals.tolist(), r_o["evals"].tolist()))
        assert max(conds) > 5e2, name                      # the case is what it claims to be (bench scenes: ~1e2)
        assert np.isfinite(R_g).all() and np.isfinite(T_g).all() and np.isfinite(err_g), name
        e0 = gt.mOptimizer.evalAt(g_ref, g_cur, np.eye(3), np.zeros(3), 0)[0]
        assert err_g <= e0 * (1 + 1e-6), name              # LM only ever accepts a smaller error
        if max(conds) < 1e6:
            # badly conditioned but determined: the two sides land in the same place
            assert abs(err_g - r_o["err"]) <= 1e-2 * max(r_o["err"], 1e-3), name
            assert dr < 5e-3 and dt < 5e-3, name
        # (numerically singular systems: the step along the null space is decided by the last bits of the sums,
        #  which the GPU orders differently -- the reference itself is chaotic there; the solver parity above is
        #  the testable statement)


def test_bench_configuration_every_pair_against_the_oracle(api, ro):
    """BASELINE configs[2] exactly as bench.py runs it: 640x480, 4 levels, 32 frame-pairs in ONE batch with the
    default cluster, seeds 0..31 -- every pair against the oracle (1e-4 rad / 1e-4 m; a pair may sit inside the
    LM's documented 0.999 convergence slack, the count of those is asserted)."""
    import torch
    s = ImgPyramidSettings.scaled(640, 480, 4, hist_patch=(20, 10, 5, 0, 0, 0))
    n = 32
    pairs = [synth.make_pair(i, s) for i in range(n)]
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).cuda()
    dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).cuda()
    bt = api.BatchTracker(cam, n)
    d_res = torch.zeros(n * 96, dtype=torch.uint8, device="cuda")
    bt.track(bgr.data_ptr(), dep.data_ptr(), d_res.data_ptr())
    bt.sync()
    res = api.results_from_buffer(d_res.cpu().numpy().tobytes(), n)
    ot = ro.Tracker(s)
    drot, dtr, same_evals = [], [], 0
    for i, p in enumerate(pairs):
        o_ref, o_cur = ro.Pyramid(s, *p["ref"]), ro.Pyramid(s, *p["curr"])
        o_ref.makeKeyframe()
        r_o = ot.trackFrames(o_ref, o_cur, np.eye(3), np.zeros(3))
        assert res[i]["flags"] & (2 | 4 | 8) == 0
        drot.append(rot_angle(res[i]["R"], r_o["R"]))
        dtr.append(float(np.linalg.norm(res[i]["T"] - r_o["T"])))
        same_evals += int(res[i]["evals"][:4].tolist() == r_o["evals"][:4].tolist())
        if i < 2:  # the batch's pyramids themselves, bit for bit (full size)
            view = bt.frame(2 * i + 1, s)
            for lvl in range(4):
                assert np.array_equal(view.return3DEdges(lvl), o_cur.read(6, lvl))
    drot, dtr = np.array(drot), np.array(dtr)
    inside = (drot < ROT_TOL) & (dtr < TRANS_TOL)
    print("bench config: %d/%d pairs within 1e-4; max %.2e rad %.2e m; identical evaluation counts: %d/%d"
          % (inside.sum(), n, drot.max(), dtr.max(), same_evals, n))
    assert inside.sum() >= n - 1
    assert drot.max() < 5e-4 and dtr.max() < 5e-4  # the slack of a borderline accept/stop decision, never more (round 6: was n - 2 / 5e-3; measured 32 of 32, max 2.2e-5 rad / 4.2e-5 m)


def test_shared_reciprocal_division_is_bit_identical_to_ieee(tmp_path):
    """revo_div.h (one refined reciprocal shared by the divisions of a projection) returns the bits of __fdiv_rn on
    2^26 operand pairs drawn from the tracker's ranges (tests/cpp/div_exact.hip, compiled here with hipcc)."""
    import shutil
    import subprocess
    hipcc = shutil.which("hipcc") or "/opt/rocm/bin/hipcc"
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    exe = str(tmp_path / "div_exact")
    subprocess.check_call([hipcc, "--offload-arch=gfx950", "-O3", "-ffp-contract=off", "-fno-fast-math", "-w", "-o", exe,
                           os.path.join(root, "tests", "cpp", "div_exact.hip")], timeout=300)
    out = subprocess.run([exe], capture_output=True, text=True, timeout=120)
    assert out.returncode == 0 and "TOTAL_MISMATCHES 0" in out.stdout, out.stdout + out.stderr


def test_tracker_grids_on_two_streams_overlap_under_the_resident_gate(api, ro):
    """Round 3: tracker grids of one device are no longer serialised completely -- grid n+1 (another stream) may start once
    every workgroup of grid n has started (census counter + gate kernel, revo_host.hip).  Three batches on two tracker
    streams, several rounds back to back with nothing waiting in between: every record must equal the record the same
    batch produces alone on one stream, bit for bit, and no record may carry flag 8 (a cluster that never completed)."""
    import torch
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    n = 8
    dev = torch.device("cuda", 0)
    bts, inputs, ref = [], [], []
    for b in range(3):
        pairs = [synth.make_pair(500 + 10 * b + i, s) for i in range(n)]
        bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).to(dev)
        dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).to(dev)
        bt = api.BatchTracker(cam, n)
        res = torch.zeros(n * 96, dtype=torch.uint8, device=dev)
        bt.track(bgr.data_ptr(), dep.data_ptr(), res.data_ptr())
        bt.sync()
        bts.append(bt); inputs.append((bgr, dep)); ref.append(res.cpu().numpy().tobytes())
        assert all(r["flags"] & (2 | 4 | 8) == 0 for r in api.results_from_buffer(ref[-1], n))
    streams = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    s_build = torch.cuda.Stream(device=dev)
    rounds = 6
    outs = [torch.zeros(n * 96, dtype=torch.uint8, device=dev) for _ in range(3 * rounds)]
    built = [torch.cuda.Event() for _ in range(3)]
    tracked = [torch.cuda.Event() for _ in range(3)]
    for t in range(3 * rounds):
        b = t % 3
        st = streams[t % 2]
        s_build.wait_event(tracked[b])
        bts[b].build(inputs[b][0].data_ptr(), inputs[b][1].data_ptr(), stream=s_build.cuda_stream)
        built[b].record(s_build)
        st.wait_event(built[b])
        bts[b].track_only(outs[t].data_ptr(), stream=st.cuda_stream)
        tracked[b].record(st)
    torch.cuda.synchronize()
    for t in range(3 * rounds):
        assert outs[t].cpu().numpy().tobytes() == ref[t % 3], "step %d differs from the batch alone" % t


def test_three_stage_pipeline_with_a_stream_for_the_deferred_kernels(api, ro):
    """Round 4: the shape bench.py measures -- four batches in rotation, builds on one stream, what a build leaves to its
    first consumer (edge lists + keyframe EDT, REVO_DEFER=2) on a stream of its own through revo_batch_prepare, tracker grids
    alternating over two streams.  The library orders the tracker launch, the accessors and the NEXT build of a batch behind
    the prepared work by its own per-FrameSet event: the caller only hands over the build's event.  Every record
    of every round must equal the record the same batch produces alone on one stream, bit for bit."""
    import torch
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    cam = api.CameraPyr(s)
    api.TrackerNew(TrackerSettings(), s, cam)
    n, nb = 8, 4
    dev = torch.device("cuda", 0)
    bts, inputs, ref = [], [], []
    for b in range(nb):
        pairs = [synth.make_pair(700 + 10 * b + i, s) for i in range(n)]
        bgr = torch.from_numpy(np.stack([p[k][0] for p in pairs for k in ("ref", "curr")])).to(dev)
        dep = torch.from_numpy(np.stack([p[k][1] for p in pairs for k in ("ref", "curr")])).to(dev)
        bt = api.BatchTracker(cam, n)
        res = torch.zeros(n * 96, dtype=torch.uint8, device=dev)
        bt.track(bgr.data_ptr(), dep.data_ptr(), res.data_ptr())
        bt.sync()
        bts.append(bt); inputs.append((bgr, dep)); ref.append(res.cpu().numpy().tobytes())
        assert all(r["flags"] & (2 | 4 | 8) == 0 for r in api.results_from_buffer(ref[-1], n))
    s_trk = [torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)]
    s_build, s_aux = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    rounds = 5
    outs = [torch.zeros(n * 96, dtype=torch.uint8, device=dev) for _ in range(nb * rounds)]
    built = [torch.cuda.Event() for _ in range(nb)]
    tracked = [torch.cuda.Event() for _ in range(nb)]
    for t in range(nb * rounds):
        b, st = t % nb, s_trk[t % 2]
        s_build.wait_event(tracked[b])
        bts[b].build(inputs[b][0].data_ptr(), inputs[b][1].data_ptr(), stream=s_build.cuda_stream)
        built[b].record(s_build)
        s_aux.wait_event(built[b])
        bts[b].prepare(stream=s_aux.cuda_stream)
        # NO event from the aux stream to the tracker stream: the library's own event must order the grid behind the prepared work
        bts[b].track_only(outs[t].data_ptr(), stream=st.cuda_stream)
        tracked[b].record(st)
    torch.cuda.synchronize()
    for t in range(nb * rounds):
        assert outs[t].cpu().numpy().tobytes() == ref[t % nb], "step %d differs from the batch alone" % t
    # an accessor on a batch view after the pipelined rounds sees finished lists and DT planes
    v = bts[0].frame(0, s)
    o = ro.Pyramid(s, inputs[0][0][0].cpu().numpy(), inputs[0][1][0].cpu().numpy())
    o.makeKeyframe()
    assert np.array_equal(v.returnDistTransform(0), o.read(PLANE_DT, 0))
