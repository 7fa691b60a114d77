"""Deterministic synthetic RGB-D renderer (SURVEY.md 8(d) "Synthetic inputs").

No TUM data and no network exist in this environment, so every benchmark and
parity input is rendered from a seed: a piecewise-planar room plus random
axis-aligned boxes, world-anchored checker textures (strong intensity edges),
depth discontinuities with zero-valued holes (exercise the hole-aware depth
subsample and the validity test of the 3-D edge list), and a known camera pose
per frame so pose error / ATE are exact.

Conventions: camera looks along +z, x right, y down (pinhole of
config/dataset_tum1.yaml); pose T_w_c maps camera points to world.
"""
import numpy as np


def se3_exp(xi):
    """expm(hat(xi)) for xi = [v, w] (float64) -> 4x4."""
    xi = np.asarray(xi, np.float64)
    v, w = xi[:3], xi[3:]
    th = np.linalg.norm(w)
    W = np.array([[0, -w[2], w[1]], [w[2], 0, -w[0]], [-w[1], w[0], 0]])
    if th < 1e-10:
        R = np.eye(3) + W
        V = np.eye(3) + 0.5 * W
    else:
        R = np.eye(3) + np.sin(th) / th * W + (1 - np.cos(th)) / th ** 2 * W @ W
        V = np.eye(3) + (1 - np.cos(th)) / th ** 2 * W + (th - np.sin(th)) / th ** 3 * W @ W
    T = np.eye(4)
    T[:3, :3] = R
    T[:3, 3] = V @ v
    return T


class Scene:
    """Room (6 faces) + n_boxes axis-aligned boxes, each face with its own texture."""

    def __init__(self, seed, n_boxes=12):
        rng = np.random.default_rng(seed)
        self.seed = seed
        # room extents (metres): camera starts near the origin
        self.room_lo = np.array([-2.6 - rng.uniform(0, 0.6), -1.6 - rng.uniform(0, 0.3), -1.5])
        self.room_hi = np.array([2.6 + rng.uniform(0, 0.6), 1.3 + rng.uniform(0, 0.3), 3.6 + rng.uniform(0, 0.9)])
        c = np.stack([rng.uniform(-2.0, 2.0, n_boxes), rng.uniform(-0.9, 1.0, n_boxes),
                      rng.uniform(1.3, 3.4, n_boxes)], 1)
        hs = rng.uniform(0.10, 0.38, (n_boxes, 3))
        self.box_lo = c - hs
        self.box_hi = c + hs
        n_faces = 6 * (1 + n_boxes)
        self.base = rng.uniform(60, 200, n_faces)
        self.amp = rng.uniform(35, 85, n_faces) * rng.choice([-1, 1], n_faces)
        self.period = rng.uniform(0.11, 0.30, (n_faces, 2))
        self.phase = rng.uniform(0, 1, (n_faces, 2))
        self.tint = rng.uniform(0.75, 1.0, (n_faces, 3))
        self.shade_k = rng.uniform(0.5, 1.5, (n_faces, 2))

    def render(self, T_w_c, width, height, fx, fy, cx, cy, noise_seed=0, hole_frac=0.03,
               noise_sigma=2.0):
        """-> (bgr u8 [H,W,3], depth f32 [H,W] metres with 0 = hole)."""
        T_w_c = np.asarray(T_w_c, np.float64)
        R, t = T_w_c[:3, :3], T_w_c[:3, 3]
        xs = (np.arange(width, dtype=np.float64) - cx) / fx
        ys = (np.arange(height, dtype=np.float64) - cy) / fy
        dx, dy = np.meshgrid(xs, ys)
        d_c = np.stack([dx, dy, np.ones_like(dx)], -1).reshape(-1, 3)  # z = 1 => param == depth
        d_w = d_c @ R.T
        o = t
        n = d_w.shape[0]
        best_t = np.full(n, np.inf)
        best_face = np.zeros(n, np.int64)
        inv = 1.0 / np.where(np.abs(d_w) < 1e-12, 1e-12, d_w)
        # room: camera inside, take the exit distance
        t1 = (self.room_lo - o) * inv
        t2 = (self.room_hi - o) * inv
        tfar = np.maximum(t1, t2)
        ax = np.argmin(tfar, 1)
        texit = tfar[np.arange(n), ax]
        side = (d_w[np.arange(n), ax] > 0).astype(np.int64)
        best_t = texit
        best_face = ax * 2 + side
        # boxes: slab entry distance, evaluated only inside each box's screen-space bounding
        # rectangle (identical result: rays outside it cannot hit the box)
        Rt = R.T
        for b in range(self.box_lo.shape[0]):
            lo, hi = self.box_lo[b], self.box_hi[b]
            corners = np.array([[x, y, z] for x in (lo[0], hi[0]) for y in (lo[1], hi[1]) for z in (lo[2], hi[2])])
            cc = (corners - t) @ Rt.T  # world -> camera
            if np.any(cc[:, 2] < 0.05):
                x0, x1, y0, y1 = 0, width, 0, height
            else:
                uu = cc[:, 0] / cc[:, 2] * fx + cx
                vv = cc[:, 1] / cc[:, 2] * fy + cy
                x0, x1 = int(max(0, np.floor(uu.min()) - 1)), int(min(width, np.ceil(uu.max()) + 2))
                y0, y1 = int(max(0, np.floor(vv.min()) - 1)), int(min(height, np.ceil(vv.max()) + 2))
                if x0 >= x1 or y0 >= y1:
                    continue
            idx = (np.arange(y0, y1)[:, None] * width + np.arange(x0, x1)[None, :]).reshape(-1)
            dsub, isub = d_w[idx], inv[idx]
            m = idx.shape[0]
            t1 = (lo - o) * isub
            t2 = (hi - o) * isub
            tn = np.minimum(t1, t2)
            tf = np.maximum(t1, t2)
            ax = np.argmax(tn, 1)
            tenter = tn[np.arange(m), ax]
            texit = tf.min(1)
            hit = (tenter < texit) & (tenter > 0.05) & (tenter < best_t[idx])
            side = (dsub[np.arange(m), ax] < 0).astype(np.int64)
            face = 6 * (1 + b) + ax * 2 + side
            best_t[idx] = np.where(hit, tenter, best_t[idx])
            best_face[idx] = np.where(hit, face, best_face[idx])
        P = o + d_w * best_t[:, None]
        ax = (best_face % 6) // 2
        # in-plane coordinates = the two axes other than the face normal
        u = np.where(ax == 0, P[:, 1], P[:, 0])
        v = np.where(ax == 2, P[:, 1], P[:, 2])
        per = self.period[best_face]
        ph = self.phase[best_face]
        cu = np.floor(u / per[:, 0] + ph[:, 0]).astype(np.int64)
        cv = np.floor(v / per[:, 1] + ph[:, 1]).astype(np.int64)
        checker = ((cu + cv) & 1).astype(np.float64)
        sk = self.shade_k[best_face]
        shade = 12.0 * np.sin(sk[:, 0] * u + 0.3) * np.cos(sk[:, 1] * v - 0.2)
        inten = self.base[best_face] + self.amp[best_face] * (checker - 0.5) + shade
        rng = np.random.default_rng([self.seed, noise_seed, 7])
        img = inten[:, None] * self.tint[best_face] + rng.normal(0, noise_sigma, (n, 3))
        bgr = np.clip(np.rint(img), 0, 255).astype(np.uint8).reshape(height, width, 3)
        depth = best_t.reshape(height, width).astype(np.float32)
        if hole_frac > 0:
            # holes near depth discontinuities
            gx = np.abs(np.diff(depth, axis=1, prepend=depth[:, :1]))
            gy = np.abs(np.diff(depth, axis=0, prepend=depth[:1, :]))
            disc = (gx > 0.08) | (gy > 0.08)
            near = disc.copy()
            for _ in range(2):  # dilate by 2 px (4-neighbourhood)
                nn = near.copy()
                nn[1:, :] |= near[:-1, :]
                nn[:-1, :] |= near[1:, :]
                nn[:, 1:] |= near[:, :-1]
                nn[:, :-1] |= near[:, 1:]
                near = nn
            frac_near = max(near.mean(), 1e-6)
            p = min(1.0, hole_frac / frac_near)
            holes = near & (rng.uniform(0, 1, near.shape) < p)
            depth = np.where(holes, np.float32(0), depth)
        return bgr, depth


def random_twist(rng, max_t=0.03, max_rot_deg=1.5):
    """Small hand-held-scale motion: |t| ~ U(0,max_t) m, |w| ~ U(0,max_rot) deg."""
    dt = rng.normal(size=3)
    dt *= rng.uniform(0, max_t) / np.linalg.norm(dt)
    dw = rng.normal(size=3)
    dw *= np.deg2rad(rng.uniform(0, max_rot_deg)) / np.linalg.norm(dw)
    return np.concatenate([dt, dw])


def make_pair(seed, settings, max_t=0.03, max_rot_deg=1.5, hole_frac=0.03):
    """One independent frame-pair.  Returns dict(ref=(bgr,depth), curr=(bgr,depth),
    T_ref_curr = ground-truth 4x4 mapping CURRENT-frame points into the REFERENCE frame
    (the R,T convention of TrackerNew::trackFrames)."""
    rng = np.random.default_rng([seed, 11])
    scene = Scene(seed)
    T_w_ref = se3_exp(random_twist(rng, 0.05, 3.0))
    T_ref_curr = se3_exp(random_twist(rng, max_t, max_rot_deg))
    T_w_curr = T_w_ref @ T_ref_curr
    k = (settings.width, settings.height, settings.fx, settings.fy, settings.cx, settings.cy)
    ref = scene.render(T_w_ref, *k, noise_seed=0, hole_frac=hole_frac)
    curr = scene.render(T_w_curr, *k, noise_seed=1, hole_frac=hole_frac)
    return dict(ref=ref, curr=curr, T_ref_curr=T_ref_curr, T_w_ref=T_w_ref, T_w_curr=T_w_curr)


def _frame_job(args):
    seed, T, k, i, hole_frac = args
    return Scene(seed).render(T, *k, noise_seed=i, hole_frac=hole_frac)


def make_sequence(seed, settings, n_frames, max_t=0.012, max_rot_deg=0.6, hole_frac=0.03, bias=None, workers=1):
    """Smooth random-walk trajectory (TUM-like stand-in): list of (bgr, depth, ts, T_w_c).
    bias: optional constant twist added every frame (e.g. a steady pan that forces new keyframes).
    workers > 1: the frames are rendered on that many host cores (spawned like make_pairs; the same pixels: a frame depends on
    the seed, its pose and its index only)."""
    rng = np.random.default_rng([seed, 23])
    k = (settings.width, settings.height, settings.fx, settings.fy, settings.cx, settings.cy)
    T = np.eye(4)
    vel = random_twist(rng, max_t, max_rot_deg)
    poses = []
    for i in range(n_frames):
        poses.append(T.copy())
        vel = 0.85 * vel + 0.15 * random_twist(rng, max_t, max_rot_deg)
        # pull back towards the room centre so long sequences stay inside
        vel[:3] -= 0.02 * T[:3, :3].T @ T[:3, 3] * 0.05
        T = T @ se3_exp(vel if bias is None else vel + np.asarray(bias, np.float64))
    jobs = [(seed, poses[i], k, i, hole_frac) for i in range(n_frames)]
    import os
    import sys
    main = sys.modules.get("__main__")
    if workers > 1 and os.path.isfile(getattr(main, "__file__", None) or ""):
        import multiprocessing as mp
        with mp.get_context("spawn").Pool(min(workers, n_frames)) as pool:
            imgs = pool.map(_frame_job, jobs, chunksize=max(1, n_frames // (4 * workers)))
    else:
        scene = Scene(seed)
        imgs = [scene.render(poses[i], *k, noise_seed=i, hole_frac=hole_frac) for i in range(n_frames)]
    return [(imgs[i][0], imgs[i][1], 1305031102.0 + i / 30.0, poses[i]) for i in range(n_frames)]


def ate_rmse(est, gt):
    """Absolute trajectory error (RMSE, metres) after Horn/Umeyama rigid alignment
    of the estimated positions onto ground truth (TUM evaluate_ate.py semantics)."""
    P = np.asarray([T[:3, 3] for T in est], np.float64)
    Q = np.asarray([T[:3, 3] for T in gt], np.float64)
    mp, mq = P.mean(0), Q.mean(0)
    H = (P - mp).T @ (Q - mq)
    U, _, Vt = np.linalg.svd(H)
    D = np.diag([1, 1, np.sign(np.linalg.det(Vt.T @ U.T))])
    R = Vt.T @ D @ U.T
    t = mq - R @ mp
    err = (P @ R.T + t) - Q
    return float(np.sqrt((err ** 2).sum(1).mean()))


def rpe_rmse(est, gt, delta=1):
    """Relative pose error over `delta` frames (TUM evaluate_rpe.py semantics, fixed frame delta):
    E_i = (Q_i^-1 Q_{i+d})^-1 (P_i^-1 P_{i+d}); returns (translational RMSE in metres, rotational
    RMSE in rad)."""
    tr, ro = [], []
    for i in range(len(est) - delta):
        Pd = np.linalg.inv(np.asarray(est[i], np.float64)) @ np.asarray(est[i + delta], np.float64)
        Qd = np.linalg.inv(np.asarray(gt[i], np.float64)) @ np.asarray(gt[i + delta], np.float64)
        E = np.linalg.inv(Qd) @ Pd
        tr.append(float(np.linalg.norm(E[:3, 3])))
        ro.append(rot_angle(np.eye(3), E[:3, :3]))
    if not tr:
        return 0.0, 0.0
    return float(np.sqrt(np.mean(np.square(tr)))), float(np.sqrt(np.mean(np.square(ro))))


def rot_angle(Ra, Rb):
    """Angle (rad) of Ra^T Rb via the skew part (|sin| of the angle): accurate for the tiny
    angles compared here, where arccos((trace-1)/2) has a ~3e-4 rad float32 noise floor."""
    d = np.asarray(Ra, np.float64).T @ np.asarray(Rb, np.float64)
    w = 0.5 * np.array([d[2, 1] - d[1, 2], d[0, 2] - d[2, 0], d[1, 0] - d[0, 1]])
    s = float(np.linalg.norm(w))
    c = (np.trace(d) - 1.0) / 2.0
    return float(np.arctan2(s, c)) if s > 1e-3 else s


def pose_error(R_est, T_est, T_gt):
    """(rotation error rad, translation error m) of (R,T) vs a 4x4 ground truth."""
    return rot_angle(R_est, T_gt[:3, :3]), float(np.linalg.norm(np.asarray(T_est, np.float64) - T_gt[:3, 3]))


def _pair_job(args):
    seed, w, h, fx, fy, cx, cy = args

    class _K:  # the camera fields make_pair reads
        pass
    k = _K()
    k.width, k.height, k.fx, k.fy, k.cx, k.cy = w, h, fx, fy, cx, cy
    return make_pair(seed, k)


def make_pairs(seeds, settings, workers=None):
    """make_pair for many seeds on several host cores (rendering a 640x480 pair takes ~0.3 s).  The workers are SPAWNED,
    not forked: the calling process may hold an initialised HIP runtime (GPU tests), and this module imports numpy only."""
    import multiprocessing as mp
    import os
    seeds = list(seeds)
    jobs = [(sd, settings.width, settings.height, settings.fx, settings.fy, settings.cx, settings.cy) for sd in seeds]
    if workers is None:
        try:
            workers = len(os.sched_getaffinity(0))
        except AttributeError:
            workers = os.cpu_count() or 1
        workers = max(1, min(16, workers, len(jobs)))
    import sys
    main = sys.modules.get("__main__")
    if workers <= 1 or not os.path.isfile(getattr(main, "__file__", None) or ""):
        # (spawned workers re-import __main__: a script fed through stdin or `python -c` has no file to import and the pool
        # would wait for ever -- render serially there)
        return [_pair_job(j) for j in jobs]
    with mp.get_context("spawn").Pool(workers) as pool:
        return pool.map(_pair_job, jobs, chunksize=max(1, len(jobs) // (4 * workers)))
