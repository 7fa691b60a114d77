"""ctypes wrapper of the CPU oracle (oracle/librevo_oracle.so).

TEST INFRASTRUCTURE ONLY: imported by tests/, __graft_entry__.smoke() and
bench.py's cpu_baseline leg.  The product package (revo_amd/) never imports it.
"""
import ctypes as C
import os
import subprocess

import numpy as np

from revo_amd.settings import (ImgPyramidSettings, OptimizerSettings, ResidualInfo,
                               TrackerSettings, MAX_LEVELS, PLANE_GRAY, PLANE_DEPTH,
                               PLANE_EDGES, PLANE_EDGES_ORIG, PLANE_DT, PLANE_GRADTABLE,
                               PLANE_EDGES3D, PLANE_HIST)

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "librevo_oracle.so")


def build(force=False):
    src = [os.path.join(_HERE, f) for f in ("revo_oracle.c", "revo_oracle.h")]
    src.append(os.path.join(_HERE, "..", "include", "revo_hip.h"))
    stale = (not os.path.exists(_SO)) or any(
        os.path.getmtime(s) > os.path.getmtime(_SO) for s in src if os.path.exists(s))
    if force or stale:
        subprocess.check_call(["make", "-C", _HERE, "-B", "librevo_oracle.so"],
                              stdout=subprocess.DEVNULL)
    return _SO


_lib = None
u8p = C.POINTER(C.c_uint8)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)


def _p(a, t):
    return a.ctypes.data_as(t)


def lib():
    global _lib
    if _lib is None:
        build()
        L = C.CDLL(_SO)
        L.ro_pyramid_create.restype = C.c_void_p
        L.ro_pyramid_create.argtypes = [C.POINTER(ImgPyramidSettings), u8p, C.c_size_t, f32p,
                                        C.c_size_t, C.c_double]
        L.ro_pyramid_destroy.argtypes = [C.c_void_p]
        L.ro_pyramid_make_keyframe.argtypes = [C.c_void_p]
        L.ro_pyramid_is_keyframe.argtypes = [C.c_void_p]
        L.ro_pyramid_read.restype = C.c_size_t
        L.ro_pyramid_read.argtypes = [C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_size_t]
        L.ro_pyramid_camera.argtypes = [C.c_void_p, C.c_int, f32p]
        L.ro_pyramid_colored_pcl.restype = C.c_size_t
        L.ro_pyramid_colored_pcl.argtypes = [C.c_void_p, C.c_int, C.c_int, f32p, C.c_size_t]
        L.ro_tracker_create.restype = C.c_void_p
        L.ro_tracker_create.argtypes = [C.POINTER(ImgPyramidSettings), C.POINTER(OptimizerSettings),
                                        C.POINTER(TrackerSettings)]
        L.ro_tracker_destroy.argtypes = [C.c_void_p]
        L.ro_optimizer_eval.restype = C.c_float
        L.ro_optimizer_eval.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, f32p, f32p, C.c_int,
                                        C.POINTER(ResidualInfo), f32p, f32p, f32p]
        L.ro_optimizer_track_level.restype = C.c_float
        L.ro_optimizer_track_level.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, f32p, f32p,
                                               C.c_int, C.POINTER(ResidualInfo),
                                               C.POINTER(C.c_int), C.POINTER(C.c_int)]
        L.ro_tracker_eval_cost.restype = C.c_float
        L.ro_tracker_eval_cost.argtypes = [C.c_void_p, f32p, f32p, C.c_int, C.c_void_p, C.c_void_p]
        L.ro_tracker_track_frames.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p, f32p, f32p, f32p,
                                              C.POINTER(ResidualInfo), i32p, C.POINTER(C.c_int)]
        L.ro_tracker_assess_quality.argtypes = [C.c_void_p, f32p, C.c_void_p, i32p, i32p]
        L.ro_tracker_add_old_pcl.argtypes = [C.c_void_p, C.c_void_p, C.c_int, f32p, C.c_double]
        L.ro_tracker_clear_past.argtypes = [C.c_void_p]
        L.ro_tracker_past_size.argtypes = [C.c_void_p]
        L.ro_vo_create.restype = C.c_void_p
        L.ro_vo_create.argtypes = [C.POINTER(ImgPyramidSettings), C.POINTER(OptimizerSettings),
                                   C.POINTER(TrackerSettings)]
        L.ro_vo_destroy.argtypes = [C.c_void_p]
        L.ro_vo_push.argtypes = [C.c_void_p, u8p, C.c_size_t, f32p, C.c_size_t, C.c_double, f32p]
        L.ro_vo_num_keyframes.argtypes = [C.c_void_p]
        L.ro_vo_times.argtypes = [C.c_void_p, C.POINTER(C.c_double)]
        L.ro_ldlt6_solve.argtypes = [f32p, f32p, f32p]
        L.ro_se3_exp.argtypes = [f32p, f32p, f32p]
        L.ro_se3_mul.argtypes = [f32p] * 6
        L.ro_quat_from_R.argtypes = [f32p, f32p]
        L.ro_quat_to_R.argtypes = [f32p, f32p]
        L.ro_is_orthogonal.argtypes = [f32p]
        L.ro_mat4_inverse.argtypes = [f32p, f32p]
        L.ro_canny.argtypes = [u8p, C.c_int, C.c_int, C.c_double, C.c_double, u8p]
        L.ro_sobel3.argtypes = [u8p, C.c_int, C.c_int, C.POINTER(C.c_int16), C.POINTER(C.c_int16)]
        L.ro_bgr2gray.argtypes = [u8p, C.c_size_t, C.c_int, C.c_int, u8p]
        L.ro_pyrdown_u8.argtypes = [u8p, C.c_int, C.c_int, u8p]
        L.ro_depth_subsample.argtypes = [f32p, C.c_int, C.c_int, f32p]
        L.ro_edt.argtypes = [u8p, C.c_int, C.c_int, f32p]
        L.ro_grad_table.argtypes = [f32p, C.c_int, C.c_int, f32p]
        L.ro_dist_histogram.restype = C.c_float
        L.ro_dist_histogram.argtypes = [u8p, C.c_int, C.c_int, C.c_int, u8p]
        L.ro_fill_in_edges.argtypes = [u8p, C.c_int, u8p, C.c_int, C.c_int, C.c_int, C.c_int, u8p, C.c_int]
        L.ro_edges3d.argtypes = [u8p, f32p, C.c_int, C.c_int] + [C.c_float] * 6 + [f32p]
        L.ro_u16_to_depth.argtypes = [C.POINTER(C.c_uint16), C.c_size_t, C.c_int, C.c_int, C.c_double, f32p]
        L.ro_set_accum_double.argtypes = [C.c_int]
        L.ro_set_ab_ulp_noise.argtypes = [C.c_uint]
        L.ro_vo_run_pipelined.restype = C.c_double
        L.ro_vo_run_pipelined.argtypes = [C.c_void_p, C.c_int, u8p, f32p, C.POINTER(C.c_double), C.c_int, C.c_int, C.c_int, f32p]
        L.ro_bench_pairs_pipelined.restype = None
        L.ro_bench_pairs_pipelined.argtypes = [C.POINTER(ImgPyramidSettings), C.POINTER(OptimizerSettings),
                                               C.POINTER(TrackerSettings), u8p, f32p, C.c_int, C.c_int, C.c_int, C.c_int,
                                               C.POINTER(C.c_double)]
        L.ro_bench_pairs_mt.restype = C.c_long
        L.ro_bench_pairs_mt.argtypes = [C.POINTER(ImgPyramidSettings), C.POINTER(OptimizerSettings),
                                        C.POINTER(TrackerSettings), u8p, f32p, C.c_int, C.c_int, C.c_double,
                                        C.POINTER(C.c_double)]
        _lib = L
    return _lib


# ---- primitives (numpy in / numpy out) -------------------------------------
def bgr2gray(bgr):
    bgr = np.ascontiguousarray(bgr, np.uint8)
    h, w, _ = bgr.shape
    out = np.empty((h, w), np.uint8)
    lib().ro_bgr2gray(_p(bgr, u8p), w * 3, w, h, _p(out, u8p))
    return out


def pyrdown(gray):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.empty((h // 2, w // 2), np.uint8)
    lib().ro_pyrdown_u8(_p(gray, u8p), w, h, _p(out, u8p))
    return out


def depth_subsample(d):
    d = np.ascontiguousarray(d, np.float32)
    h, w = d.shape
    out = np.empty((h // 2, w // 2), np.float32)
    lib().ro_depth_subsample(_p(d, f32p), w, h, _p(out, f32p))
    return out


def sobel3(gray):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    dx = np.empty((h, w), np.int16)
    dy = np.empty((h, w), np.int16)
    lib().ro_sobel3(_p(gray, u8p), w, h, _p(dx, C.POINTER(C.c_int16)), _p(dy, C.POINTER(C.c_int16)))
    return dx, dy


def canny(gray, t1=150, t2=100):
    gray = np.ascontiguousarray(gray, np.uint8)
    h, w = gray.shape
    out = np.empty((h, w), np.uint8)
    lib().ro_canny(_p(gray, u8p), w, h, float(t1), float(t2), _p(out, u8p))
    return out


def edt(edges):
    edges = np.ascontiguousarray(edges, np.uint8)
    h, w = edges.shape
    out = np.empty((h, w), np.float32)
    lib().ro_edt(_p(edges, u8p), w, h, _p(out, f32p))
    return out


def grad_table(dt):
    dt = np.ascontiguousarray(dt, np.float32)
    h, w = dt.shape
    out = np.empty((h, w, 4), np.float32)
    lib().ro_grad_table(_p(dt, f32p), w, h, _p(out, f32p))
    return out


def dist_histogram(edges, patch):
    edges = np.ascontiguousarray(edges, np.uint8)
    h, w = edges.shape
    hist = np.empty((h // patch, w // patch), np.uint8)
    frac = lib().ro_dist_histogram(_p(edges, u8p), w, h, patch, _p(hist, u8p))
    return hist, frac


def edges3d(edges, depth, fx, fy, cx, cy, dmin=0.1, dmax=5.2):
    edges = np.ascontiguousarray(edges, np.uint8)
    depth = np.ascontiguousarray(depth, np.float32)
    h, w = edges.shape
    out = np.empty((h * w, 4), np.float32)
    n = lib().ro_edges3d(_p(edges, u8p), _p(depth, f32p), w, h, fx, fy, cx, cy, dmin, dmax, _p(out, f32p))
    return out[:n].copy()


def u16_to_depth(raw, scale):
    raw = np.ascontiguousarray(raw, np.uint16)
    h, w = raw.shape
    out = np.empty((h, w), np.float32)
    lib().ro_u16_to_depth(_p(raw, C.POINTER(C.c_uint16)), w * 2, w, h, float(scale), _p(out, f32p))
    return out


def ldlt6_solve(A, b):
    A = np.ascontiguousarray(A, np.float32)
    b = np.ascontiguousarray(b, np.float32)
    x = np.empty(6, np.float32)
    lib().ro_ldlt6_solve(_p(A, f32p), _p(b, f32p), _p(x, f32p))
    return x


def se3_exp(a):
    a = np.ascontiguousarray(a, np.float32)
    q = np.empty(4, np.float32)
    t = np.empty(3, np.float32)
    lib().ro_se3_exp(_p(a, f32p), _p(q, f32p), _p(t, f32p))
    return q, t


def quat_to_R(q):
    q = np.ascontiguousarray(q, np.float32)
    R = np.empty(9, np.float32)
    lib().ro_quat_to_R(_p(q, f32p), _p(R, f32p))
    return R.reshape(3, 3).T.copy()  # column-major -> numpy row-major


def quat_from_R(R):
    Rc = np.ascontiguousarray(np.asarray(R, np.float32).T)  # to column-major storage
    q = np.empty(4, np.float32)
    lib().ro_quat_from_R(_p(Rc, f32p), _p(q, f32p))
    return q


def se3_mul(qa, ta, qb, tb):
    args = [np.ascontiguousarray(x, np.float32) for x in (qa, ta, qb, tb)]
    qo = np.empty(4, np.float32)
    to = np.empty(3, np.float32)
    lib().ro_se3_mul(*[_p(x, f32p) for x in args], _p(qo, f32p), _p(to, f32p))
    return qo, to


def mat4_inverse(M):
    Mc = np.ascontiguousarray(np.asarray(M, np.float32).T)
    out = np.empty(16, np.float32)
    lib().ro_mat4_inverse(_p(Mc, f32p), _p(out, f32p))
    return out.reshape(4, 4).T.copy()


def _cm3(R):
    """numpy 3x3 (row-major semantics) -> column-major float32[9]"""
    return np.ascontiguousarray(np.asarray(R, np.float32).T).reshape(9)


def _cm4(M):
    return np.ascontiguousarray(np.asarray(M, np.float32).T).reshape(16)


_ESZ = {PLANE_GRAY: (np.uint8, 1), PLANE_DEPTH: (np.float32, 1), PLANE_EDGES: (np.uint8, 1),
        PLANE_EDGES_ORIG: (np.uint8, 1), PLANE_DT: (np.float32, 1), PLANE_GRADTABLE: (np.float32, 4),
        PLANE_EDGES3D: (np.float32, 4), PLANE_HIST: (np.uint8, 1)}


class Pyramid:
    """ImgPyramidRGBD (oracle)."""

    def __init__(self, settings, bgr, depth, ts=0.0):
        self.s = settings
        bgr = np.ascontiguousarray(bgr, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        h, w = depth.shape
        assert bgr.shape == (h, w, 3) and (w, h) == (settings.width, settings.height)
        self.h = lib().ro_pyramid_create(C.byref(settings), _p(bgr, u8p), w * 3, _p(depth, f32p), w * 4, ts)

    def __del__(self):
        if getattr(self, "h", None):
            lib().ro_pyramid_destroy(self.h)
            self.h = None

    def makeKeyframe(self):
        lib().ro_pyramid_make_keyframe(self.h)

    def read(self, what, lvl):
        dt, k = _ESZ[what]
        n = lib().ro_pyramid_read(self.h, what, lvl, None, 0)
        w, h = self.s.level_size(lvl)
        buf = np.empty((n, k) if k > 1 else (n,), dt)
        lib().ro_pyramid_read(self.h, what, lvl, buf.ctypes.data_as(C.c_void_p), buf.nbytes)
        if what in (PLANE_EDGES3D,):
            return buf
        if what == PLANE_HIST:
            p = self.s.hist_patch[lvl]
            return buf.reshape(h // p, w // p) if p > 0 and n else buf
        if n == 0:
            return buf
        return buf.reshape((h, w, 4) if k > 1 else (h, w))

    def camera(self, lvl):
        out = np.empty(6, np.float32)
        lib().ro_pyramid_camera(self.h, lvl, _p(out, f32p))
        return out

    def generateColoredPcl(self, lvl, dense):
        n = lib().ro_pyramid_colored_pcl(self.h, lvl, int(dense), None, 0)
        out = np.empty((n, 8), np.float32)
        lib().ro_pyramid_colored_pcl(self.h, lvl, int(dense), _p(out, f32p), n)
        return out


class Tracker:
    """TrackerNew + Optimizer (oracle)."""

    def __init__(self, ps, os_=None, ts=None):
        self.ps, self.os, self.ts = ps, os_ or OptimizerSettings(), ts or TrackerSettings()
        self.h = lib().ro_tracker_create(C.byref(self.ps), C.byref(self.os), C.byref(self.ts))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ro_tracker_destroy(self.h)
            self.h = None

    def eval(self, ref, curr, R, T, lvl):
        Rc, Tc = _cm3(R), np.ascontiguousarray(T, np.float32)
        info = ResidualInfo()
        A = np.empty(36, np.float32)
        b = np.empty(6, np.float32)
        e = C.c_float()
        err = lib().ro_optimizer_eval(self.h, ref.h, curr.h, _p(Rc, f32p), _p(Tc, f32p), lvl,
                                      C.byref(info), _p(A, f32p), _p(b, f32p), C.byref(e))
        return err, info, A.reshape(6, 6), b

    def track_level(self, ref, curr, R, T, lvl):
        Rc, Tc = _cm3(R), np.array(T, np.float32)
        info = ResidualInfo()
        ev, ab = C.c_int(), C.c_int()
        err = lib().ro_optimizer_track_level(self.h, ref.h, curr.h, _p(Rc, f32p), _p(Tc, f32p), lvl,
                                             C.byref(info), C.byref(ev), C.byref(ab))
        return Rc.reshape(3, 3).T.copy(), Tc, err, info, ev.value, ab.value

    def eval_cost(self, R, T, lvl, curr, ref):
        Rc, Tc = _cm3(R), np.ascontiguousarray(T, np.float32)
        return lib().ro_tracker_eval_cost(self.h, _p(Rc, f32p), _p(Tc, f32p), lvl, curr.h, ref.h)

    def trackFrames(self, ref, curr, R, T):
        Rc, Tc = _cm3(R), np.array(T, np.float32)
        info = ResidualInfo()
        err = C.c_float()
        evals = np.zeros(MAX_LEVELS, np.int32)
        flags = C.c_int()
        status = lib().ro_tracker_track_frames(self.h, ref.h, curr.h, _p(Rc, f32p), _p(Tc, f32p),
                                               C.byref(err), C.byref(info), _p(evals, i32p), C.byref(flags))
        return dict(R=Rc.reshape(3, 3).T.copy(), T=Tc, err=err.value, status=status, info=info,
                    evals=evals, flags=flags.value)

    def assessTrackingQuality(self, T_w_curr, curr):
        Mc = _cm4(T_w_curr)
        h4 = np.zeros(4, np.int32)
        o4 = np.zeros(4, np.int32)
        st = lib().ro_tracker_assess_quality(self.h, _p(Mc, f32p), curr.h, _p(h4, i32p), _p(o4, i32p))
        return st, h4, o4

    def addOldPclAndPose(self, src, lvl, T_w, ts=0.0):
        Mc = _cm4(T_w)
        lib().ro_tracker_add_old_pcl(self.h, src.h, lvl, _p(Mc, f32p), ts)

    def clearUpPastLists(self):
        lib().ro_tracker_clear_past(self.h)

    def past_size(self):
        return lib().ro_tracker_past_size(self.h)


class VO:
    """REVO::start sequencing (oracle)."""

    def __init__(self, ps, os_=None, ts=None):
        self.ps, self.os, self.ts = ps, os_ or OptimizerSettings(), ts or TrackerSettings()
        self.h = lib().ro_vo_create(C.byref(self.ps), C.byref(self.os), C.byref(self.ts))

    def __del__(self):
        if getattr(self, "h", None):
            lib().ro_vo_destroy(self.h)
            self.h = None

    def push(self, bgr, depth, ts):
        bgr = np.ascontiguousarray(bgr, np.uint8)
        depth = np.ascontiguousarray(depth, np.float32)
        h, w = depth.shape
        pose = np.empty(16, np.float32)
        kf = lib().ro_vo_push(self.h, _p(bgr, u8p), w * 3, _p(depth, f32p), w * 4, ts, _p(pose, f32p))
        return pose.reshape(4, 4).T.copy(), kf

    def run_pipelined(self, bgr_frames, depth_frames, timestamps=None, cpu_io=-1, cpu_main=-1, queue_cap=4):
        """REVO::start with the reference's IO thread (system.cpp:96): pyramids on one pinned core, keyframes +
        tracking on another.  frames [n,H,W,3] u8 / [n,H,W] f32 -> (wall seconds, poses [n,4,4])."""
        bgr = np.ascontiguousarray(bgr_frames, np.uint8)
        dep = np.ascontiguousarray(depth_frames, np.float32)
        n = bgr.shape[0]
        ts = np.ascontiguousarray(timestamps if timestamps is not None else np.arange(n), np.float64)
        poses = np.empty((n, 16), np.float32)
        dt = lib().ro_vo_run_pipelined(self.h, n, _p(bgr, u8p), _p(dep, f32p), ts.ctypes.data_as(C.POINTER(C.c_double)),
                                       int(cpu_io), int(cpu_main), int(queue_cap), _p(poses, f32p))
        return dt, poses.reshape(n, 4, 4).transpose(0, 2, 1).copy()

    def num_keyframes(self):
        return lib().ro_vo_num_keyframes(self.h)

    def times(self):
        out = (C.c_double * 3)()
        lib().ro_vo_times(self.h, out)
        return tuple(out)


def bench_pairs_pipelined(ps, bgr_frames, depth_frames, cpu_io, cpu_main, passes, os_=None, ts=None):
    """The reference's two threads on the batch workload (frames packed [ref0, curr0, ref1, ...]): the IO thread builds
    pyramids, the main thread runs makeKeyframe + trackFrames; both pinned.  -> seconds per pass."""
    bgr = np.ascontiguousarray(bgr_frames, np.uint8)
    dep = np.ascontiguousarray(depth_frames, np.float32)
    os_, ts = os_ or OptimizerSettings(), ts or TrackerSettings()
    out = (C.c_double * passes)()
    lib().ro_bench_pairs_pipelined(C.byref(ps), C.byref(os_), C.byref(ts), _p(bgr, u8p), _p(dep, f32p), bgr.shape[0] // 2,
                                   int(cpu_io), int(cpu_main), int(passes), out)
    return [float(x) for x in out]


def bench_pairs_mt(ps, bgr_frames, depth_frames, n_threads, seconds, os_=None, ts=None):
    """All-core CPU baseline (native pthreads): frames packed [ref0, curr0, ref1, ...] -> (pairs done, elapsed s)."""
    bgr = np.ascontiguousarray(bgr_frames, np.uint8)
    dep = np.ascontiguousarray(depth_frames, np.float32)
    el = C.c_double()
    os_, ts = os_ or OptimizerSettings(), ts or TrackerSettings()
    n = lib().ro_bench_pairs_mt(C.byref(ps), C.byref(os_), C.byref(ts), _p(bgr, u8p), _p(dep, f32p), bgr.shape[0] // 2,
                                int(n_threads), float(seconds), C.byref(el))
    return int(n), el.value
