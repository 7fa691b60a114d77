"""Host-side mirror of the reference's hot-path classes over the C ABI.

Same names, argument meaning and error behaviour as the reference so the parity
tests read like tests of the reference itself:

  CameraPyr        datastructures/camerapyr.h:113-193   (owns the device context)
  ImgPyramidRGBD   datastructures/imgpyramidrgbd.h:27-250
  Optimizer        system/optimizer.h:114-186
  TrackerNew       system/tracker.h:56-105
  BatchTracker     new: B independent frame-pairs resident in HBM

numpy matrices are ordinary row-major views (R[i, j]); the column-major
conversion the C ABI wants (Eigen storage) happens here.
"""
import ctypes as C

import numpy as np

from . import _lib
from ._lib import RevoError, check, f32p, i32p, u8p, u16p, vp
from .settings import (ImgPyramidSettings, OptimizerSettings, TrackerSettings, ResidualInfo, PairResult, PairIn,
                       MAX_LEVELS, PLANE_GRAY, PLANE_DEPTH, PLANE_EDGES, PLANE_EDGES_ORIG, PLANE_DT,
                       PLANE_GRADTABLE, PLANE_EDGES3D, PLANE_HIST, PLANE_EDGES3D_TILED, TRACKER_STATE_OK, TRACKER_STATE_NEW_KF)


def _p(a, t):
    return a.ctypes.data_as(t)


def _cm3(R):
    return np.ascontiguousarray(np.asarray(R, np.float32).T).reshape(9)


def _cm4(M):
    return np.ascontiguousarray(np.asarray(M, np.float32).T).reshape(16)


class Camera:
    """camerapyr.h:90-111"""

    def __init__(self, fx, fy, cx, cy, width, height):
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.width, self.height = int(width), int(height)
        self.area = self.width * self.height

    def returnSize(self):
        return (self.width, self.height)


class CameraPyr:
    """camerapyr.h:113-193.  Also owns the HIP context (device, stream, HBM pools)."""

    def __init__(self, settingsPyr, device=0, optimizerSettings=None, trackerSettings=None):
        self.settings = settingsPyr
        self._h = vp()
        opt = optimizerSettings or OptimizerSettings()
        trk = trackerSettings or TrackerSettings()
        check(_lib.lib().revo_ctx_create(device, C.byref(settingsPyr), C.byref(opt), C.byref(trk), C.byref(self._h)))
        self.camPyr = []
        for lvl in range(settingsPyr.nLevels()):
            out = np.empty(6, np.float32)
            check(_lib.lib().revo_ctx_camera(self._h, lvl, _p(out, f32p)))
            self.camPyr.append(Camera(*[float(x) for x in out[:4]], out[4], out[5]))

    def size(self):
        return len(self.camPyr)

    def at(self, lvl):
        return self.camPyr[lvl]

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().revo_ctx_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


_DT = {PLANE_GRAY: (np.uint8, 1), PLANE_DEPTH: (np.float32, 1), PLANE_EDGES: (np.uint8, 1),
       PLANE_EDGES_ORIG: (np.uint8, 1), PLANE_DT: (np.float32, 1), PLANE_GRADTABLE: (np.float32, 4),
       PLANE_EDGES3D: (np.float32, 4), PLANE_HIST: (np.uint8, 1), PLANE_EDGES3D_TILED: (np.float32, 4)}


class ImgPyramidRGBD:
    """imgpyramidrgbd.h:27-250.  fullResRgb is BGR8 [H,W,3], fullResDepth float32 metres [H,W]
    (or uint16 raw + depth_scale_factor, fusing iowrapperRGBD.cpp:326-327)."""

    def __init__(self, settings, cameraPyr, fullResRgb=None, fullResDepth=None, timestamp=0.0,
                 depth_scale_factor=None, _handle=None, _owned=True):
        self.mSettings = settings
        self.cameraPyr = cameraPyr
        self.frameId = 0
        self._owned = _owned
        self._T_w_f = np.eye(4, dtype=np.float32)
        if _handle is not None:
            self._h = _handle
            return
        bgr = np.ascontiguousarray(fullResRgb, np.uint8)
        h, w = bgr.shape[:2]
        if bgr.shape != (h, w, 3) or np.shape(fullResDepth) != (h, w) or (w, h) != (settings.width, settings.height):
            raise ValueError("image size does not match the settings")
        self._h = vp()
        L = _lib.lib()
        if depth_scale_factor is not None:
            d = np.ascontiguousarray(fullResDepth, np.uint16)
            check(L.revo_pyramid_create_u16(cameraPyr._h, _p(bgr, u8p), w * 3, _p(d, u16p), w * 2,
                                            float(depth_scale_factor), float(timestamp), C.byref(self._h)))
        else:
            d = np.ascontiguousarray(fullResDepth, np.float32)
            check(L.revo_pyramid_create(cameraPyr._h, _p(bgr, u8p), w * 3, _p(d, f32p), w * 4, float(timestamp),
                                        C.byref(self._h)))

    def __del__(self):
        try:
            if self._owned and getattr(self, "_h", None):
                _lib.lib().revo_pyramid_destroy(self._h)
                self._h = None
        except Exception:
            pass

    # -- keyframe promotion (imgpyramidrgbd.cpp:231-252)
    def makeKeyframe(self):
        check(_lib.lib().revo_pyramid_make_keyframe(self._h))

    def _read(self, what, lvl):
        dt, k = _DT[what]
        n = C.c_size_t()
        check(_lib.lib().revo_pyramid_read(self._h, what, lvl, None, 0, C.byref(n)))
        buf = np.empty((n.value, k) if k > 1 else (n.value,), dt)
        if n.value:
            check(_lib.lib().revo_pyramid_read(self._h, what, lvl, buf.ctypes.data_as(vp), buf.nbytes, C.byref(n)))
        w, h = self.mSettings.level_size(lvl)
        if what in (PLANE_EDGES3D, PLANE_EDGES3D_TILED) or n.value == 0:
            return buf
        if what == PLANE_HIST:
            p = self.mSettings.hist_patch[lvl]
            return buf.reshape(h // p, w // p)
        return buf.reshape((h, w, 4) if k > 1 else (h, w))

    # -- accessors (imgpyramidrgbd.h:45-117)
    def returnK(self, lvl):
        cam = self.cameraPyr.at(lvl)
        K = np.eye(3, dtype=np.float32)
        K[0, 0], K[1, 1], K[0, 2], K[1, 2] = cam.fx, cam.fy, cam.cx, cam.cy
        return K

    def returnDistTransform(self, lvl):
        return self._read(PLANE_DT, lvl)

    def returnEdges(self, lvl):
        return self._read(PLANE_EDGES, lvl)

    def returnOrigEdges(self, lvl):
        return self._read(PLANE_EDGES_ORIG, lvl)

    def return3DEdges(self, lvl):
        """N x 4 float32 rows (X,Y,Z,1) == the columns of the reference's 4xN Eigen::MatrixXf."""
        return self._read(PLANE_EDGES3D, lvl)

    def edges3DTiled(self, lvl):
        """The same N points as return3DEdges in the order the tracker reads them: 32x32-pixel tiles in raster order,
        row-major inside a tile (what the per-frame build writes; not a reference accessor)."""
        return self._read(PLANE_EDGES3D_TILED, lvl)

    def generateColoredPcl(self, lvl, densePcl=False):
        """imgpyramidrgbd.cpp:279-327: N x 8 float32 rows (X,Y,Z,1,r,g,b,1), colours in [0,1] ==
        the columns of the reference's 8xN clrPcl; densePcl: every usable-depth pixel, else edges."""
        w, h = self.mSettings.level_size(lvl)
        buf = np.empty((w * h, 8), np.float32)
        n = C.c_size_t()
        check(_lib.lib().revo_pyramid_colored_pcl(self._h, lvl, int(bool(densePcl)), buf.ctypes.data_as(_lib.f32p),
                                                  w * h, C.byref(n)))
        return buf[:n.value].copy()

    def returnDepth(self, lvl):
        return self._read(PLANE_DEPTH, lvl)

    def returnGray(self, lvl):
        return self._read(PLANE_GRAY, lvl)

    def returnOptimizationStructure(self, lvl):
        return self._read(PLANE_GRADTABLE, lvl)

    def returnHist(self, lvl):
        return self._read(PLANE_HIST, lvl)

    def returnTimestamp(self):
        return _lib.lib().revo_pyramid_timestamp(self._h)

    def returnMaxLvl(self):
        return self.mSettings.pyr_max_lvl

    def returnMinLvl(self):
        return self.mSettings.pyr_min_lvl

    def isKeyframe(self):
        return bool(_lib.lib().revo_pyramid_is_keyframe(self._h))

    # -- pose bookkeeping (imgpyramidrgbd.h:126-151): plain host state
    def setTwf(self, T):
        self._T_w_f = np.array(T, np.float32).reshape(4, 4)

    def getTransKFtoWorld(self):
        return self._T_w_f

    def prepareKfForStorage(self):  # imgpyramidrgbd.h:156-169: effectively a no-op in the reference
        return None


class Optimizer:
    """system/optimizer.h:114-186 (the LM loop of one level runs on the device)."""

    ResidualInfo = ResidualInfo

    def __init__(self, settings, cameraPyr):
        self.mSettings = settings
        self._cam = cameraPyr

    def trackFrames(self, refFrame, currFrame, R, T, lvl, resInfo=None):
        """-> (last_residual, R, T): optimizer.cpp:235-311."""
        Rc, Tc = _cm3(R), np.array(T, np.float32).reshape(3)
        info = resInfo if resInfo is not None else ResidualInfo()
        err = C.c_float()
        check(_lib.lib().revo_optimizer_track_level(self._cam._h, refFrame._h, currFrame._h, _p(Rc, f32p),
                                                    _p(Tc, f32p), lvl, C.byref(info), C.byref(err)))
        return err.value, Rc.reshape(3, 3).T.copy(), Tc

    def evalAt(self, refFrame, currFrame, R, T, lvl):
        """calcErrorAndBuffers + calculateWarpUpdate at a fixed pose -> (err, info, A[6,6], b[6])."""
        Rc, Tc = _cm3(R), np.ascontiguousarray(T, np.float32).reshape(3)
        info = ResidualInfo()
        err = C.c_float()
        A = np.empty(36, np.float32)
        b = np.empty(6, np.float32)
        check(_lib.lib().revo_optimizer_eval(self._cam._h, refFrame._h, currFrame._h, _p(Rc, f32p), _p(Tc, f32p), lvl,
                                             C.byref(info), C.byref(err), _p(A, f32p), _p(b, f32p)))
        return err.value, info, A.reshape(6, 6), b


def solve6(cameraPyr, A, b, lam):
    """A.ldlt().solve(b) with A(i,i) *= 1 + lam (optimizer.cpp:258-262) on the device.
    A [n,6,6] symmetric, b [n,6], lam [n] -> x [n,6]."""
    A = np.asarray(A, np.float32).reshape(-1, 36)
    n = A.shape[0]
    buf = np.concatenate([A, np.asarray(b, np.float32).reshape(n, 6), np.asarray(lam, np.float32).reshape(n, 1)], axis=1)
    buf = np.ascontiguousarray(buf, np.float32)
    x = np.empty((n, 6), np.float32)
    check(_lib.lib().revo_optimizer_solve6(cameraPyr._h, n, _p(buf, f32p), _p(x, f32p)))
    return x


class TrackerNew:
    """system/tracker.h:56-105."""

    TRACKER_STATE_OK, TRACKER_STATE_LOST, TRACKER_STATE_NEW_KF, TRACKER_STATE_UNKNOWN = 0, 1, 2, 3

    def __init__(self, config, pyrConfig, cameraPyr):
        self.mSettings = config
        self.mPyrConfig = pyrConfig
        self._cam = cameraPyr
        self.optimizerSettings = getattr(config, "optimizerSettings", None) or OptimizerSettings()
        check(_lib.lib().revo_ctx_set_tracker(cameraPyr._h, C.byref(self.optimizerSettings), C.byref(config)))
        self.histogramLevel = config.histogram_level
        self.mOptimizer = Optimizer(self.optimizerSettings, cameraPyr)
        self.last_evals = np.zeros(MAX_LEVELS, np.int32)
        self.last_info = ResidualInfo()

    def trackFrames(self, R, T, refFrame, currFrame):
        """-> (status, R, T, error): tracker.cpp:294-353."""
        Rc, Tc = _cm3(R), np.array(T, np.float32).reshape(3)
        err, status = C.c_float(), C.c_int()
        evals = np.zeros(MAX_LEVELS, np.int32)
        info = ResidualInfo()
        check(_lib.lib().revo_tracker_track_frames(self._cam._h, refFrame._h, currFrame._h, _p(Rc, f32p), _p(Tc, f32p),
                                                   C.byref(err), C.byref(status), C.byref(info), _p(evals, i32p)))
        self.last_evals, self.last_info = evals, info
        return status.value, Rc.reshape(3, 3).T.copy(), Tc, err.value

    def assessTrackingQuality(self, estimatedPose, currFrame, return_hist=False):
        Mc = _cm4(estimatedPose)
        st = C.c_int()
        h4 = np.zeros(4, np.int32)
        o4 = np.zeros(4, np.int32)
        check(_lib.lib().revo_tracker_assess_quality(self._cam._h, _p(Mc, f32p), currFrame._h, C.byref(st), _p(h4, i32p),
                                                     _p(o4, i32p)))
        return (st.value, h4, o4) if return_hist else st.value

    def addOldPclAndPose(self, srcFrame, lvl, worldPose, timeStamp=0.0):
        """tracker.cpp:209-223; the cloud is srcFrame.return3DEdges(lvl) and stays in HBM."""
        Mc = _cm4(worldPose)
        check(_lib.lib().revo_tracker_add_old_pcl(self._cam._h, srcFrame._h, lvl, _p(Mc, f32p), float(timeStamp)))

    def addOldPcl(self, pcl, worldPose, timeStamp=0.0):
        """The reference's own signature, addOldPclAndPose(const Eigen::MatrixXf& pcl, worldPose, timeStamp)
        (tracker.cpp:209-223): pcl = N x 4 rows (X,Y,Z,1) in host memory (what return3DEdges gives)."""
        a = np.ascontiguousarray(pcl, np.float32).reshape(-1, 4)
        Mc = _cm4(worldPose)
        check(_lib.lib().revo_tracker_add_old_pcl_host(self._cam._h, _p(a, f32p), a.shape[0], _p(Mc, f32p), float(timeStamp)))

    def clearUpPastLists(self):
        check(_lib.lib().revo_tracker_clear_past(self._cam._h))

    def pastSize(self):
        return _lib.lib().revo_tracker_past_size(self._cam._h)


class BatchTracker:
    """B independent frame-pairs resident in HBM: pyramids of all 2B frames, keyframe
    promotion of the B references and TrackerNew::trackFrames of every pair, enqueued on
    one stream without host round trips.  Inputs/outputs are raw device pointers (e.g.
    torch tensors' data_ptr())."""

    def __init__(self, cameraPyr, n_pairs):
        self._cam = cameraPyr
        self.n_pairs = n_pairs
        self._h = vp()
        check(_lib.lib().revo_batch_create(cameraPyr._h, n_pairs, C.byref(self._h)))

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().revo_batch_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @staticmethod
    def _init(init_RT):
        if init_RT is None:
            return None, None
        a = np.ascontiguousarray(init_RT, np.float32)
        return a, _p(a, f32p)

    def track(self, d_bgr, d_depth, d_results, init_RT=None, stream=None):
        keep, ptr = self._init(init_RT)
        check(_lib.lib().revo_batch_track(self._h, d_bgr, d_depth, ptr, d_results, stream))

    def build(self, d_bgr, d_depth, stream=None, borrow_depth=False):
        """borrow_depth: level 0 of the depth pyramid is d_depth itself (no copy); the caller keeps the buffer alive
        and unchanged until the next build of this batch."""
        fn = _lib.lib().revo_batch_build_borrow if borrow_depth else _lib.lib().revo_batch_build
        check(fn(self._h, d_bgr, d_depth, stream))

    def build_u16(self, d_bgr, d_depth_raw, depth_scale_factor, stream=None):
        """build() for raw uint16 depth (device pointer); the metres conversion is fused into the build."""
        check(_lib.lib().revo_batch_build_u16(self._h, d_bgr, d_depth_raw, float(depth_scale_factor), stream))

    def prepare(self, stream=None):
        """Runs on `stream` the part of the last build that was left to its first consumer (the keyframes' distance
        transforms); track_only() does this itself -- call it first only to keep that work out of a timed launch."""
        check(_lib.lib().revo_batch_prepare(self._h, stream))

    def track_only(self, d_results, init_RT=None, stream=None):
        keep, ptr = self._init(init_RT)
        check(_lib.lib().revo_batch_track_only(self._h, ptr, d_results, stream))

    def sync(self, stream=None):
        """Waits for the stream and for the batch's last tracker grid (on whichever stream it ran) and checks its records'
        flags.  The d_results buffer of that launch must still be alive: it is read here (revo_hip.h, revo_batch_sync)."""
        check(_lib.lib().revo_batch_sync(self._h, stream))

    def frame(self, f, settings):
        h = vp()
        check(_lib.lib().revo_batch_frame(self._h, f, C.byref(h)))
        return ImgPyramidRGBD(settings, self._cam, _handle=h, _owned=False)

    def profile_build(self, d_bgr, d_depth, reps=3):
        """-> [(kernel name, mean us alone)]: every kernel of build(borrow_depth=True) + prepare(), HIP events between the
        launches on the batch's own stream (revo_batch_profile_build)."""
        st = StageTimes()
        check(_lib.lib().revo_batch_profile_build(self._h, d_bgr, d_depth, reps, C.byref(st)))
        return [(st.name[i].value.decode(), float(st.us[i])) for i in range(st.n)]

    def time_tracker(self, d_results, reps=5, init_RT=None, stream=None):
        keep, ptr = self._init(init_RT)
        ms = C.c_float()
        check(_lib.lib().revo_batch_time_tracker(self._h, ptr, d_results, stream, reps, C.byref(ms)))
        return ms.value


class StageTimes(C.Structure):
    """revo_stage_times"""
    _fields_ = [("n", C.c_int32), ("us", C.c_float * 24), ("name", (C.c_char * 32) * 24)]


class PipelineInfo(C.Structure):
    """revo_pipeline_info_t"""
    _fields_ = [("batches", C.c_int32), ("pairs_per_step", C.c_int32), ("tracker_streams", C.c_int32),
                ("distinct_hw_queues", C.c_int32), ("streams_replaced", C.c_int32), ("probes_run", C.c_int32),
                ("streams", C.c_void_p * 4), ("steps_submitted", C.c_uint64)]


class Pipeline:
    """revo_pipeline_*: the pipelined batch mode behind one handle -- `depth` resident batches rotate over four streams
    the library owns (build | edge lists + keyframe EDT | two tracker streams), the counterpart of the reference's IO
    thread + REVO::start loop (system.cpp:96,128-284).  submit() returns (ticket, stream): work enqueued on that stream
    before the next-but-one submit runs behind the step's tracker grid (the slot for the result collective)."""

    DEPTH_F32_BORROWED, DEPTH_F32_COPIED, DEPTH_U16 = 0, 1, 2

    def __init__(self, cameraPyr, n_pairs, depth=0, host_results=False):
        self._cam = cameraPyr
        self.n_pairs = n_pairs
        self.host_results = bool(host_results)
        self._h = vp()
        check(_lib.lib().revo_pipeline_create(cameraPyr._h, n_pairs, depth, int(bool(host_results)), C.byref(self._h)))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().revo_pipeline_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass

    def submit(self, d_bgr, d_depth, d_results=None, init_RT=None, depth_kind=0, depth_scale_factor=1.0, input_ready_event=None):
        """-> (ticket, stream handle the step's tracker grid runs on).  d_*: raw device pointers."""
        keep, ptr = BatchTracker._init(init_RT)
        ticket, stream = C.c_uint64(), vp()
        check(_lib.lib().revo_pipeline_submit(self._h, d_bgr, d_depth, depth_kind, float(depth_scale_factor), ptr, d_results,
                                              input_ready_event, C.byref(ticket), C.byref(stream)))
        return ticket.value, stream.value

    def wait(self, ticket):
        """Blocks until the step (and what the caller put behind its grid) is complete; with host_results -> its records."""
        if not self.host_results:
            check(_lib.lib().revo_pipeline_wait(self._h, ticket, None))
            return None
        out = (PairResult * self.n_pairs)()
        check(_lib.lib().revo_pipeline_wait(self._h, ticket, C.cast(out, vp)))
        return results_from_buffer(bytes(out), self.n_pairs)

    def drain(self):
        check(_lib.lib().revo_pipeline_drain(self._h))

    def info(self):
        i = PipelineInfo()
        check(_lib.lib().revo_pipeline_info(self._h, C.byref(i)))
        return dict(batches=i.batches, pairs_per_step=i.pairs_per_step, tracker_streams=i.tracker_streams,
                    distinct_hw_queues=i.distinct_hw_queues, streams_replaced=i.streams_replaced, probes_run=i.probes_run,
                    streams=[int(x or 0) for x in i.streams], steps_submitted=int(i.steps_submitted))

    def batch_frame(self, ticket, f, settings):
        """Pyramid view of frame f of the batch that holds step `ticket` (valid until its slot is submitted again)."""
        b, h = vp(), vp()
        check(_lib.lib().revo_pipeline_batch(self._h, ticket, C.byref(b)))
        check(_lib.lib().revo_batch_frame(b, f, C.byref(h)))
        return ImgPyramidRGBD(settings, self._cam, _handle=h, _owned=False)

    def time_tracker(self, every_n):
        check(_lib.lib().revo_pipeline_time_tracker(self._h, every_n))

    def set_comm(self, comm, every, d_gathered, ring):
        """revo_pipeline_set_comm: the handle enqueues the all-gather of every window of `every` steps itself, in the
        after-grid slot of the window's last step; d_gathered: raw device pointer, ring * world * every * n_pairs records."""
        check(_lib.lib().revo_pipeline_set_comm(self._h, comm._h if comm is not None else None, every, d_gathered, ring))
        self._comm = comm  # keep it alive as long as it is attached

    def flush_comm(self):
        """-> (steps of the incomplete last window that were gathered, its slot in d_gathered); (0, slot) if nothing was pending."""
        n, slot = C.c_int(), C.c_int()
        check(_lib.lib().revo_pipeline_flush_comm(self._h, C.byref(n), C.byref(slot)))
        return n.value, slot.value

    def tracker_ms(self):
        ms, n = C.c_float(), C.c_int()
        check(_lib.lib().revo_pipeline_tracker_ms(self._h, C.byref(ms), C.byref(n)))
        return ms.value, n.value


def rccl_available():
    """(path, version) of the RCCL library librevo_hip.so binds at run time, or raises RevoError."""
    buf, ver = C.create_string_buffer(256), C.c_int()
    check(_lib.lib().revo_comm_available(buf, 256, C.byref(ver)))
    return buf.value.decode(), ver.value


def comm_unique_id():
    """revo_comm_unique_id (rank 0): 128 bytes to ship to the other ranks."""
    buf = (C.c_uint8 * 128)()
    check(_lib.lib().revo_comm_unique_id(buf))
    return bytes(buf)


class Comm:
    """revo_comm_*: an RCCL communicator behind the C ABI (one process per GPU).  `uid`: the 128 bytes rank 0 got
    from comm_unique_id(), shipped to every rank by the caller.  Creation is a collective (ncclCommInitRank)."""

    def __init__(self, cameraPyr, uid, world_size, rank):
        self._cam = cameraPyr
        self._h = vp()
        buf = (C.c_uint8 * 128).from_buffer_copy(bytes(uid))
        check(_lib.lib().revo_comm_create(cameraPyr._h, buf, int(world_size), int(rank), C.byref(self._h)))
        self.world_size, self.rank = int(world_size), int(rank)

    def allgather_records(self, d_send, d_recv, n_records, stream=None):
        check(_lib.lib().revo_comm_allgather_records(self._h, d_send, d_recv, int(n_records), stream))

    def close(self):
        if getattr(self, "_h", None):
            _lib.lib().revo_comm_destroy(self._h)
            self._h = None

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class HostBatchTracker:
    """n independent frame-pairs from HOST buffers (revo_track_pairs_*): what a producer like
    IOWrapperRGBD::readNextFrame (iowrapperRGBD.cpp:301-333) holds after decoding -- BGR8 [H,W,3] and depth
    [H,W] (float32 metres, or raw uint16 + depth_scale_factor) per frame.  submit() returns when the inputs
    have been read (H2D done) while the kernels of this and earlier jobs continue; wait() returns the
    records.  Page-locked arrays (e.g. torch pin_memory) are read by DMA at PCIe speed."""

    def __init__(self, cameraPyr, depth_scale_factor=None):
        self._cam = cameraPyr
        self.depth_scale_factor = depth_scale_factor

    def _pack(self, pairs, init_RT):
        n = len(pairs)
        arr = (PairIn * n)()
        keep = []
        want = np.uint16 if self.depth_scale_factor is not None else np.float32
        # revo_track_pairs_submit takes H and W from the context and copies H rows of W pixels out of every buffer:
        # a smaller or differently shaped frame would be read past its end
        H, W = int(self._cam.settings.height), int(self._cam.settings.width)
        if init_RT is not None and len(init_RT) != n:
            raise ValueError("init_RT has %d entries for %d pairs" % (len(init_RT), n))
        for i, (ref, cur) in enumerate(pairs):
            for name, (bgr, dep) in (("ref", ref), ("cur", cur)):
                if bgr.shape != (H, W, 3) or dep.shape != (H, W):
                    raise ValueError("pair %d %s: frames must be BGR [%d,%d,3] and depth [%d,%d] (the context's size), got %s and %s"
                                     % (i, name, H, W, H, W, tuple(bgr.shape), tuple(dep.shape)))
                if bgr.dtype != np.uint8 or dep.dtype != want or bgr.strides[-1] != 1 or bgr.strides[-2] != 3 or dep.strides[-1] != dep.itemsize:
                    raise ValueError("frames must be uint8 BGR [H,W,3] and %s depth [H,W] with contiguous rows" % np.dtype(want).name)
                keep += [bgr, dep]
                setattr(arr[i], name + "_bgr", bgr.ctypes.data)
                setattr(arr[i], name + "_bgr_stride", bgr.strides[0])
                setattr(arr[i], name + "_depth", dep.ctypes.data)
                setattr(arr[i], name + "_depth_stride", dep.strides[0])
            if init_RT is not None:
                R, T = init_RT[i]
                arr[i].R_init[:] = _cm3(R).tolist()
                arr[i].T_init[:] = np.asarray(T, np.float32).tolist()
                arr[i].use_init = 1
        return arr, keep

    def submit(self, pairs, init_RT=None):
        """pairs: list of ((ref_bgr, ref_depth), (cur_bgr, cur_depth)) -> job handle."""
        arr, keep = self._pack(pairs, init_RT)
        job = vp()
        u16 = self.depth_scale_factor is not None
        check(_lib.lib().revo_track_pairs_submit(self._cam._h, len(pairs), C.cast(arr, vp), int(u16),
                                                 float(1.0 if self.depth_scale_factor is None else self.depth_scale_factor), C.byref(job)))
        return (job, len(pairs))

    def wait(self, job):
        h, n = job
        out = (PairResult * n)()
        check(_lib.lib().revo_track_pairs_wait(h, C.cast(out, vp)))
        return results_from_buffer(bytes(out), n)

    def track(self, pairs, init_RT=None):
        return self.wait(self.submit(pairs, init_RT))


def pack_init_RT(Rs, Ts):
    """list of (R 3x3 row-major numpy, T) -> n x 12 float32 (R column-major, T) for BatchTracker."""
    out = np.empty((len(Rs), 12), np.float32)
    for i, (R, T) in enumerate(zip(Rs, Ts)):
        out[i, :9] = _cm3(R)
        out[i, 9:] = np.asarray(T, np.float32)
    return out


def results_from_buffer(buf, n):
    """bytes / uint8 numpy of n revo_pair_result records -> list of dicts."""
    arr = (PairResult * n).from_buffer_copy(bytes(buf))
    out = []
    for r in arr:
        out.append(dict(R=np.array(r.R, np.float32).reshape(3, 3).T.copy(), T=np.array(r.T, np.float32),
                        err=r.err, good=r.good, bad=r.bad, status=r.status,
                        evals=np.array(r.evals, np.int32), flags=r.flags, n_pts0=r.n_pts0))
    return out
