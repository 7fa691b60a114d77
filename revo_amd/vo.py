"""Sequential visual-odometry driver: REVO::start (system/system.cpp:84-305) through the C ABI
(revo_vo_* in include/revo_hip.h).

The reference runs a producer thread that builds pyramids into a queue and a consumer loop that
tracks the oldest one.  `submit` is the producer side (asynchronous on the device: the pyramid of
frame N+1 is built while frame N is tracked), `track_next` one body of the consumer loop.  `push`
is submit + track_next (no look-ahead); `run` drives both sides like the reference: a producer
(IO) thread feeding a bounded queue and the consumer loop on the calling thread.

Single stream = one GPU (frame N's initial pose and keyframe depend on frame N-1); see
api.BatchTracker for the throughput mode.
"""
import ctypes as C

import numpy as np

from . import _lib, api
from ._lib import check, f32p, u16p, u8p, vp
from .settings import TrackerSettings


class REVO:
    def __init__(self, settingsPyr, settingsTracker=None, device=0, cameraPyr=None, depth_scale_factor=None,
                 mapDrawer=None, generate_dense_pcl=False):
        self.settingsPyr = settingsPyr
        self.settingsTracker = settingsTracker or TrackerSettings()
        self.camPyr = cameraPyr or api.CameraPyr(settingsPyr, device=device)
        self.mTracker = api.TrackerNew(self.settingsTracker, settingsPyr, self.camPyr)
        self._h = vp()
        check(_lib.lib().revo_vo_create(self.camPyr._h, C.byref(self._h)))
        self.poses = []  # (timestamp, 4x4 curr->world)
        self.depth_scale_factor = depth_scale_factor  # set: depth arrives as raw uint16
        # ply.ModelExporter (MapDrawer's model half): gets one coloured cloud + pose per keyframe like
        # system.cpp:162-168,232-238; generate_dense_pcl is DO_GENERATE_DENSE_PCL
        self.mpMapDrawer = mapDrawer
        self.generate_dense_pcl = bool(generate_dense_pcl)

    def __del__(self):
        try:
            if getattr(self, "_h", None):
                _lib.lib().revo_vo_destroy(self._h)
                self._h = None
        except Exception:
            pass

    @property
    def nKeyFrames(self):
        return _lib.lib().revo_vo_num_keyframes(self._h)

    def submit(self, bgr, depth, timestamp):
        """IOWrapperRGBD::generateImgPyramidFromFiles: build the pyramid, push it to the queue."""
        bgr = np.ascontiguousarray(bgr, np.uint8)
        if self.depth_scale_factor is not None and np.asarray(depth).dtype == np.uint16:
            # iowrapperRGBD.cpp:326-327 (depth.convertTo(CV_32FC1, 1/scale)) runs inside the device build
            raw = np.ascontiguousarray(depth, np.uint16)
            h, w = raw.shape
            if bgr.shape != (h, w, 3) or (w, h) != (self.settingsPyr.width, self.settingsPyr.height):
                raise ValueError("image size does not match the settings")
            check(_lib.lib().revo_vo_submit_u16(self._h, bgr.ctypes.data_as(u8p), w * 3, raw.ctypes.data_as(u16p), w * 2,
                                                float(self.depth_scale_factor), float(timestamp)))
            return
        depth = np.ascontiguousarray(depth, np.float32)
        h, w = depth.shape
        if bgr.shape != (h, w, 3) or (w, h) != (self.settingsPyr.width, self.settingsPyr.height):
            raise ValueError("image size does not match the settings")
        check(_lib.lib().revo_vo_submit(self._h, bgr.ctypes.data_as(u8p), w * 3, depth.ctypes.data_as(f32p), w * 4,
                                        float(timestamp)))

    def track_next(self):
        """One loop body of REVO::start on the oldest queued frame -> (4x4 pose, new_keyframe)."""
        pose = np.empty(16, np.float32)
        kf, ts = C.c_int(), C.c_double()
        check(_lib.lib().revo_vo_track_next(self._h, pose.ctypes.data_as(f32p), C.byref(kf), C.byref(ts)))
        M = pose.reshape(4, 4).T.copy()
        self.poses.append((ts.value, M))
        if kf.value and self.mpMapDrawer is not None:
            kfPyr, T_w_kf = self.keyframe()
            self.mpMapDrawer.addPclAndKfPoseToQueue(kfPyr.generateColoredPcl(0, self.generate_dense_pcl), T_w_kf)
        return M, bool(kf.value)

    def keyframe(self):
        """(kfPyr, kfPyr->getTransKFtoWorld()); the pyramid is borrowed -- valid until the next track_next."""
        h = vp()
        T = np.empty(16, np.float32)
        check(_lib.lib().revo_vo_keyframe(self._h, C.byref(h), T.ctypes.data_as(f32p)))
        pyr = api.ImgPyramidRGBD(self.settingsPyr, self.camPyr, _handle=h, _owned=False)
        pyr._vo = self  # keep the driver (owner of the handle) alive
        return pyr, T.reshape(4, 4).T.copy()

    def push(self, bgr, depth, timestamp):
        self.submit(bgr, depth, timestamp)
        return self.track_next()

    def run(self, frames, io_thread=True, max_queue=4):
        """frames: iterable of (bgr, depth, timestamp).

        io_thread=True is the reference's threading (system.cpp:96, iowrapperRGBD.cpp:279-288): a
        producer thread builds pyramids into the queue (host copy into pinned staging + asynchronous
        device build) while this thread runs the consumer loop.  io_thread=False keeps one frame of
        look-ahead on a single thread.  Both give the same bits as `push` per frame."""
        if not io_thread:
            out = []
            it = iter(frames)
            try:
                f = next(it)
            except StopIteration:
                return out
            self.submit(*f[:3])
            for f in it:
                self.submit(*f[:3])
                out.append(self.track_next())
            out.append(self.track_next())
            return out
        import threading
        L = _lib.lib()
        check(L.revo_vo_set_max_queue(self._h, int(max_queue)))  # bounded queue inside the library, stream open
        state = {"err": None}

        def producer():
            try:
                for f in frames:
                    if state["err"] is not None:
                        break
                    self.submit(*f[:3])  # blocks while max_queue pyramids wait
            except BaseException as e:  # surfaced on the consumer thread
                state["err"] = e
            finally:
                L.revo_vo_close(self._h)  # end of stream: the consumer's wait returns 0 once the queue drained

        th = threading.Thread(target=producer, name="revo-io", daemon=True)
        th.start()
        out = []
        try:
            while L.revo_vo_wait_frame(self._h) == 1:
                out.append(self.track_next())
        except BaseException as e:
            state["err"] = state["err"] or e
            raise
        finally:
            L.revo_vo_set_max_queue(self._h, 0)  # never leave the producer blocked on a full queue
            th.join()
        if state["err"] is not None:
            raise state["err"]
        return out

    def tum_lines(self):
        """REVO::writePose, system.cpp:76-80: 'ts tx ty tz qx qy qz qw', std::fixed.  The stream's precision is set to 9
        AFTER the time stamp has been written and stays set (std::setprecision is sticky): the first line carries the time
        stamp with the default 6 decimals, every later line with 9 -- reproduced as the reference's file has it."""
        out = []
        for i, (ts, M) in enumerate(self.poses):
            q = _quat_xyzw(M[:3, :3])
            out.append(("%.6f" if i == 0 else "%.9f") % ts + " %.9f %.9f %.9f %.9f %.9f %.9f %.9f" % (tuple(M[:3, 3]) + tuple(q)))
        return out


def _quat_xyzw(R):
    """Eigen::Quaternionf(R) (system.cpp:78), returned as (x, y, z, w)."""
    R = np.asarray(R, np.float32)
    t = R[0, 0] + R[1, 1] + R[2, 2]
    q = np.zeros(4, np.float32)  # w x y z
    if t > 0:
        t = np.sqrt(np.float32(t + 1.0))
        q[0] = 0.5 * t
        t = np.float32(0.5) / t
        q[1], q[2], q[3] = (R[2, 1] - R[1, 2]) * t, (R[0, 2] - R[2, 0]) * t, (R[1, 0] - R[0, 1]) * t
    else:
        i = 0
        if R[1, 1] > R[0, 0]:
            i = 1
        if R[2, 2] > R[i, i]:
            i = 2
        j, k = (i + 1) % 3, (i + 2) % 3
        t = np.sqrt(np.float32(R[i, i] - R[j, j] - R[k, k] + 1.0))
        q[1 + i] = 0.5 * t
        t = np.float32(0.5) / t
        q[0] = (R[k, j] - R[j, k]) * t
        q[1 + j] = (R[j, i] + R[i, j]) * t
        q[1 + k] = (R[k, i] + R[i, k]) * t
    return q[1], q[2], q[3], q[0]
