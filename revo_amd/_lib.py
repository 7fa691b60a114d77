"""ctypes loader for revo_amd/librevo_hip.so (the C ABI of include/revo_hip.h).

There is no CPU fallback: a missing library is a hard error, and every entry
point that needs a device fails with REVO_ERR_HIP when no MI355X is visible.
"""
import ctypes as C
import os
import re

from .settings import (ImgPyramidSettings, OptimizerSettings, TrackerSettings, ResidualInfo,
                       PairResult, MAX_LEVELS)

_HERE = os.path.dirname(os.path.abspath(__file__))
# REVO_HIP_SO: an alternative build of the same library (profiling builds under profiles/); never a fallback
SO_PATH = os.environ.get("REVO_HIP_SO") or os.path.join(_HERE, "librevo_hip.so")
HEADER = os.path.join(_HERE, "..", "include", "revo_hip.h")

u8p = C.POINTER(C.c_uint8)
u16p = C.POINTER(C.c_uint16)
f32p = C.POINTER(C.c_float)
i32p = C.POINTER(C.c_int32)
vp = C.c_void_p
vpp = C.POINTER(C.c_void_p)


class RevoError(RuntimeError):
    def __init__(self, code, msg):
        super().__init__("revo_hip error %d: %s" % (code, msg))
        self.code = code


_lib = None


def declared_symbols():
    """Every function the C header declares (used by the CPU-side export test)."""
    txt = open(HEADER).read()
    txt = re.sub(r"/\*.*?\*/", "", txt, flags=re.S)
    return sorted(set(re.findall(r"\b(revo_[a-z0-9_]+)\s*\(", txt)))


def lib():
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(SO_PATH):
        raise ImportError(
            "revo_amd: %s is missing. Build it with `python -c 'import __graft_entry__ as g; g.build()'` "
            "(hipcc --offload-arch=gfx950). There is no CPU fallback." % SO_PATH)
    # torch (device memory / streams / torch.distributed plumbing) bundles its own ROCm
    # runtime; loading it FIRST makes librevo_hip.so bind to that single copy instead of
    # mixing it with /opt/rocm's (two HIP runtimes in one process corrupt the heap).
    try:
        import torch  # noqa: F401
    except ImportError:
        pass
    L = C.CDLL(SO_PATH)
    L.revo_last_error.restype = C.c_char_p
    L.revo_version.restype = C.c_char_p
    L.revo_pyr_settings_default.argtypes = [C.POINTER(ImgPyramidSettings)]
    L.revo_opt_settings_default.argtypes = [C.POINTER(OptimizerSettings)]
    L.revo_tracker_settings_default.argtypes = [C.POINTER(TrackerSettings)]
    L.revo_ctx_create.argtypes = [C.c_int, C.POINTER(ImgPyramidSettings), C.POINTER(OptimizerSettings),
                                  C.POINTER(TrackerSettings), vpp]
    L.revo_ctx_destroy.argtypes = [vp]
    L.revo_ctx_destroy.restype = None
    L.revo_ctx_set_tracker.argtypes = [vp, C.POINTER(OptimizerSettings), C.POINTER(TrackerSettings)]
    L.revo_ctx_camera.argtypes = [vp, C.c_int, f32p]
    L.revo_pyramid_create.argtypes = [vp, u8p, C.c_size_t, f32p, C.c_size_t, C.c_double, vpp]
    L.revo_pyramid_create_u16.argtypes = [vp, u8p, C.c_size_t, u16p, C.c_size_t, C.c_double, C.c_double, vpp]
    L.revo_pyramid_destroy.argtypes = [vp]
    L.revo_pyramid_destroy.restype = None
    L.revo_pyramid_make_keyframe.argtypes = [vp]
    L.revo_pyramid_is_keyframe.argtypes = [vp]
    L.revo_pyramid_timestamp.argtypes = [vp]
    L.revo_pyramid_timestamp.restype = C.c_double
    L.revo_pyramid_read.argtypes = [vp, C.c_int, C.c_int, vp, C.c_size_t, C.POINTER(C.c_size_t)]
    L.revo_pyramid_colored_pcl.argtypes = [vp, C.c_int, C.c_int, f32p, C.c_size_t, C.POINTER(C.c_size_t)]
    L.revo_optimizer_track_level.argtypes = [vp, vp, vp, f32p, f32p, C.c_int, C.POINTER(ResidualInfo), f32p]
    L.revo_optimizer_eval.argtypes = [vp, vp, vp, f32p, f32p, C.c_int, C.POINTER(ResidualInfo), f32p, f32p, f32p]
    L.revo_optimizer_solve6.argtypes = [vp, C.c_int, f32p, f32p]
    L.revo_tracker_track_frames.argtypes = [vp, vp, vp, f32p, f32p, f32p, C.POINTER(C.c_int),
                                            C.POINTER(ResidualInfo), i32p]
    L.revo_tracker_assess_quality.argtypes = [vp, f32p, vp, C.POINTER(C.c_int), i32p, i32p]
    L.revo_tracker_add_old_pcl.argtypes = [vp, vp, C.c_int, f32p, C.c_double]
    L.revo_tracker_add_old_pcl_host.argtypes = [vp, f32p, C.c_size_t, f32p, C.c_double]
    L.revo_tracker_clear_past.argtypes = [vp]
    L.revo_tracker_past_size.argtypes = [vp]
    L.revo_batch_create.argtypes = [vp, C.c_int, vpp]
    L.revo_batch_destroy.argtypes = [vp]
    L.revo_batch_destroy.restype = None
    L.revo_batch_track.argtypes = [vp, vp, vp, f32p, vp, vp]
    L.revo_batch_build.argtypes = [vp, vp, vp, vp]
    L.revo_batch_build_borrow.argtypes = [vp, vp, vp, vp]
    L.revo_batch_build_u16.argtypes = [vp, vp, vp, C.c_double, vp]
    L.revo_batch_track_only.argtypes = [vp, f32p, vp, vp]
    L.revo_batch_prepare.argtypes = [vp, vp]
    L.revo_batch_sync.argtypes = [vp, vp]
    L.revo_track_pairs_submit.argtypes = [vp, C.c_int, vp, C.c_int, C.c_double, vpp]
    L.revo_track_pairs_wait.argtypes = [vp, vp]
    L.revo_track_pairs.argtypes = [vp, C.c_int, vp, C.c_int, C.c_double, vp]
    L.revo_batch_frame.argtypes = [vp, C.c_int, vpp]
    L.revo_batch_time_tracker.argtypes = [vp, f32p, vp, vp, C.c_int, f32p]
    L.revo_batch_profile_build.argtypes = [vp, vp, vp, C.c_int, vp]
    L.revo_ctx_histogram_level.argtypes = [vp]
    L.revo_vo_create.argtypes = [vp, vpp]
    L.revo_vo_destroy.argtypes = [vp]
    L.revo_vo_destroy.restype = None
    L.revo_vo_submit.argtypes = [vp, u8p, C.c_size_t, f32p, C.c_size_t, C.c_double]
    L.revo_vo_submit_u16.argtypes = [vp, u8p, C.c_size_t, u16p, C.c_size_t, C.c_double, C.c_double]
    L.revo_vo_track_next.argtypes = [vp, f32p, C.POINTER(C.c_int), C.POINTER(C.c_double)]
    L.revo_vo_queued.argtypes = [vp]
    L.revo_vo_keyframe.argtypes = [vp, vpp, f32p]
    L.revo_vo_set_max_queue.argtypes = [vp, C.c_int]
    L.revo_vo_close.argtypes = [vp]
    L.revo_vo_wait_frame.argtypes = [vp]
    L.revo_vo_num_keyframes.argtypes = [vp]
    L.revo_pipeline_create.argtypes = [vp, C.c_int, C.c_int, C.c_int, vpp]
    L.revo_pipeline_destroy.argtypes = [vp]
    L.revo_pipeline_destroy.restype = None
    L.revo_pipeline_submit.argtypes = [vp, vp, vp, C.c_int, C.c_double, f32p, vp, vp, C.POINTER(C.c_uint64), vpp]
    L.revo_pipeline_wait.argtypes = [vp, C.c_uint64, vp]
    L.revo_pipeline_drain.argtypes = [vp]
    L.revo_pipeline_info.argtypes = [vp, vp]
    L.revo_pipeline_batch.argtypes = [vp, C.c_uint64, vpp]
    L.revo_pipeline_time_tracker.argtypes = [vp, C.c_int]
    L.revo_pipeline_tracker_ms.argtypes = [vp, f32p, C.POINTER(C.c_int)]
    L.revo_comm_available.argtypes = [C.c_char_p, C.c_size_t, C.POINTER(C.c_int)]
    L.revo_comm_unique_id.argtypes = [u8p]
    L.revo_comm_create.argtypes = [vp, u8p, C.c_int, C.c_int, vpp]
    L.revo_comm_destroy.argtypes = [vp]
    L.revo_comm_destroy.restype = None
    L.revo_comm_world.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    L.revo_comm_allgather_records.argtypes = [vp, vp, vp, C.c_int, vp]
    L.revo_pipeline_set_comm.argtypes = [vp, vp, C.c_int, vp, C.c_int]
    L.revo_pipeline_flush_comm.argtypes = [vp, C.POINTER(C.c_int), C.POINTER(C.c_int)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise RevoError(rc, (lib().revo_last_error() or b"").decode())
