"""POD mirrors of the reference's settings classes (ctypes layout of include/revo_hip.h).

Reference:
  ImgPyramidSettings  datastructures/camerapyr.h:27-89   (+ config/dataset_tum1.yaml)
  OptimizerSettings   system/optimizer.h:42-112
  TrackerSettings     system/tracker.h:31-55             (+ config/revo_settings.yaml)
  ResidualInfo        system/optimizer.h:118-140
"""
import ctypes as C

MAX_LEVELS = 6

TRACKER_STATE_OK = 0
TRACKER_STATE_LOST = 1
TRACKER_STATE_NEW_KF = 2
TRACKER_STATE_UNKNOWN = 3

PLANE_GRAY, PLANE_DEPTH, PLANE_EDGES, PLANE_EDGES_ORIG = 0, 1, 2, 3
PLANE_DT, PLANE_GRADTABLE, PLANE_EDGES3D, PLANE_HIST = 4, 5, 6, 7
PLANE_EDGES3D_TILED = 8  # the tracker's tile-ordered copy of EDGES3D (what the per-frame build writes)


class ImgPyramidSettings(C.Structure):
    """camerapyr.h:27-89; defaults = config/dataset_tum1.yaml."""
    _fields_ = [
        ("width", C.c_int32), ("height", C.c_int32),
        ("fx", C.c_float), ("fy", C.c_float), ("cx", C.c_float), ("cy", C.c_float),
        ("pyr_min_lvl", C.c_int32), ("pyr_max_lvl", C.c_int32),
        ("canny_threshold1", C.c_int32), ("canny_threshold2", C.c_int32),
        ("depth_min", C.c_float), ("depth_max", C.c_float),
        ("use_edge_hist", C.c_int32), ("n_percentage", C.c_float),
        ("hist_patch", C.c_int32 * MAX_LEVELS),
    ]

    def __init__(self, width=640, height=480, fx=517.306408, fy=516.469215,
                 cx=318.643040, cy=255.313989, pyr_min_lvl=2, pyr_max_lvl=0,
                 canny_threshold1=150, canny_threshold2=100, depth_min=0.1,
                 depth_max=5.2, use_edge_hist=1, n_percentage=0.3,
                 hist_patch=(20, 10, 5, 0, 0, 0)):
        super().__init__()
        self.width, self.height = width, height
        self.fx, self.fy, self.cx, self.cy = fx, fy, cx, cy
        self.pyr_min_lvl, self.pyr_max_lvl = pyr_min_lvl, pyr_max_lvl
        self.canny_threshold1, self.canny_threshold2 = canny_threshold1, canny_threshold2
        self.depth_min, self.depth_max = depth_min, depth_max
        self.use_edge_hist, self.n_percentage = use_edge_hist, n_percentage
        for i in range(MAX_LEVELS):
            self.hist_patch[i] = hist_patch[i] if i < len(hist_patch) else 0

    def nLevels(self):  # camerapyr.h:68-71
        return self.pyr_min_lvl - self.pyr_max_lvl + 1

    def level_size(self, lvl):  # Camera(...,scale), camerapyr.h:98-103
        s = 1.0 / (2 ** lvl)
        return int(self.width * s), int(self.height * s)

    @classmethod
    def scaled(cls, width, height, levels, **kw):
        """TUM-1 intrinsics scaled to another resolution (e.g. 1280x960, 160x120)."""
        sx = width / 640.0
        sy = height / 480.0
        return cls(width=width, height=height, fx=517.306408 * sx, fy=516.469215 * sy,
                   cx=318.643040 * sx, cy=255.313989 * sy, pyr_min_lvl=levels - 1, **kw)


class OptimizerSettings(C.Structure):
    """optimizer.h:42-112 (hot-path fields); USE_EDGE_FILTER from tracker.h:46."""
    _fields_ = [
        ("lambda_success_fac", C.c_float), ("lambda_fail_fac", C.c_float),
        ("lambda_initial", C.c_float * MAX_LEVELS),
        ("step_size_min", C.c_float * MAX_LEVELS),
        ("convergence_eps", C.c_float * MAX_LEVELS),
        ("max_its_per_lvl", C.c_int32 * MAX_LEVELS),
        ("edge_distance_lvl", C.c_float * MAX_LEVELS),
        ("huber_edge", C.c_float), ("use_edge_filter", C.c_int32),
    ]

    def __init__(self, use_edge_filter=1):
        super().__init__()
        self.lambda_success_fac = 0.5
        self.lambda_fail_fac = 2.0
        for i, ed in enumerate((30, 20, 10, 5, 5, 5)):
            self.lambda_initial[i] = 0.0
            self.step_size_min[i] = 1e-16
            self.convergence_eps[i] = 0.999
            self.max_its_per_lvl[i] = 100
            self.edge_distance_lvl[i] = ed
        self.huber_edge = 0.3
        self.use_edge_filter = use_edge_filter


class TrackerSettings(C.Structure):
    """tracker.h:31-55; histogramLevel tracker.cpp:229."""
    _fields_ = [
        ("check_tracking_results", C.c_int32), ("check_init_values", C.c_int32),
        ("n_frames_hist_voting", C.c_int32), ("histogram_level", C.c_int32),
    ]

    def __init__(self, check_tracking_results=1, check_init_values=1,
                 n_frames_hist_voting=3, histogram_level=2):
        super().__init__()
        self.check_tracking_results = check_tracking_results
        self.check_init_values = check_init_values
        self.n_frames_hist_voting = n_frames_hist_voting
        self.histogram_level = histogram_level


class ResidualInfo(C.Structure):
    """optimizer.h:118-140."""
    _fields_ = [
        ("good_pts_edges", C.c_int32), ("bad_pts_edges", C.c_int32),
        ("sum_error_unweighted", C.c_float), ("sum_error_weighted", C.c_float),
    ]


class PairResult(C.Structure):
    """revo_pair_result (include/revo_hip.h), 96 bytes."""
    _fields_ = [
        ("R", C.c_float * 9), ("T", C.c_float * 3), ("err", C.c_float),
        ("good", C.c_int32), ("bad", C.c_int32), ("status", C.c_int32),
        ("evals", C.c_int32 * MAX_LEVELS), ("flags", C.c_int32), ("n_pts0", C.c_int32),
    ]


assert C.sizeof(PairResult) == 96


class PairIn(C.Structure):
    """revo_pair_in (include/revo_hip.h): one frame-pair in host memory."""
    _fields_ = [
        ("ref_bgr", C.c_void_p), ("ref_bgr_stride", C.c_size_t),
        ("ref_depth", C.c_void_p), ("ref_depth_stride", C.c_size_t),
        ("cur_bgr", C.c_void_p), ("cur_bgr_stride", C.c_size_t),
        ("cur_depth", C.c_void_p), ("cur_depth_stride", C.c_size_t),
        ("R_init", C.c_float * 9), ("T_init", C.c_float * 3), ("use_init", C.c_int32),
    ]
