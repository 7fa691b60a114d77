"""Reader for the reference's two configuration files (OpenCV FileStorage YAML, '%YAML:1.0').

  dataset file   -> ImgPyramidSettings (datastructures/camerapyr.h:40-64) + IO settings
                    (io/iowrapperRGBD.h:57-127: MainFolder, Datasets, ASSOCIATE, DEPTH_SCALE_FACTOR,
                    SKIP_FIRST_N_FRAMES, READ_N_IMAGES, useDepthTimeStamp)
  settings file  -> TrackerSettings (system/tracker.h:43-47)

Keys and defaults are the ones the reference reads with cv::read(fs[key], var, default).
"""
import re

import yaml

from .settings import ImgPyramidSettings, TrackerSettings


def _load(path):
    txt = open(path).read()
    txt = re.sub(r"^%YAML[: ]*1\.0\s*\n", "", txt)  # OpenCV's directive line is not valid YAML 1.1
    txt = txt.lstrip("-\n")
    data = yaml.safe_load(txt) or {}
    if not isinstance(data, dict):
        raise ValueError("%s: expected a key/value mapping" % path)
    return data


def load_dataset_yaml(path):
    """-> (ImgPyramidSettings, io dict)"""
    d = _load(path)
    # the pyramid's size comes from "width" / "height" (camerapyr.h:51-53; the shipped files do not set them: 640x480);
    # "Camera.width" / "Camera.height" only size the IO wrapper's images (iowrapperRGBD.h:120-121)
    width = int(d.get("width", 640))
    height = int(d.get("height", 480))
    fx = float(d.get("Camera.fx", (width + height) / 2))  # camerapyr.h:56
    s = ImgPyramidSettings(
        width=width, height=height, fx=fx, fy=float(d.get("Camera.fy", fx)),
        cx=float(d.get("Camera.cx", width / 2.0)), cy=float(d.get("Camera.cy", height / 2.0)),
        pyr_min_lvl=int(d.get("PYR_MIN_LVL", 2)), pyr_max_lvl=int(d.get("PYR_MAX_LVL", 0)),
        canny_threshold1=int(d.get("cannyThreshold1", 150)), canny_threshold2=int(d.get("cannyThreshold2", 100)),
        depth_min=float(d.get("DEPTH_MIN", 0.1)), depth_max=float(d.get("DEPTH_MAX", 5.2)),
        use_edge_hist=int(bool(d.get("USE_EDGE_HIST", 1))), n_percentage=float(d.get("nPercentage", 0.3)))
    datasets = d.get("Datasets", "")
    io = dict(main_folder=str(d.get("MainFolder", "")),
              datasets=[datasets] if isinstance(datasets, str) else list(datasets),
              associate=str(d.get("ASSOCIATE", "associate.txt")),
              depth_scale_factor=float(d.get("DEPTH_SCALE_FACTOR", 1000.0)),  # iowrapperRGBD.h:119
              img_width=int(d.get("Camera.width", 640)), img_height=int(d.get("Camera.height", 480)),
              skip_first_n_frames=int(d.get("SKIP_FIRST_N_FRAMES", 0)),
              read_n_images=int(d.get("READ_N_IMAGES", 100000)),
              use_depth_timestamp=int(bool(d.get("useDepthTimeStamp", 1))))  # iowrapperRGBD.h:126
    return s, io


def load_settings_yaml(path):
    """-> (TrackerSettings, use_edge_filter, system dict)"""
    d = _load(path)
    n_hist = int(d.get("N_FRAMES_HIST_VOTING", d.get("nFramesHistogramVoting", 3)))  # tracker.h:44,47
    ts = TrackerSettings(check_tracking_results=int(bool(d.get("CHECK_TRACKING_RESULTS", 1))),
                         check_init_values=int(bool(d.get("CHECK_INIT_VALUES", 1))),
                         n_frames_hist_voting=n_hist)
    sysd = dict(do_output_poses=int(bool(d.get("DO_OUTPUT_POSES", 1))),
                do_generate_dense_pcl=int(bool(d.get("DO_GENERATE_DENSE_PCL", 0))))
    return ts, int(bool(d.get("USE_EDGE_FILTER", 1))), sysd
