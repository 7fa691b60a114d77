"""Dataset front-end and configuration readers: TUM associate/PNG layout
and the reference's OpenCV-FileStorage YAML keys, without OpenCV."""
import numpy as np

from revo_amd import config, synth, tum
from revo_amd.settings import ImgPyramidSettings

DATASET_YAML = """%YAML:1.0
# Camera Parameters
Camera.fx: 517.306408
Camera.fy: 516.469215
Camera.cx: 318.643040
Camera.cy: 255.313989
Camera.width: 640
Camera.height: 480
cannyThreshold1: 150
cannyThreshold2: 100
MainFolder: "/data/tum/"
Datasets: "rgbd_dataset_freiburg1_xyz"
ASSOCIATE: "associate.txt"
PYR_MIN_LVL: 2
PYR_MAX_LVL: 0
DEPTH_MIN: 0.1 #in [m]
DEPTH_MAX: 5.2
USE_EDGE_HIST: 1
nPercentage: 0.3
useDepthTimeStamp: 0
SKIP_FIRST_N_FRAMES: 0
READ_N_IMAGES: 1000
DEPTH_SCALE_FACTOR: 5000.0
"""
SETTINGS_YAML = """%YAML:1.0
DO_GENERATE_DENSE_PCL: 0
CHECK_TRACKING_RESULTS: 1
CHECK_INIT_VALUES: 0
USE_EDGE_FILTER: 1
N_FRAMES_HIST_VOTING: 3
DO_OUTPUT_POSES: 1
"""


def test_yaml_readers(tmp_path):
    d, st = tmp_path / "dataset.yaml", tmp_path / "settings.yaml"
    d.write_text(DATASET_YAML)
    st.write_text(SETTINGS_YAML)
    s, io = config.load_dataset_yaml(str(d))
    ref = ImgPyramidSettings()
    assert bytes(s) == bytes(ref)  # the TUM-1 values are the library defaults
    assert io["datasets"] == ["rgbd_dataset_freiburg1_xyz"] and io["depth_scale_factor"] == 5000.0
    assert io["read_n_images"] == 1000 and io["main_folder"] == "/data/tum/"
    ts, filt, sysd = config.load_settings_yaml(str(st))
    assert ts.check_init_values == 0 and ts.check_tracking_results == 1 and ts.n_frames_hist_voting == 3
    assert filt == 1 and sysd["do_output_poses"] == 1
    # defaults for missing keys follow cv::read(..., default) (camerapyr.h:40-64)
    e = tmp_path / "empty.yaml"
    e.write_text("%YAML:1.0\nwidth: 320\nheight: 240\nCamera.width: 1280\n")
    s2, io2 = config.load_dataset_yaml(str(e))
    # camerapyr.h:51-61: the pyramid reads "width"/"height"; Camera.width only sizes the IO wrapper's images
    assert (s2.width, s2.height) == (320, 240) and io2["img_width"] == 1280
    assert s2.fx == (320 + 240) / 2 and s2.fy == s2.fx and s2.cx == 160 and s2.pyr_min_lvl == 2
    # the reference's defaults for absent keys (iowrapperRGBD.h:119,126)
    assert io2["depth_scale_factor"] == 1000.0 and io2["use_depth_timestamp"] == 1


def test_tum_layout_roundtrip(tmp_path):
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    seq = synth.make_sequence(1, s, 4)
    folder = str(tmp_path / "rgbd_dataset_synth")
    tum.write_synthetic_dataset(folder, seq)
    rows = tum.read_associate(folder + "/associate.txt")
    assert len(rows) == 4 and rows[0][1].startswith("rgb/") and rows[0][3].startswith("depth/")
    # READ_N_IMAGES = n reads n + 1 frames, like the reference's `if (nFrames > READ_N_IMAGES) break;` (iowrapperRGBD.cpp:291)
    assert len(tum.read_associate(folder + "/associate.txt", skip_first_n_frames=1, read_n_images=1)) == 2
    got = list(tum.frames(folder))
    for (bgr, raw, ts), (bgr0, depth0, ts0, T) in zip(got, seq):
        assert np.array_equal(bgr, bgr0) and abs(ts - ts0) < 1e-6 and raw.dtype == np.uint16
        assert np.abs(raw.astype(np.float32) / 5000.0 - depth0).max() <= 0.5 / 5000.0 + 1e-6
    gt = tum.read_groundtruth_positions(folder + "/groundtruth.txt")
    assert len(gt) == 4


def test_ate_and_rpe_evaluators():
    """In-repo trajectory metrics (TUM evaluate_ate / evaluate_rpe semantics)."""
    import numpy as np
    from revo_amd import synth
    rng = np.random.default_rng(0)
    gt = [np.eye(4)]
    for _ in range(30):
        gt.append(gt[-1] @ synth.se3_exp(np.concatenate([rng.normal(0, 0.01, 3), rng.normal(0, 0.01, 3)])))
    # a rigidly moved copy has zero ATE (alignment) and zero RPE (relative motion is unchanged)
    M = synth.se3_exp([0.3, -0.2, 0.5, 0.2, -0.1, 0.4])
    moved = [M @ T for T in gt]
    assert synth.ate_rmse(moved, gt) < 1e-9
    rt, rr = synth.rpe_rmse(moved, gt)
    assert rt < 1e-9 and rr < 1e-9
    # a constant 1 cm drift per frame along x of the camera: RPE = 1 cm exactly, ATE grows with the length
    drift = synth.se3_exp([0.01, 0, 0, 0, 0, 0])
    est = [gt[0]]
    for i in range(1, len(gt)):
        est.append(est[-1] @ (np.linalg.inv(gt[i - 1]) @ gt[i]) @ drift)
    rt, rr = synth.rpe_rmse(est, gt)
    assert abs(rt - 0.01) < 1e-9 and rr < 1e-9
    assert synth.ate_rmse(est, gt) > 0.02
    rt5, _ = synth.rpe_rmse(est, gt, delta=5)
    assert 0.03 < rt5 < 0.06


def test_make_pairs_renders_the_same_pairs_on_several_cores():
    """synth.make_pairs (spawned workers: GPU tests call it from a process that holds an initialised HIP runtime) returns exactly
    what make_pair returns seed by seed."""
    import numpy as np
    from revo_amd import synth
    from revo_amd.settings import ImgPyramidSettings
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    many = synth.make_pairs([11, 12, 13], s, workers=2)
    for seed, got in zip([11, 12, 13], many):
        one = synth.make_pair(seed, s)
        for k in ("ref", "curr"):
            assert np.array_equal(got[k][0], one[k][0]) and np.array_equal(got[k][1], one[k][1])
        assert np.array_equal(got["T_ref_curr"], one["T_ref_curr"])


def test_decode_pool_delivers_the_same_frames_in_order(tmp_path):
    """tum.DecodePool (several decoder processes -> a ring of frame slots in shared memory) yields exactly what the single-process
    reader yields, in order, also when the ring is shorter than the sequence and the consumer is slow or fast; a broken file
    surfaces as an error in the consumer."""
    import time
    import pytest
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    seq = synth.make_sequence(2, s, 9)
    folder = str(tmp_path / "rgbd_dataset_synth")
    tum.write_synthetic_dataset(folder, seq)
    rows = tum.read_associate(folder + "/associate.txt")
    want = list(tum.frames(folder))
    for workers, ring, nap in ((3, 4, 0.0), (2, 3, 0.01), (1, 2, 0.0)):
        with tum.DecodePool(folder, rows, 160, 120, workers=workers, ring=ring, pin=False) as pool:
            got = []
            for bgr, raw, ts in pool:
                got.append((bgr.copy(), raw.copy(), ts))  # the views are only valid until the next frame is taken
                time.sleep(nap)
        assert len(got) == len(want)
        for (b, d, t), (b0, d0, t0) in zip(got, want):
            assert np.array_equal(b, b0) and np.array_equal(d, d0) and t == t0 and d.dtype == np.uint16
    open(folder + "/" + rows[4][3], "wb").write(b"not a png")
    with tum.DecodePool(folder, rows, 160, 120, workers=2, pin=False) as pool:
        with pytest.raises(RuntimeError):
            list(pool)
