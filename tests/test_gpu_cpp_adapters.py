"""The header-only C++ adapters (revo_amd/cpp/revo_adapters.hpp) drive the same C ABI as the
Python mirror: a system.cpp-style C++ host must get bit-identical poses."""
import os
import subprocess

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_cpp_adapter_host_matches_python_host(tmp_path):
    from revo_amd import api, synth
    from revo_amd.settings import ImgPyramidSettings, TrackerSettings
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    pair = synth.make_pair(11, s)
    exe = str(tmp_path / "adapter_track")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "cpp", "adapter_track.cpp"),
                           "-pthread", "-L" + os.path.join(ROOT, "revo_amd"), "-lrevo_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "revo_amd")])
    names = []
    for tag in ("ref", "curr"):
        for k, ext in ((0, "bgr"), (1, "depth")):
            p = str(tmp_path / ("%s.%s" % (tag, ext)))
            np.ascontiguousarray(pair[tag][k]).tofile(p)
            names.append(p)
    out = subprocess.check_output([exe, "320", "240"] + names, timeout=120).decode()
    vo_lines = [ln.split()[1:] for ln in out.strip().splitlines() if ln.startswith("vo ")]
    vals = {ln.split()[0]: ln.split()[1:] for ln in out.strip().splitlines() if not ln.startswith("vo ")}
    R_cpp = np.array(vals["R"], np.float32).reshape(3, 3).T
    T_cpp = np.array(vals["T"], np.float32)

    cam = api.CameraPyr(s)
    trk = api.TrackerNew(TrackerSettings(), s, cam)
    ref = api.ImgPyramidRGBD(s, cam, *pair["ref"])
    cur = api.ImgPyramidRGBD(s, cam, *pair["curr"], timestamp=1.0 / 30)
    ref.makeKeyframe()
    trk.addOldPclAndPose(ref, trk.histogramLevel, np.eye(4), 0.0)
    st, R, T, err = trk.trackFrames(np.eye(3), np.zeros(3), ref, cur)
    M = np.eye(4, dtype=np.float32)
    M[:3, :3], M[:3, 3] = R, T
    st2 = trk.assessTrackingQuality(M, cur)
    assert np.array_equal(R_cpp, R) and np.array_equal(T_cpp, T)
    assert np.float32(vals["err"][0]) == np.float32(err)
    assert int(vals["status"][0]) == st2
    assert int(vals["n0"][0]) == cur.return3DEdges(0).shape[0]
    assert vals["notkf"] == ["error"]
    pcl = ref.generateColoredPcl(1, True)
    assert int(vals["pcl"][0]) == len(pcl) and abs(float(vals["pcl"][1]) - float(pcl.astype(np.float64).sum())) < 1e-3
    er, et = synth.pose_error(R_cpp, T_cpp, pair["T_ref_curr"])
    assert er < 3e-3 and et < 5e-3

    # revo::REVO::run (IO thread + consumer loop in C++) vs the Python driver on the same 8 frames: same bits
    from revo_amd import vo
    frames = [(pair["curr" if i % 2 else "ref"][0], pair["curr" if i % 2 else "ref"][1], i / 30.0) for i in range(8)]
    drv = vo.REVO(s)
    res = drv.run(frames)
    assert len(vo_lines) == 8 and int(vals["vokf"][0]) == drv.nKeyFrames
    for ln, (M, kf) in zip(vo_lines, res):
        assert int(ln[0]) == int(kf)
        assert np.array_equal(np.array(ln[2:], np.float32).reshape(4, 4).T, M)


def test_reference_typed_adapters_match_python_host(tmp_path):
    """The adapters' reference-shaped surface (const Eigen::MatrixXf& return3DEdges, const cv::Mat& returnEdges,
    const Eigen::Vector4f* returnOptimizationStructure, TrackerNew(settings, pyrSettings),
    addOldPclAndPose(pcl, pose, ts)) against Eigen / cv shaped stand-ins: same numbers as the Python host."""
    from revo_amd import api, synth
    from revo_amd.settings import ImgPyramidSettings, TrackerSettings
    s = ImgPyramidSettings.scaled(320, 240, 3, hist_patch=(10, 5, 0, 0, 0, 0))
    pair = synth.make_pair(11, s)
    exe = str(tmp_path / "adapter_refshape")
    subprocess.check_call(["g++", "-std=c++17", "-O2", "-o", exe, os.path.join(ROOT, "tests", "cpp", "adapter_refshape.cpp"),
                           "-pthread", "-L" + os.path.join(ROOT, "revo_amd"), "-lrevo_hip",
                           "-Wl,-rpath," + os.path.join(ROOT, "revo_amd")])
    names = []
    for tag in ("ref", "curr"):
        for k, ext in ((0, "bgr"), (1, "depth")):
            p = str(tmp_path / ("%s.%s" % (tag, ext)))
            np.ascontiguousarray(pair[tag][k]).tofile(p)
            names.append(p)
    out = subprocess.check_output([exe, "320", "240"] + names, timeout=120).decode()
    vals = {ln.split()[0]: ln.split()[1:] for ln in out.strip().splitlines()}

    cam = api.CameraPyr(s)
    trk = api.TrackerNew(TrackerSettings(), s, cam)
    ref = api.ImgPyramidRGBD(s, cam, *pair["ref"])
    cur = api.ImgPyramidRGBD(s, cam, *pair["curr"], timestamp=1.0 / 30)
    ref.makeKeyframe()
    st, R, T, err = trk.trackFrames(np.eye(3), np.zeros(3), ref, cur)
    trk.addOldPclAndPose(ref, trk.histogramLevel, np.eye(4), 0.0)        # device copy of the keyframe's cloud
    M = np.eye(4, dtype=np.float32)
    M[:3, :3], M[:3, 3] = R, T
    trk.addOldPcl(cur.return3DEdges(trk.histogramLevel), M, 1.0 / 30)    # the reference's signature: host matrix
    st2 = trk.assessTrackingQuality(M, cur)
    assert np.array_equal(np.array(vals["R"], np.float32).reshape(3, 3).T, R)
    assert np.array_equal(np.array(vals["T"], np.float32), T)
    assert np.float32(vals["err"][0]) == np.float32(err) and int(vals["status"][0]) == st2
    e3 = cur.return3DEdges(0).astype(np.float64)
    assert int(vals["n0"][0]) == e3.shape[0] and vals["n0"][2] == "1"    # the same mirror object both times
    assert abs(float(vals["sum3d"][0]) - float((e3 * [1, 2, 3, 1]).sum())) <= 1e-6 * abs(float((e3 * [1, 2, 3, 1]).sum()))
    dt, ed, tab = ref.returnDistTransform(1), ref.returnEdges(1), ref.returnOptimizationStructure(1).astype(np.float64)
    assert vals["dt"][:3] == [str(dt.shape[0]), str(dt.shape[1]), "5"]   # CV_32FC1
    assert abs(float(vals["dt"][3]) - float(dt.astype(np.float64).sum())) <= 1e-6 * float(dt.astype(np.float64).sum())
    assert int(vals["edges"][0]) == int((ed > 0).sum())
    want = float((tab * [1, 2, 3, 1]).sum())
    assert abs(float(vals["tab"][0]) - want) <= 1e-6 * abs(want)
    assert vals["notkf"] == ["error"]
