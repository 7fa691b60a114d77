"""Coloured keyframe cloud (imgpyramidrgbd.cpp:279-327) in the oracle, and the PLY writer
(gui/MapDrawer.h:97-170) -- CPU only."""
import numpy as np

from oracle import ro
from revo_amd import ply, synth
from revo_amd.settings import ImgPyramidSettings, PLANE_DEPTH, PLANE_EDGES3D


def _pyr():
    s = ImgPyramidSettings.scaled(160, 120, 3, hist_patch=(5, 0, 0, 0, 0, 0))
    p = synth.make_pair(3, s)
    return s, p, ro.Pyramid(s, p["ref"][0], p["ref"][1])


def test_oracle_colored_pcl_known_answers():
    s, p, o = _pyr()
    bgr = p["ref"][0]
    for lvl in range(3):
        sparse, dense = o.generateColoredPcl(lvl, False), o.generateColoredPcl(lvl, True)
        # same predicate and order as the 3-D edge list (imgpyramidrgbd.cpp:180-214 vs 300-322)
        assert np.array_equal(sparse[:, :4], o.read(PLANE_EDGES3D, lvl))
        depth = o.read(PLANE_DEPTH, lvl)
        ok = np.isfinite(depth) & (depth > s.depth_min) & (depth < s.depth_max)
        assert len(dense) == int(ok.sum())
        # colours: per-channel pyrDown of the BGR image, emitted as r,g,b in [0,1]; x-outer / y-inner
        chans = [np.ascontiguousarray(bgr[:, :, c]) for c in range(3)]
        for _ in range(lvl):
            chans = [ro.pyrdown(c) for c in chans]
        ys, xs = np.nonzero(ok.T)[1], np.nonzero(ok.T)[0]
        for k, c in ((4, 2), (5, 1), (6, 0)):
            assert np.array_equal(dense[:, k], chans[c][ys, xs].astype(np.float32) / np.float32(255.0))
        assert np.all(dense[:, 3] == 1) and np.all(dense[:, 7] == 1)
        assert np.array_equal(dense[:, 2], depth[ys, xs])


def test_model_exporter_files(tmp_path):
    s, p, o = _pyr()
    m = ply.ModelExporter()
    T1 = np.eye(4, dtype=np.float32)
    T2 = synth.se3_exp(np.array([0.1, -0.2, 0.05, 0.02, 0.01, -0.03])).astype(np.float32)
    c1, c2 = o.generateColoredPcl(0, False), o.generateColoredPcl(1, True)
    m.addPclAndKfPoseToQueue(c1, T1)
    m.addPclAndKfPoseToQueue(c2, T2)
    pcl_path, kf_path = m.saveModel(str(tmp_path))
    head = open(pcl_path).read().split("end_header\n")[0]
    assert head == ("ply\nformat ascii 1.0\nelement vertex %d\nproperty float32 x\nproperty float32 y\n"
                    "property float32 z\nproperty uchar red\nproperty uchar green\nproperty uchar blue\n"
                    % (len(c1) + len(c2)))
    v, e, ne = ply.read_ply_vertices(pcl_path)
    assert len(v) == len(c1) + len(c2) and not e and ne == 0
    allp = np.concatenate([c1, c2])
    assert np.allclose(v[:, :3], allp[:, :3], rtol=1e-5, atol=1e-6)  # %g keeps 6 significant digits
    assert np.allclose(v[:, 3:], allp[:, 4:7] * 255.0, rtol=1e-5)
    # keyframe frusta: 5 vertices and 8 edges per keyframe + 1 link, header says 9K-1 (MapDrawer.h:120)
    v, e, ne = ply.read_ply_vertices(kf_path)
    assert len(v) == 10 and ne == 17 and len(e) == 17
    assert np.allclose(v[5, :3], T2[:3, 3], rtol=1e-5) and np.all(v[:, 3:] == [0, 0, 255])
    corner = T2[:3, :3] @ np.array([0.1, 0.075, 0.06], np.float32) + T2[:3, 3]
    assert np.allclose(v[6, :3], corner, rtol=1e-5, atol=1e-6)
    assert e[0] == (0, 1, 0, 0, 255) and e[-1] == (0, 5, 0, 255, 0) and e[8] == (5, 6, 0, 0, 255)
    # world=True moves the points with the keyframe pose
    m.saveModel(str(tmp_path / "w"), world=True)
    vw, _, _ = ply.read_ply_vertices(str(tmp_path / "w" / "outputPcl.ply"))
    exp = c2[:, :3] @ T2[:3, :3].T + T2[:3, 3]
    assert np.allclose(vw[len(c1):, :3], exp, rtol=1e-4, atol=1e-5)
