"""Keyframe model export: the non-GL half of gui/MapDrawer.h.

MapDrawer::addPclAndKfPoseToQueue (MapDrawer.h:81-95) collects one coloured cloud
(ImgPyramidRGBD::generateColoredPcl, imgpyramidrgbd.cpp:279-327) and one pose per keyframe;
MapDrawer::saveModel (MapDrawer.h:97-170) writes them as two ASCII PLY files.  The file layout
below is the reference's, including its quirks: outputPcl.ply holds the points in their own
keyframe's coordinates (the viewer applies the pose when drawing), colours are written as
float*255 in "%g" form, and the edge count in outputKf.ply's header is 9*K-1.
`world=True` is an addition (points moved into the world frame) and is off by default.
"""
import os

import numpy as np


def _g(v):
    return "%g" % float(v)  # operator<<(float), 6 significant digits


class ModelExporter:
    def __init__(self):
        self.pclKfHost = []  # N x 8 float32 per keyframe
        self.vpKfsF = []     # 4x4 float32 T_w_kf per keyframe

    @property
    def nPts(self):
        return int(sum(len(p) for p in self.pclKfHost))

    def addPclAndKfPoseToQueue(self, kfPcl, T_Wkf):
        kfPcl = np.asarray(kfPcl, np.float32)
        if kfPcl.ndim != 2 or kfPcl.shape[1] != 8:
            raise ValueError("kfPcl must be N x 8 (X,Y,Z,1,r,g,b,1)")
        self.pclKfHost.append(kfPcl)
        self.vpKfsF.append(np.asarray(T_Wkf, np.float32).reshape(4, 4))

    def pcl_lines(self, world=False):
        for pcl, T in zip(self.pclKfHost, self.vpKfsF):
            xyz = pcl[:, :3]
            if world:
                xyz = (xyz @ T[:3, :3].T + T[:3, 3]).astype(np.float32)
            rgb = pcl[:, 4:7] * np.float32(255.0)
            for p, c in zip(xyz, rgb):
                yield "%s %s %s %s %s %s" % (_g(p[0]), _g(p[1]), _g(p[2]), _g(c[0]), _g(c[1]), _g(c[2]))

    def kf_lines(self):
        w = np.float32(0.1)
        h = np.float32(w * np.float32(0.75))
        z = np.float32(w * np.float32(0.6))
        corners = np.array([[w, h, z], [w, -h, z], [-w, -h, z], [-w, h, z]], np.float32)
        for T in self.vpKfsF:
            R, t = T[:3, :3], T[:3, 3]
            for p in [t] + [R @ c + t for c in corners]:
                yield "%s %s %s 0 0 255" % (_g(p[0]), _g(p[1]), _g(p[2]))
        for k in range(len(self.vpKfsF)):
            cc, p1, p2, p3, p4 = (5 * k + i for i in range(5))
            for a, b in ((cc, p1), (cc, p2), (cc, p3), (cc, p4), (p1, p4), (p1, p2), (p2, p3), (p3, p4)):
                yield "%d %d 0 0 255" % (a, b)
            if k > 0:
                yield "%d %d 0 255 0" % ((k - 1) * 5, cc)

    def saveModel(self, directory=".", world=False):
        """Writes outputPcl.ply and outputKf.ply (MapDrawer.h:101,118) into `directory`."""
        os.makedirs(directory, exist_ok=True)
        pcl_path = os.path.join(directory, "outputPcl.ply")
        with open(pcl_path, "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex %d\n" % self.nPts)
            f.write("property float32 x\nproperty float32 y\nproperty float32 z\n"
                    "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n")
            for line in self.pcl_lines(world):
                f.write(line + "\n")
        kf_path = os.path.join(directory, "outputKf.ply")
        K = len(self.vpKfsF)
        with open(kf_path, "w") as f:
            f.write("ply\nformat ascii 1.0\nelement vertex %d\n" % (K * 5))
            f.write("property float32 x\nproperty float32 y\nproperty float32 z\n"
                    "property uchar red\nproperty uchar green\nproperty uchar blue\n")
            f.write("element edge %d\nproperty int vertex1\nproperty int vertex2\n"
                    "property uchar red\nproperty uchar green\nproperty uchar blue\nend_header\n" % (K * 9 - 1))
            for line in self.kf_lines():
                f.write(line + "\n")
        return pcl_path, kf_path


def read_ply_vertices(path):
    """Minimal reader for the files above (tests): -> (V x 6 float array, list of edge tuples)."""
    with open(path) as f:
        lines = f.read().split("\n")
    assert lines[0] == "ply" and lines[1] == "format ascii 1.0"
    nv = ne = 0
    i = 2
    while lines[i] != "end_header":
        t = lines[i].split()
        if t[:2] == ["element", "vertex"]:
            nv = int(t[2])
        if t[:2] == ["element", "edge"]:
            ne = int(t[2])
        i += 1
    body = [l for l in lines[i + 1:] if l]
    verts = np.array([[float(x) for x in l.split()] for l in body[:nv]], np.float64).reshape(nv, 6)
    edges = [tuple(int(x) for x in l.split()) for l in body[nv:]]
    return verts, edges, ne
