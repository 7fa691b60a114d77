"""Randomised soak of the build path against the oracle (not collected by pytest: run it by hand on a GPU box,
`python tests/tools/soak_gpu_parity.py [n_scenes]`).  Scene families chosen to reach the rare paths of the round-2
kernels: low-contrast noise (thousands of weak runs per level: banded union-find / table overgrowth of E / flood-fill
fallback), smooth fields (long weak chains), sparse corners (fill-in -> bitmap update in k_fill), invalid depths
(validity bits), sizes whose levels are / are not multiples of 32 pixels wide (bit-tile vs byte paths)."""
import os
import sys
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import scipy.ndimage as ndi  # noqa: E402
from oracle import ro  # noqa: E402
from revo_amd import api  # noqa: E402
from revo_amd.settings import (ImgPyramidSettings, PLANE_GRAY, PLANE_DEPTH, PLANE_EDGES, PLANE_EDGES_ORIG, PLANE_HIST,  # noqa: E402
                               PLANE_EDGES3D, PLANE_DT)

SIZES = [(640, 480, 4, (20, 10, 5, 0, 0, 0)), (320, 240, 3, (10, 5, 0, 0, 0, 0)), (640, 480, 3, (20, 10, 5, 0, 0, 0)),
         (256, 192, 4, (8, 4, 0, 0, 0, 0)), (448, 336, 3, (16, 8, 4, 0, 0, 0)), (1280, 960, 5, (20, 10, 5, 0, 0, 0)),
         (1280, 1024, 3, (20, 10, 5, 0, 0, 0)), (1920, 1080, 4, (20, 10, 5, 0, 0, 0))]  # round 4: beyond one workgroup's LDS / 1024 rows


def scene(rng, w, h, kind):
    if kind == 0:    # low-contrast noise: gradients between the Canny thresholds almost everywhere
        amp = rng.uniform(25, 70)
        img = np.clip(128 + rng.normal(0, amp, (h, w)), 0, 255)
        img = ndi.gaussian_filter(img, rng.uniform(0.7, 1.3))
    elif kind == 1:  # smooth field, stretched: long curvy chains
        img = ndi.gaussian_filter(rng.uniform(0, 255, (h, w)), rng.uniform(2.0, 5.0))
        img = np.clip((img - img.mean()) * rng.uniform(6, 20) + 128, 0, 255)
    elif kind == 2:  # sparse corner
        img = np.full((h, w), 90.0)
        hh, ww = h // rng.integers(3, 8), w // rng.integers(3, 8)
        img[:hh, :ww] = rng.integers(0, 2, (hh, ww)) * 160 + 40
    else:            # blocks + noise
        img = np.kron(rng.uniform(0, 255, (h // 16 + 1, w // 16 + 1)), np.ones((16, 16)))[:h, :w] + rng.normal(0, 6, (h, w))
        img = np.clip(img, 0, 255)
    bgr = np.repeat(img.astype(np.uint8)[..., None], 3, 2)
    bgr[..., 1] = np.clip(bgr[..., 1].astype(int) + rng.integers(-3, 4), 0, 255)
    d = rng.uniform(0.05, 6.0, (h, w)).astype(np.float32)
    d[rng.uniform(0, 1, (h, w)) < rng.uniform(0, 0.4)] = 0.0
    d[rng.integers(0, h), rng.integers(0, w)] = np.nan
    d[rng.integers(0, h), rng.integers(0, w)] = np.inf
    return np.ascontiguousarray(bgr), d


def main():
    n = int(sys.argv[1]) if len(sys.argv) > 1 else 48
    rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 2026)
    bad = 0
    for i in range(n):
        w, h, lv, hist = SIZES[i % len(SIZES)] if i % 12 else SIZES[5]
        if (w, h) == (1280, 960) and i % 12:
            w, h, lv, hist = SIZES[0]
        s = ImgPyramidSettings.scaled(w, h, lv, hist_patch=hist)
        cam = api.CameraPyr(s)
        bgr, d = scene(rng, w, h, i % 4)
        gp = api.ImgPyramidRGBD(s, cam, bgr, d)
        gp.makeKeyframe()
        op = ro.Pyramid(s, bgr, d)
        op.makeKeyframe()
        names = []
        for lvl in range(s.nLevels()):
            for nm, pl in (("gray", PLANE_GRAY), ("depth", PLANE_DEPTH), ("edges", PLANE_EDGES), ("orig", PLANE_EDGES_ORIG),
                           ("pts", PLANE_EDGES3D), ("dt", PLANE_DT)) + ((("hist", PLANE_HIST),) if s.hist_patch[lvl] > 0 else ()):
                a = gp.return3DEdges(lvl) if pl == PLANE_EDGES3D else gp._read(pl, lvl)
                b = op.read(pl, lvl)
                if a.shape != b.shape or not np.array_equal(a, b, equal_nan=True):
                    names.append("%s%d" % (nm, lvl))
        ne = [int((gp._read(PLANE_EDGES, l) > 0).sum()) for l in range(s.nLevels())]
        print("scene %3d kind %d %4dx%-4d L%d edges %s %s" % (i, i % 4, w, h, lv, ne, "MISMATCH " + ",".join(names) if names else "ok"), flush=True)
        bad += bool(names)
        del gp, op, cam
    print("SOAK %s: %d of %d scenes differ" % ("FAILED" if bad else "OK", bad, n))
    return 1 if bad else 0


if __name__ == "__main__":
    sys.exit(main())
