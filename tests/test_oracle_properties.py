"""Property tests (hypothesis) of the oracle's integer / byte primitives against independent numpy /
scipy statements, over random shapes including the tiny and odd ones the fixed-size tests skip.
CPU only.  They widen the evidence for the rows DESIGN.md lists as "parity unpinned" (cvtColor, pyrDown,
Sobel, Canny, distanceTransform): no OpenCV here, so each is checked against its published definition."""
import numpy as np
import scipy.ndimage as ndi
from hypothesis import given, settings, strategies as st

from oracle import ro
from test_oracle_primitives import _canny_bruteforce

SET = settings(max_examples=40, deadline=None)


def _img(draw, hmin, hmax, wmin, wmax, even=False, smooth=False):
    h = draw(st.integers(hmin, hmax))
    w = draw(st.integers(wmin, wmax))
    if even:
        h, w = 2 * h, 2 * w
    seed = draw(st.integers(0, 2 ** 31 - 1))
    r = np.random.default_rng(seed)
    if smooth:
        g = ndi.gaussian_filter(r.uniform(0, 255, (h, w)), draw(st.floats(0.6, 2.5)))
        g = np.clip((g - g.mean()) * draw(st.floats(2.0, 12.0)) + 128, 0, 255)
        return g.astype(np.uint8)
    return r.integers(0, 256, (h, w), dtype=np.uint8)


@SET
@given(st.data())
def test_pyrdown_property(data):
    g = _img(data.draw, 1, 20, 1, 24, even=True)
    h, w = g.shape
    k = np.array([1, 4, 6, 4, 1], np.int64)

    def refl(i, n):  # BORDER_REFLECT_101, also for n < 3 where numpy's pad would need several bounces
        i = np.abs(i)
        return np.where(i >= n, 2 * n - 2 - i, i) if n > 1 else np.zeros_like(i)

    ys = refl(np.arange(-2, h + 2), h)
    xs = refl(np.arange(-2, w + 2), w)
    if h < 3 or w < 3:  # the routine's reflect bounces once; so does this statement
        ys, xs = np.clip(ys, 0, h - 1), np.clip(xs, 0, w - 1)
    pad = g.astype(np.int64)[np.ix_(ys, xs)]
    full = sum(k[j] * k[i] * pad[j:j + h, i:i + w] for j in range(5) for i in range(5))
    ref = ((full[::2, ::2] + 128) >> 8).astype(np.uint8)
    if h >= 4 and w >= 4:
        assert np.array_equal(ro.pyrdown(g), ref)
    else:
        assert ro.pyrdown(g).shape == (h // 2, w // 2)  # degenerate sizes: shape only


@SET
@given(st.data())
def test_sobel_property(data):
    g = _img(data.draw, 1, 30, 1, 40)
    dx, dy = ro.sobel3(g)
    gi = g.astype(np.int32)
    sx = ndi.correlate(gi, np.array([[-1, 0, 1], [-2, 0, 2], [-1, 0, 1]]), mode="nearest")
    sy = ndi.correlate(gi, np.array([[-1, -2, -1], [0, 0, 0], [1, 2, 1]]), mode="nearest")
    assert np.array_equal(dx, sx) and np.array_equal(dy, sy)


@SET
@given(st.data())
def test_edt_property(data):
    h = data.draw(st.integers(1, 40))
    w = data.draw(st.integers(1, 50))
    dens = data.draw(st.sampled_from([0.0, 0.002, 0.02, 0.2, 0.9]))
    r = np.random.default_rng(data.draw(st.integers(0, 2 ** 31 - 1)))
    e = (r.uniform(0, 1, (h, w)) < dens).astype(np.uint8) * 255
    d = ro.edt(e)
    if e.any():
        assert np.array_equal(d, ndi.distance_transform_edt(e == 0).astype(np.float32))
    else:
        assert np.all(d == np.sqrt(np.float32(1e15)))


@settings(max_examples=12, deadline=None)
@given(st.data())
def test_canny_property(data):
    g = _img(data.draw, 3, 28, 3, 36, smooth=True)
    t1, t2 = data.draw(st.sampled_from([(150, 100), (100, 150), (60, 20), (255, 254), (300, 10)]))
    assert np.array_equal(ro.canny(g, t1, t2), _canny_bruteforce(g, t1, t2))


@SET
@given(st.data())
def test_histogram_and_edge_list_property(data):
    patch = data.draw(st.sampled_from([2, 5, 10]))
    th, tw = data.draw(st.integers(1, 6)), data.draw(st.integers(1, 6))
    h, w = th * patch, tw * patch
    r = np.random.default_rng(data.draw(st.integers(0, 2 ** 31 - 1)))
    e = (r.uniform(0, 1, (h, w)) < data.draw(st.sampled_from([0.0, 0.05, 0.5, 1.0]))).astype(np.uint8) * 255
    hist, frac = ro.dist_histogram(e, patch)
    ref = (e > 0).reshape(th, patch, tw, patch).sum((1, 3))
    assert np.array_equal(hist, (ref % 256).astype(np.uint8))  # u8 wrap like ++ on uchar (imgpyramidrgbd.cpp:160)
    assert abs(frac - float((hist > 0).sum()) / (th * tw)) < 1e-6
    # 3-D edge list: x outer / y inner, valid depth only, (Z*(x-cx)/fx, Z*(y-cy)/fy, Z, 1) in float32
    depth = r.uniform(-0.5, 6.0, (h, w)).astype(np.float32)
    depth[r.uniform(0, 1, (h, w)) < 0.1] = np.nan
    fx, fy, cx, cy = np.float32(50.5), np.float32(49.25), np.float32(w / 2 - 0.3), np.float32(h / 2 + 0.2)
    pts = ro.edges3d(e, depth, fx, fy, cx, cy)
    exp = []
    for x in range(w):
        for y in range(h):
            Z = depth[y, x]
            if e[y, x] > 0 and np.isfinite(Z) and np.float32(0.1) < Z < np.float32(5.2):
                exp.append([Z * (np.float32(x) - cx) / fx, Z * (np.float32(y) - cy) / fy, Z, np.float32(1)])
    assert pts.shape[0] == len(exp)
    if exp:
        assert np.array_equal(pts, np.array(exp, np.float32))


@SET
@given(st.data())
def test_depth_subsample_property(data):
    h, w = 2 * data.draw(st.integers(1, 12)), 2 * data.draw(st.integers(1, 12))
    r = np.random.default_rng(data.draw(st.integers(0, 2 ** 31 - 1)))
    d = r.uniform(-1.0, 5.0, (h, w)).astype(np.float32)
    d[r.uniform(0, 1, (h, w)) < 0.3] = 0.0
    out = ro.depth_subsample(d)
    for y in range(h // 2):
        for x in range(w // 2):
            acc, n = np.float32(0), np.float32(0)
            for v in (d[2 * y, 2 * x], d[2 * y, 2 * x + 1], d[2 * y + 1, 2 * x], d[2 * y + 1, 2 * x + 1]):
                if v > 0:
                    acc, n = np.float32(acc + v), np.float32(n + 1)
            assert out[y, x] == (np.float32(acc / n) if n > 0 else np.float32(0))
