"""NumPy restatement of the reference's heat-map post-processing.  TEST INFRASTRUCTURE ONLY.

Follows PC/src/visual.py: calculate_heatmap :143-188, find_power_center :295-322, the blends of Viewer.loop :450-452.

Pinning status: the NumPy arithmetic of calculate_heatmap (clip, log10, normalise, threshold, power, LUT index, flip) is
restated line by line and the jet table is a committed fixture (tests/golden/jet_lut.npy, generated from matplotlib as
the reference does).  The three OpenCV calls (cv2.resize INTER_LINEAR, cv2.addWeighted, cv2.GaussianBlur) cannot be
executed here -- `opencv-python` is not installed and the reference does not pin its version (PC/requirements.txt:2) --
so they are restated from OpenCV's published algorithms: PARITY UNPINNED for those three."""
import os

import numpy as np

_LUT = None


def jet_lut():
    global _LUT
    if _LUT is None:
        _LUT = np.load(os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden", "jet_lut.npy"))
    return _LUT


def small_heatmap(image, threshold=1e-7, amount=0.5, exponent=5):
    """visual.py:143-185 up to (not including) cv2.resize.  image float32 [X, Y] -> (uint8 [Y, X, 3], should_overlay)."""
    X, Y = image.shape
    small = np.zeros((Y, X, 3), dtype=np.uint8)
    colors = jet_lut()
    should = False
    safe = np.clip(image, 1e-12, None)
    if np.max(image) > threshold:
        img = np.log10(safe)
        img -= np.log10(np.min(safe))
        img /= np.max(img)
        should = True
        for x in range(X):
            for y in range(Y):
                level = img[x, y]
                if level >= amount:
                    level -= amount
                    level /= amount
                    small[Y - 1 - y, X - 1 - x] = colors[int(255 * level ** exponent)]
    return small, should


def resize_linear_u8(src, out_w, out_h):
    """cv2.resize(src, (out_w, out_h), interpolation=cv2.INTER_LINEAR) for uint8 [h, w, c] (OpenCV's fixed-point path:
    half-pixel centres, weights rounded to 11 bits each, ((b0*(r0>>4))>>16 + (b1*(r1>>4))>>16 + 2) >> 2)."""
    sh, sw = src.shape[:2]

    def axis(dsz, ssz):
        o = np.arange(dsz)
        f = ((o + 0.5) * (float(ssz) / dsz) - 0.5).astype(np.float32)
        i = np.floor(f).astype(np.int64)
        f = f - i.astype(np.float32)
        lo = i < 0
        i[lo] = 0; f[lo] = 0
        hi = i >= ssz - 1
        i[hi] = ssz - 1; f[hi] = 0
        i1 = np.where(hi, i, i + 1)
        w0 = np.rint((np.float32(1.0) - f) * np.float32(2048)).astype(np.int64)
        w1 = np.rint(f * np.float32(2048)).astype(np.int64)
        return i, i1, w0, w1

    x0, x1, wx0, wx1 = axis(out_w, sw)
    y0, y1, wy0, wy1 = axis(out_h, sh)
    s = src.astype(np.int64)
    rows = s[:, x0] * wx0[None, :, None] + s[:, x1] * wx1[None, :, None]           # [sh, out_w, c]
    r0, r1 = rows[y0] >> 4, rows[y1] >> 4
    out = (((wy0[:, None, None] * r0) >> 16) + ((wy1[:, None, None] * r1) >> 16) + 2) >> 2
    return out.astype(np.uint8)


def add_weighted_u8(a, alpha, b, beta):
    """cv2.addWeighted(a, alpha, b, beta, 0) for uint8: float32 arithmetic, round half to even, saturate."""
    v = np.float32(alpha) * a.astype(np.float32) + np.float32(beta) * b.astype(np.float32)
    return np.clip(np.rint(v), 0, 255).astype(np.uint8)


def calculate_heatmap(image, window=(1920, 1080), **kw):
    """visual.py:143-188 -> (heatmap uint8 [H, W, 3], should_overlay)."""
    small, should = small_heatmap(image, **kw)
    return resize_linear_u8(small, window[0], window[1]), should


def find_power_center(image):
    """visual.py:295-322 on clip(image, 1e-12): 5x5 Gaussian sigma 1 (BORDER_REFLECT_101), >= 95 % mask, cube weights."""
    img = np.clip(image, 1e-12, None).astype(np.float32)
    g = np.exp(-(np.arange(5) - 2.0) ** 2 / 2.0)
    g = (g / g.sum()).astype(np.float32)
    pad = np.pad(img, 2, mode="reflect")
    rows, cols = img.shape
    tmp = np.zeros((rows + 4, cols), dtype=np.float32)
    for k in range(5):
        tmp += g[k] * pad[:, k:k + cols]
    sm = np.zeros_like(img)
    for k in range(5):
        sm += g[k] * tmp[k:k + rows, :]
    mask = sm >= sm.max() * np.float32(0.95)
    yi, xi = np.indices(sm.shape)
    w = (sm ** 3) * mask
    tw = np.sum(w.astype(np.float64))
    return float(np.sum(xi * w.astype(np.float64)) / tw), float(np.sum(yi * w.astype(np.float64)) / tw)
