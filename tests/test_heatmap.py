"""Heat-map post-processing (PC/src/visual.py:143-188, 295-322, 450-452).

CPU: properties of the NumPy restatement.  GPU: HIP kernels vs that restatement.  Float transcendental steps (log10, pow)
feed an integer LUT index, so colours may differ by one LUT step on a handful of pixels between libm implementations;
the upscale and the blends are integer / round-to-nearest work and must match exactly given the same small image."""
import numpy as np
import pytest

import util


def _map(name="cfg2", sig="s3"):
    g = util.golden(name)
    return np.ascontiguousarray(g["img_lerp_" + sig], dtype=np.float32)


def test_oracle_small_heatmap_properties():
    import visual_np as V
    img = _map()
    small, should = V.small_heatmap(img)
    X, Y = img.shape
    assert should and small.shape == (Y, X, 3)
    px, py = np.unravel_index(np.argmax(img), img.shape)
    assert tuple(small[Y - 1 - py, X - 1 - px]) == tuple(V.jet_lut()[255])      # the peak gets the top colour, flipped position
    assert not V.small_heatmap(img * 1e-9)[1] and not V.small_heatmap(img * 1e-9)[0].any()   # below threshold: blank, no overlay
    # levels under `amount` stay black
    lvl = np.log10(np.clip(img, 1e-12, None)); lvl -= lvl.min(); lvl /= lvl.max()
    assert not small[::-1, ::-1].transpose(1, 0, 2)[lvl < 0.5].any()


def test_oracle_resize_and_blend_properties():
    import visual_np as V
    rng = np.random.default_rng(0)
    src = rng.integers(0, 256, (13, 17, 3), dtype=np.uint8)
    assert np.array_equal(V.resize_linear_u8(src, 17, 13), src)                 # identity size
    const = np.full((5, 7, 3), 93, dtype=np.uint8)
    assert (V.resize_linear_u8(const, 64, 48) == 93).all()                        # constants survive the fixed-point weights
    up = V.resize_linear_u8(src, 170, 130)
    assert up.min() >= src.min() and up.max() <= src.max()
    a = rng.integers(0, 256, (4, 4, 3), dtype=np.uint8)
    assert np.array_equal(V.add_weighted_u8(a, 0.5, a, 0.5), a)
    assert V.add_weighted_u8(np.full((1, 1, 3), 255, np.uint8), 0.9, np.full((1, 1, 3), 255, np.uint8), 0.9).max() == 255   # saturates


@pytest.mark.gpu
@pytest.mark.parametrize("name,sig", [("cfg2", "s3"), ("cfg2", "s1"), ("shipped", "s2"), ("cfg1", "s3")])
def test_gpu_heatmap_matches_oracle(native, name, sig):
    import torch
    import visual
    import visual_np as V
    c = util.configure(name)
    img = _map(name, sig)
    want_small, want_flag = V.small_heatmap(img)
    st = visual.HeatmapStream(640, 640)
    small, flags = st.small_heatmaps(torch.from_numpy(img.reshape(1, -1)).cuda())
    got_small = small[0].cpu().numpy()
    assert bool(flags[0].item()) == want_flag
    diff = (got_small != want_small).any(axis=-1)
    assert diff.mean() <= 2e-3, diff.mean()                                       # LUT-step flips from log10f / powf rounding only
    lut = V.jet_lut().astype(int)
    for (y, x) in zip(*np.nonzero(diff)):                                         # and then by exactly one LUT step
        gi = np.flatnonzero((lut == got_small[y, x].astype(int)).all(axis=1))
        wi = np.flatnonzero((lut == want_small[y, x].astype(int)).all(axis=1))
        assert gi.size and wi.size and np.abs(gi[:, None] - wi[None, :]).min() <= 1
    # upscale + temporal blend + camera overlay: exact, given the kernel's own small image
    cam = torch.from_numpy(np.random.default_rng(2).integers(0, 256, (2, 640, 640, 3), dtype=np.uint8)).cuda()
    two = small.expand(2, -1, -1, -1).contiguous()
    out = st.overlay(two, cam).cpu().numpy()
    up = V.resize_linear_u8(got_small, 640, 640)
    prev = np.zeros_like(up)
    for f in range(2):
        res = V.add_weighted_u8(prev, 0.5, up, 0.5)
        prev = res
        assert np.array_equal(out[f], V.add_weighted_u8(cam[f].cpu().numpy(), 0.9, res, 0.9))
    assert np.array_equal(st.prev.cpu().numpy(), prev)


@pytest.mark.gpu
def test_gpu_reference_named_functions(native):
    import visual
    import visual_np as V
    util.configure("shipped")
    img = _map("shipped", "s1")
    heat, flag = visual.calculate_heatmap(img, window=(480, 270))
    want, wflag = V.calculate_heatmap(img, window=(480, 270))
    assert flag == wflag and heat.shape == (270, 480, 3)
    assert (heat != want).any(axis=-1).mean() <= 5e-3
    cx, cy = visual.find_power_center(np.clip(img, 1e-12, None))
    wx, wy = V.find_power_center(img)
    assert abs(cx - wx) < 1e-3 and abs(cy - wy) < 1e-3


@pytest.mark.gpu
@pytest.mark.parametrize("w,h,with_cam,n_frames", [(640, 640, False, 3), (322, 200, True, 3), (322, 200, False, 3), (1280, 720, True, 3), (4, 3, True, 3),
                                                   (256, 128, True, 60), (640, 640, True, 11)])
def test_gpu_overlay_both_kernels(native, w, h, with_cam, n_frames):
    """bf_heatmap_overlay_device picks the tiled kernel when the width allows it and the one-pixel kernel otherwise: both equal the
    oracle's resize + blend chain byte for byte, with and without a camera frame, over the frames of a batch with a carried `prev`
    (60 frames at 256 x 128: the tile's source pixels of all frames do not fit LDS at once -- three chunks; 11 frames: a ragged group of eight)."""
    import torch
    import visual
    import visual_np as V
    c = util.configure("cfg2")
    rng = np.random.default_rng(w * 7 + h)
    small = rng.integers(0, 256, (n_frames, c["Y"], c["X"], 3), dtype=np.uint8)
    cam = rng.integers(0, 256, (n_frames, h, w, 3), dtype=np.uint8)
    st = visual.HeatmapStream(w, h)
    st.prev.copy_(torch.from_numpy(rng.integers(0, 256, (h, w, 3), dtype=np.uint8)))
    prev = st.prev.cpu().numpy()
    out = st.overlay(torch.from_numpy(small).cuda(), torch.from_numpy(cam).cuda() if with_cam else None, 0.4, 0.7, 0.9, 0.8).cpu().numpy()
    for f in range(n_frames):
        prev = V.add_weighted_u8(prev, 0.4, V.resize_linear_u8(small[f], w, h), 0.7)
        assert np.array_equal(out[f], V.add_weighted_u8(cam[f], 0.9, prev, 0.8) if with_cam else prev)
    assert np.array_equal(st.prev.cpu().numpy(), prev)
