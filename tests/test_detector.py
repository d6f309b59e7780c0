"""Detector (BASELINE.json config 4): HIP decode + NMS against the NumPy restatement, the network against its own fp32
forward, and the reference-named caller surface.  Parity with the reference is unpinned (see oracle/detect_np.py)."""
import numpy as np
import pytest

import util


def test_oracle_nms_properties():
    import detect_np as D
    boxes = np.array([[0, 0, 10, 10], [1, 1, 11, 11], [20, 20, 30, 30], [0, 0, 10, 10.5]], dtype=np.float32)
    scores = np.array([0.9, 0.8, 0.7, 0.6], dtype=np.float32)
    assert D.nms(boxes, scores, 0.45, 300) == [0, 2]
    assert D.nms(boxes, scores, 0.96, 300) == [0, 1, 2, 3]          # IoU(box 3, box 0) = 100/105
    assert D.nms(boxes, scores, 0.45, 1) == [0]


@pytest.mark.gpu
@pytest.mark.parametrize("nhwc", [False, True])
@pytest.mark.parametrize("half", [False, True])
def test_gpu_decode_and_nms_match_oracle(native, half, nhwc):
    import torch
    import detect_np as D
    from image_detection.model import yolov5s
    from image_detection.src.yolo_smooth_tracking import Detector
    det = Detector(half=half)
    rng = np.random.default_rng(4)
    B, nc = 2, 1
    raw_np = [rng.normal(0, 1.5, (B, 3 * (5 + nc), 640 // s, 640 // s)).astype(np.float32) for s in yolov5s.STRIDES]
    for r in raw_np:                                   # a few confident, clustered objects so that NMS has work to do
        r[:, 4::6] -= 4.0
    raw_np[0][:, 4, 10:14, 20:24] = 6.0; raw_np[1][:, 10, 5:7, 5:9] = 5.0; raw_np[2][0, 16, 3, 3] = 7.0
    dt = torch.float16 if half else torch.float32
    raw = [torch.from_numpy(r).cuda().to(dt) for r in raw_np]
    if nhwc:                                           # the memory the detect convolutions write: [B][H][W][channels]
        raw = [r.contiguous(memory_format=torch.channels_last) for r in raw]
    raw_q = [r.float().cpu().numpy() for r in raw]     # what the kernel really reads
    out, n = det.postprocess(raw, conf_thres=0.1, iou_thres=0.45, max_det=300)
    out, n = out.cpu().numpy(), n.cpu().numpy()
    boxes, scores, cls = D.decode(raw_q, yolov5s.ANCHORS, yolov5s.STRIDES, nc, 0.1)
    for b in range(B):
        order = np.argsort(-scores[b], kind="stable")[:1024]
        keep = D.nms(boxes[b][order], scores[b][order], 0.45, 300)
        want = np.concatenate([boxes[b][order][keep], scores[b][order][keep][:, None]], axis=1)
        assert n[b] == len(keep) and n[b] > 3
        assert not out[b, n[b]:].any()                 # rows past the kept boxes are zeros
        got = out[b, : n[b], :5]
        gi, wi = np.lexsort((got[:, 0], -got[:, 4])), np.lexsort((want[:, 0], -want[:, 4]))
        assert np.allclose(got[gi], want[wi], rtol=2e-5, atol=2e-4)


@pytest.mark.gpu
def test_gpu_network_matches_its_fp32_forward(native):
    """fp16 channels_last inference vs the same seeded network in fp32: head logits agree to fp16 accuracy."""
    import torch
    from image_detection.model import yolov5s
    x = torch.rand((2, 3, 640, 640), device="cuda")
    ref = yolov5s.build(half=False)(x)
    got = yolov5s.build(half=True)(x.half().contiguous(memory_format=torch.channels_last))
    for a, b in zip(got, ref):
        assert a.shape == b.shape
        err = (a.float() - b).abs().max().item() / b.abs().max().item()
        assert err < 3e-2, err
    n_params = sum(p.numel() for p in yolov5s.YOLOv5s(1).parameters())
    assert 7.0e6 < n_params < 7.1e6                      # YOLOv5s with one class: 7.01 M parameters


@pytest.mark.gpu
def test_gpu_reference_named_caller(native):
    from image_detection.src import yolo_smooth_tracking as Y
    m = Y.yolo_model(None)
    frame = np.random.default_rng(2).integers(0, 256, (640, 640, 3), dtype=np.uint8)
    dets = m.get_detections(frame, conf_threshold=0.1)
    assert isinstance(dets, list) and all(len(d) == 5 and d[4] >= 0.1 for d in dets)
    # the reference's camera frame, 360 x 640 (main.pyx:632): letterboxed to 384 x 640 inside, boxes back in frame coordinates;
    # the same picture handed over already padded gives the same boxes 12 rows lower
    small = frame[:360]
    got = m.get_detections(small, conf_threshold=0.001)
    assert got and all(0 <= d[0] <= d[2] <= 640 and 0 <= d[1] <= d[3] <= 360 for d in got)
    padded = np.full((384, 640, 3), 114, dtype=np.uint8)
    padded[12:372] = small
    ref = m.get_detections(padded, conf_threshold=0.001)
    assert len(ref) == len(got)
    for a, b in zip(got, ref):
        assert a[4] == b[4] and a[0] == b[0] and a[2] == b[2]
        assert a[1] == min(max(b[1] - 12, 0.0), 360.0) and a[3] == min(max(b[3] - 12, 0.0), 360.0)
    # a 720 x 1280 frame: ultralytics scales the long side to 640 first (LetterBox, cv2.resize INTER_LINEAR) -- the same boxes as the resized picture
    # handed over directly, scaled back by 2
    import torch
    import visual_np as V
    big = np.random.default_rng(3).integers(0, 256, (720, 1280, 3), dtype=np.uint8)
    assert Y.letterbox_geometry(720, 1280) == (360, 640, 12, 0, 384, 640, 0.5) and Y.letterbox_geometry(360, 640)[:6] == (360, 640, 12, 0, 384, 640)
    assert Y.letterbox_geometry(320, 320)[:6] == (640, 640, 0, 0, 640, 640) and Y.letterbox_geometry(480, 640)[:6] == (480, 640, 0, 0, 480, 640)
    canvas = torch.empty((384, 640, 3), dtype=torch.uint8, device="cuda")
    d_big = torch.from_numpy(big).cuda()
    assert native.lib.bf_letterbox_bgr8_device(d_big.data_ptr(), 720, 1280, canvas.data_ptr(), 384, 640, 360, 640, 12, 0, 114, None) == 0, native.check()
    want = np.full((384, 640, 3), 114, dtype=np.uint8)
    want[12:372] = V.resize_linear_u8(big, 640, 360)
    assert np.array_equal(canvas.cpu().numpy(), want)
    got_big = m.get_detections(big, conf_threshold=0.001)
    ref_small = m.get_detections(want, conf_threshold=0.001)          # already 384 x 640: no letterbox inside
    assert got_big and len(got_big) == len(ref_small)
    for a, b in zip(got_big, ref_small):
        assert a[4] == b[4] and abs(a[0] - 2 * b[0]) < 1e-3 and abs(a[2] - 2 * b[2]) < 1e-3
        assert abs(a[1] - min(max(2 * (b[1] - 12), 0.0), 720.0)) < 1e-3 and abs(a[3] - min(max(2 * (b[3] - 12), 0.0), 720.0)) < 1e-3
    valid, cand = Y.split_detections([[0, 0, 1, 1, 0.7], [0, 0, 1, 1, 0.3], [0, 0, 1, 1, 0.05]])
    assert len(valid) == 1 and len(cand) == 1
    assert abs(Y.compute_iou([0, 0, 10, 10], [5, 5, 15, 15]) - 25 / 175) < 1e-12


@pytest.mark.gpu
def test_gpu_fused_pipeline(native, oracle_lib):
    """Config 4 end to end on the device: the power maps equal the standalone beamformer's, the overlaid frames equal the
    standalone post-processing, the detector consumes them."""
    import torch
    import synth, visual_np as V
    from lib import directions
    from pipeline import FusedPipeline
    c = util.configure("cfg2")
    M, N, D, B = c["M"], c["N"], c["X"] * c["Y"], 4
    pipe = FusedPipeline("lerp", 640)
    delays = directions.calculate_delays()
    pipe.load_tables(delays, np.arange(M))
    windows = np.stack([synth.s3_plane_wave(delays[30 + 10 * i, 60 - 5 * i], N, seed=i) for i in range(B)])
    cam = np.random.default_rng(8).integers(0, 256, (B, 640, 640, 3), dtype=np.uint8)
    power, frames, boxes, counts = pipe.step(torch.from_numpy(windows).cuda(), torch.from_numpy(cam).cuda())
    orc = oracle_lib.Oracle(N, c["X"], c["Y"], c["T"])
    orc.load(1, np.float32(delays))
    mics = np.arange(M, dtype=np.int32)
    p = power.cpu().numpy()
    for i in range(B):
        assert p[i].tobytes() == orc.mimo_range(1, windows[i], mics, 0, D).tobytes()
        assert np.argmax(p[i]) == (30 + 10 * i) * c["Y"] + 60 - 5 * i
    # overlay chain reproduced from the kernel's own small images
    st2 = __import__("visual").HeatmapStream(640, 640)
    small, _ = st2.small_heatmaps(power)
    prev = np.zeros((640, 640, 3), dtype=np.uint8)
    f = frames.cpu().numpy()
    for i in range(B):
        res = V.add_weighted_u8(prev, 0.5, V.resize_linear_u8(small[i].cpu().numpy(), 640, 640), 0.5)
        prev = res
        assert np.array_equal(f[i], V.add_weighted_u8(cam[i], 0.9, res, 0.9))
    assert boxes.shape == (B, 300, 6) and counts.shape == (B,)


@pytest.mark.gpu
def test_gpu_fused_step_graph_replay(native):
    """include/beamformer_hip.h says the device entry points only enqueue and are graph-capturable: capture
    FusedPipeline.step at batch 64 into a HIP graph, replay it twice, every output equals the eager step's (scripts/graph_fused.py)."""
    import importlib.util
    import os
    spec = importlib.util.spec_from_file_location("graph_fused", os.path.join(util.ROOT, "scripts", "graph_fused.py"))
    gf = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(gf)
    try:
        r = gf.graph_vs_eager(gf.build_pipeline(), 64)
        assert r == {"power": True, "frames": True, "boxes": True, "counts": True}, r
    finally:
        util.configure("cfg1")


@pytest.mark.gpu
@pytest.mark.parametrize("B,T,K", [(3, 25200, 1024), (64, 25200, 1024), (2, 700, 1024), (5, 4097, 300), (1, 1024, 1024), (3, 12000, 1000), (3, 32768, 1024), (3, 40000, 1024)])
def test_gpu_topk_candidates_match_numpy(native, B, T, K):
    """bf_topk_candidates_device (radix select + ordered tie admission + bitonic sort, one workgroup per image; keys in registers up to 32,768 boxes
    per image -- three instantiations -- and re-read from memory above that) against a
    stable NumPy argsort: descending scores, ties by lower box index, gathered boxes / classes, count of positive scores.
    Scores are drawn from a few hundred distinct values (many exact ties, also across the K-th place) plus rejected (-1) boxes."""
    import torch
    rng = np.random.default_rng(B * 100003 + T)
    scores = rng.integers(1, 400, (B, T)).astype(np.float32) / np.float32(512.0)
    scores[rng.random((B, T)) < 0.3] = -1.0
    if B >= 3:
        scores[1, :] = -1.0                       # an image without candidates
        scores[2, : T // 2] = 0.5                 # an image dominated by one value
    boxes = rng.standard_normal((B, T, 4)).astype(np.float32)
    boxes[..., 0] = np.arange(T, dtype=np.float32)[None, :]          # box 0 coordinate = its index
    cls = rng.integers(0, 80, (B, T)).astype(np.int32)
    dev = lambda a: torch.from_numpy(a).cuda()
    d_s, d_b, d_c = dev(scores), dev(boxes), dev(cls)
    top = torch.full((B, K), float("nan"), dtype=torch.float32, device="cuda")
    tb = torch.full((B, K, 4), float("nan"), dtype=torch.float32, device="cuda")
    tc = torch.full((B, K), -7, dtype=torch.int32, device="cuda")
    cnt = torch.full((B,), -7, dtype=torch.int32, device="cuda")
    assert native.lib.bf_topk_candidates_device(d_s.data_ptr(), d_b.data_ptr(), d_c.data_ptr(), B, T, K, top.data_ptr(), tb.data_ptr(), tc.data_ptr(),
                                                cnt.data_ptr(), None) == 0, native.check()
    top, tb, tc, cnt = top.cpu().numpy(), tb.cpu().numpy(), tc.cpu().numpy(), cnt.cpu().numpy()
    keff = min(K, T)
    for b in range(B):
        order = np.argsort(-scores[b], kind="stable")[:keff]
        assert np.array_equal(top[b, :keff], scores[b][order]), b
        assert np.array_equal(tb[b, :keff], boxes[b][order]) and np.array_equal(tc[b, :keff], cls[b][order])
        assert (top[b, keff:] == -1).all() and cnt[b] == int((scores[b][order] > 0).sum())


# ---- the detector's convolutions as this library's implicit-GEMM kernel (csrc/conv_kernels.hip)

import contextlib


@contextlib.contextmanager
def f32_mode(native, mode):
    """bf_conv2d_f32_mode for the duration of a block: 0 = the float32 matrix instruction (LDS-DMA and register-staged kernels then multiply and add in the same
    order: bit-identical results), 1 = the three-way bfloat16 split (the library's default; LDS-DMA kernels only, so identity across kernels is not a property)."""
    initial = native.lib.bf_conv2d_f32_mode(mode)
    try:
        yield
    finally:
        native.lib.bf_conv2d_f32_mode(initial)


CONV_CASES = [  # (B, Cin, H, W, Cout, k, stride, pad, silu)      the layer shapes of yolov5s.py plus ragged tiles
    (2, 3, 64, 96, 32, 6, 2, 2, True),        # the stem: 3 channels padded to 4, 6x6 window, stride 2 (the patch kernel)
    (1, 3, 70, 100, 32, 6, 2, 2, True),       # the stem on a picture whose 35 x 50 outputs leave ragged 8 x 16 pixel blocks
    (2, 3, 40, 48, 24, 6, 2, 2, False),       # ... with 24 output channels (weight rows past the last channel), no activation
    (2, 32, 40, 40, 64, 3, 2, 1, True),       # a strided 3x3
    (1, 64, 20, 28, 64, 3, 1, 1, True),       # a bottleneck 3x3
    (3, 128, 17, 13, 64, 1, 1, 0, True),      # 1x1; 663 pixels: a ragged last pixel tile
    (1, 1024, 10, 10, 512, 1, 1, 0, True),    # SPPF's second 1x1: the deepest K
    (2, 256, 16, 16, 18, 1, 1, 0, False),     # a detect level: 18 outputs (a ragged channel tile), bias only
    (1, 512, 7, 9, 512, 3, 2, 1, True),       # odd sizes under stride 2
    (1, 64, 12, 12, 160, 3, 1, 1, True),      # 128-channel tiles (2 x 2 wave grid), the second one a quarter full
]


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W,N,k,s,p,silu", CONV_CASES)
def test_gpu_hip_convolution_matches_torch_fp32(native, B, C, H, W, N, k, s, p, silu):
    """bf_conv2d_nhwc_f16_device (through yolov5s.HipConv) against torch's fp32 convolution of the same float16 operands.
    Tolerance: the kernel accumulates in f32 and rounds once to f16 -> 2^-10 relative to the layer's largest output, plus the
    fp16 rounding of the result itself."""
    import torch
    from image_detection.model import yolov5s
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + C + N)
    conv = torch.nn.Conv2d(C, N, k, s, p, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (C * k * k) ** 0.5)
        conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.5)
    conv = conv.cuda().half()
    x = (torch.randn((B, C, H, W), generator=g) * 1.5).cuda().half().contiguous(memory_format=torch.channels_last)
    hc = yolov5s.HipConv(conv, silu)
    got = hc(x)                                        # the default kernel: operand tiles by LDS-DMA
    assert native.lib.bf_conv2d_use_dma_kernel(0) == 1
    try:
        assert torch.equal(got, hc(x))                 # the register-staged kernel: the same products in the same order
    finally:
        native.lib.bf_conv2d_use_dma_kernel(1)
    want = torch.nn.functional.conv2d(x.float(), conv.weight.float(), conv.bias.float(), s, p)
    if silu:
        want = torch.nn.functional.silu(want)
    assert got.shape == want.shape and got.dtype == torch.float16 and got.is_contiguous(memory_format=torch.channels_last)
    err = (got.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 2e-3, err


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W,N,k,s,p,silu", CONV_CASES)
def test_gpu_hip_convolution_f32_matches_fp64(native, B, C, H, W, N, k, s, p, silu):
    """bf_conv2d_nhwc_f32_device (both float32 modes: the exact-f32 MFMA and the three-way bfloat16 split) against a float64 convolution of the same float32 operands on the CPU, and against
    torch's own float32 convolution on the GPU: 1e-5 of the layer's largest output is the bar; the kernel sits
    an order of magnitude inside it."""
    import torch
    from image_detection.model import yolov5s
    g = torch.Generator(device="cpu").manual_seed(B * 1000 + C + N)
    conv = torch.nn.Conv2d(C, N, k, s, p, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (C * k * k) ** 0.5)
        conv.bias.copy_(torch.randn(conv.bias.shape, generator=g) * 0.5)
    x = torch.randn((B, C, H, W), generator=g) * 1.5
    want64 = torch.nn.functional.conv2d(x.double(), conv.weight.double(), conv.bias.double(), s, p)
    if silu:
        want64 = torch.nn.functional.silu(want64)
    conv = conv.cuda()
    xg = x.cuda().contiguous(memory_format=torch.channels_last)
    hc = yolov5s.HipConv(conv, silu)
    want32 = torch.nn.functional.conv2d(xg, conv.weight, conv.bias, s, p)
    if silu:
        want32 = torch.nn.functional.silu(want32)
    top = want64.abs().max().item()
    initial = native.lib.bf_conv2d_f32_mode(0)
    try:
        got = hc(xg)                                       # the native float32 matrix instruction; operand tiles by LDS-DMA
        assert native.lib.bf_conv2d_use_dma_kernel(0) == 1
        try:
            assert torch.equal(got, hc(xg))                # the register-staged kernel: the same products in the same order
        finally:
            native.lib.bf_conv2d_use_dma_kernel(1)
        if C == 3:                                         # the stem's patch kernel is the default in float16 only: the same bits from it in float32
            assert native.lib.bf_conv2d_use_dma_kernel(2) == 1
            try:
                assert torch.equal(got, hc(xg))
            finally:
                native.lib.bf_conv2d_use_dma_kernel(1)
        native.lib.bf_conv2d_f32_mode(1)
        split = hc(xg)                                     # three-way bfloat16 split of both operands, six products: float32 accuracy on the bfloat16 pipes
    finally:
        native.lib.bf_conv2d_f32_mode(initial)
    for name, out in (("native", got), ("split", split)):
        assert out.shape == want64.shape and out.dtype == torch.float32 and out.is_contiguous(memory_format=torch.channels_last)
        e64 = (out.double().cpu() - want64).abs().max().item() / top
        e32 = (out - want32).abs().max().item() / top
        assert e64 < 5e-6, (name, e64)
        assert e32 < 1e-5, (name, e32)
    e_native = (got.double().cpu() - want64).abs().max().item() / top
    e_split = (split.double().cpu() - want64).abs().max().item() / top
    assert e_split <= 2.0 * e_native + 2e-7, (e_split, e_native)      # the split is as accurate as the float32 instruction (usually more: exact products)


@pytest.mark.gpu
def test_gpu_convolution_past_the_dma_kernels_2_gib_range(native):
    """The LDS-DMA kernel addresses its operands through 32-bit buffer offsets: an input tensor of 2 GiB or more (here float32 [136, 32, 352, 352] =
    2.16 GB) must take the register-staged kernel by itself, and a tensor just under the limit the DMA kernel -- both against torch on the first and the
    last image."""
    import torch
    from image_detection.model import yolov5s
    g = torch.Generator(device="cpu").manual_seed(23)
    conv = torch.nn.Conv2d(32, 64, 3, 2, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / 17.0)
    conv = conv.cuda()
    hc = yolov5s.HipConv(conv, True)
    for B in (136, 132):                                # 2.16 GB and 2.09 GB: either side of 2^31 - 16 bytes
        x = torch.empty((B, 32, 352, 352), dtype=torch.float32, device="cuda").contiguous(memory_format=torch.channels_last)
        assert (x.numel() * 4 >= 2 ** 31) == (B == 136)
        x.normal_(generator=None)
        y = hc(x)
        for b in (0, B - 1):
            want = torch.nn.functional.silu(torch.nn.functional.conv2d(x[b:b + 1], conv.weight, conv.bias, 2, 1))
            assert (y[b:b + 1] - want).abs().max().item() / want.abs().max().item() < 1e-5, (B, b)
        del x, y
        torch.cuda.empty_cache()


@pytest.mark.gpu
@pytest.mark.parametrize("half", [False, True])
def test_gpu_hip_convolution_without_bias_and_with_a_ragged_channel_count(native, half):
    """A convolution without a bias term (NULL pointer through the C-ABI) and 40 output channels (the 64-channel tile two thirds full, its bias piece
    partly out of range): both kernels against torch."""
    import torch
    from image_detection.model import yolov5s
    dt = torch.float16 if half else torch.float32
    g = torch.Generator(device="cpu").manual_seed(17)
    for bias in (False, True):
        conv = torch.nn.Conv2d(64, 40, 3, 1, 1, bias=bias)
        with torch.no_grad():
            conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / 24.0)
        conv = conv.cuda().to(dt)
        x = torch.randn((2, 64, 18, 22), generator=g).cuda().to(dt).contiguous(memory_format=torch.channels_last)
        hc = yolov5s.HipConv(conv, False)
        want = torch.nn.functional.conv2d(x.float(), conv.weight.float(), None if conv.bias is None else conv.bias.float(), 1, 1)
        with f32_mode(native, 0):
            got = hc(x)
            native.lib.bf_conv2d_use_dma_kernel(0)
            try:
                assert torch.equal(got, hc(x))
            finally:
                native.lib.bf_conv2d_use_dma_kernel(1)
        assert (got.float() - want).abs().max().item() / want.abs().max().item() < (2e-3 if half else 1e-5)
        with f32_mode(native, 1):
            split = hc(x)
        assert (split.float() - want).abs().max().item() / want.abs().max().item() < (2e-3 if half else 1e-5)


@pytest.mark.gpu
@pytest.mark.parametrize("half", [False, True])
@pytest.mark.parametrize("up", [False, True])
def test_gpu_hip_1x1_over_a_virtual_concatenation(native, half, up):
    """bf_conv1x1_cat_nhwc_*: a 1x1 layer reading torch.cat((upsample(a) if up else a, b), 1) from its two sources -- channel slices of
    wider buffers -- equals the same kernel on the materialised concatenation bit for bit (same K order), into a slice, with a residual."""
    import torch
    from image_detection.model import yolov5s
    dt = torch.float16 if half else torch.float32
    cl = torch.channels_last
    g = torch.Generator(device="cpu").manual_seed(11 + up)
    B, c1, c2, H, W, N = 2, 48, 16, 12, 20, 72
    conv = torch.nn.Conv2d(c1 + c2, N, 1, bias=True)
    with torch.no_grad():
        conv.weight.copy_(torch.randn(conv.weight.shape, generator=g) / (c1 + c2) ** 0.5)
    hc = yolov5s.HipConv(conv.cuda().to(dt), True)
    ha, wa = (H // 2, W // 2) if up else (H, W)
    big_a = torch.randn((B, c1 + 8, ha, wa), generator=g).cuda().to(dt).contiguous(memory_format=cl)
    big_b = torch.randn((B, 8 + c2, H, W), generator=g).cuda().to(dt).contiguous(memory_format=cl)
    a, b = big_a[:, 8:], big_b[:, :c2]                 # channel slices: pixel pitch wider than the channel count, offset bases
    full = torch.cat((torch.nn.functional.interpolate(a, scale_factor=2, mode="nearest") if up else a, b), 1).contiguous(memory_format=cl)
    res = torch.randn((B, N, H, W), generator=g).cuda().to(dt).contiguous(memory_format=cl)
    ref = torch.nn.functional.silu(torch.nn.functional.conv2d(full.float(), conv.weight.cuda().to(dt).float(), conv.bias.cuda().float())) + res.float()
    a2, b2 = big_a[:, 8:40], res[:, :32]
    full2 = torch.cat((torch.nn.functional.interpolate(a2, scale_factor=2, mode="nearest") if up else a2, b2), 1).contiguous(memory_format=cl)
    with f32_mode(native, 0):                           # (float16 is not touched by the mode; float32 identities hold on the float32 instruction)
        want = hc(full, residual=res)
        buf = torch.full((B, N + 8, H, W), 3.0, dtype=dt, device="cuda").contiguous(memory_format=cl)
        got = hc(a, x2=b, up=up, out=buf[:, 8:], residual=res)
        assert got.data_ptr() == buf[:, 8:].data_ptr()
        assert torch.equal(buf[:, 8:], want) and bool((buf[:, :8] == 3.0).all())
        assert (want.float() - ref).abs().max().item() / ref.abs().max().item() < (2e-3 if half else 1e-5)
        one = hc(full[:, : c1 + c2])                       # a single dense source through the plain entry
        assert torch.equal(one, hc(full[:, :c1], x2=full[:, c1:]))
        # c1 = 48 channels is not a whole 64-byte stage for float16: that call ran on the register-staged kernel.  Sources that meet on a stage
        # edge (32 + 32 channels) take the LDS-DMA kernel in both precisions: the same answer from both kernels and from the materialised tensor.
        d1 = hc(a2, x2=b2, up=up)
        assert native.lib.bf_conv2d_use_dma_kernel(0) == 1
        try:
            d0 = hc(a2, x2=b2, up=up)
        finally:
            native.lib.bf_conv2d_use_dma_kernel(1)
        assert torch.equal(d1, d0) and torch.equal(d1, hc(full2))
    if not half:
        with f32_mode(native, 1):                       # the split mode: the same calls, the same answers to float32 accuracy, and identical between the two addressings
            buf2 = torch.full((B, N + 8, H, W), 3.0, dtype=dt, device="cuda").contiguous(memory_format=cl)
            got2 = hc(a, x2=b, up=up, out=buf2[:, 8:], residual=res)
            assert (got2.float() - ref).abs().max().item() / ref.abs().max().item() < 1e-5 and bool((buf2[:, :8] == 3.0).all())
            d2 = hc(a2, x2=b2, up=up)
            assert torch.equal(d2, hc(full2))
            assert (d2 - d1).abs().max().item() / d1.abs().max().item() < 1e-5
    with pytest.raises(Exception):
        yolov5s.HipConv(torch.nn.Conv2d(64, 8, 3, padding=1).cuda().to(dt), True)(torch.zeros((1, 32, 4, 4), dtype=dt, device="cuda"), x2=torch.zeros((1, 32, 4, 4), dtype=dt, device="cuda"))


@pytest.mark.gpu
def test_gpu_network_on_hip_convolutions_f32(native):
    """The reference's precision: the whole YOLOv5s-shaped network in float32 with every convolution on the library's exact-f32 MFMA
    kernel agrees with torch's own float32 forward to 1e-4 of each head map's largest logit, and no concatenation,
    upsampling or relayout kernel of torch's is left in the forward."""
    import torch
    from image_detection.model import yolov5s
    x = torch.rand((2, 3, 640, 640), device="cuda")
    ref = yolov5s.build(half=False)(x)
    net = yolov5s.build(half=False, conv_backend="hip")
    assert sum(isinstance(m, yolov5s.HipConv) for m in net.modules()) == 60 and not any(isinstance(m, torch.nn.Conv2d) for m in net.modules())
    xp = torch.zeros((2, 4, 640, 640), device="cuda").contiguous(memory_format=torch.channels_last)
    xp[:, :3] = x
    with torch.profiler.profile(activities=[torch.profiler.ProfilerActivity.CUDA]) as prof:
        got = net(xp)
        torch.cuda.synchronize()
    for a, b in zip(got, ref):
        assert a.shape == b.shape and a.dtype == torch.float32
        err = (a - b).abs().max().item() / b.abs().max().item()
        assert err < 1e-4, err
    names = {e.key for e in prof.key_averages() if getattr(e, "device_time_total", 0) > 0 or getattr(e, "cuda_time_total", 0) > 0}
    foreign = sorted(n for n in names if "bf::" not in n and "Memcpy" not in n and "Memset" not in n)
    assert not foreign, foreign


@pytest.mark.gpu
def test_gpu_network_on_hip_convolutions(native):
    """Fast mode: the network in float16 with every convolution on the library's kernel: head logits agree with the fp32 forward
    to fp16 accuracy (the same bound the MIOpen path is held to), and with the MIOpen fp16 forward relative to the fp16 maps' own size."""
    import torch
    from image_detection.model import yolov5s
    x = torch.rand((2, 3, 640, 640), device="cuda")
    xh = x.half().contiguous(memory_format=torch.channels_last)
    ref = yolov5s.build(half=False)(x)
    mi = yolov5s.build(half=True)(xh)
    net = yolov5s.build(half=True, conv_backend="hip")
    assert sum(isinstance(m, yolov5s.HipConv) for m in net.modules()) == 60 and not any(isinstance(m, torch.nn.Conv2d) for m in net.modules())
    got = net(xh)
    for a, b, c in zip(got, ref, mi):
        assert a.shape == b.shape
        assert (a.float() - b).abs().max().item() / b.abs().max().item() < 3e-2
        assert (a.float() - c.float()).abs().max().item() / c.float().abs().max().item() < 3e-2


def _best_partner(box, others):
    """(index, IoU) of the box of `others` [n, 6] that overlaps `box` most."""
    if len(others) == 0:
        return -1, 0.0
    iw = np.minimum(box[2], others[:, 2]) - np.maximum(box[0], others[:, 0])
    ih = np.minimum(box[3], others[:, 3]) - np.maximum(box[1], others[:, 1])
    inter = np.maximum(iw, 0.0) * np.maximum(ih, 0.0)
    union = (box[2] - box[0]) * (box[3] - box[1]) + (others[:, 2] - others[:, 0]) * (others[:, 3] - others[:, 1]) - inter
    v = np.where(union > 0, inter / np.maximum(union, 1e-30), 0.0)
    j = int(np.argmax(v))
    return j, float(v[j])


def _calibrate_head(net, maps, obj_mean=-6.0, obj_std=2.5, box_std=0.5, cls_bias=8.0):
    """Rescale the detect rows so that on the frames that produced `maps` every anchor's objectness logit is ~N(obj_mean, obj_std) over
    the cells and the box logits ~N(0, box_std), class probability ~1: confidences spread over (0, 1) and varied boxes, as a trained
    head gives, from a random-init trunk (the reference's weights are not in its repository)."""
    import torch
    with torch.no_grad():
        for d, m in zip(net.detect, maps):
            no = m.shape[1] // 3
            v = m.view(m.shape[0], 3, no, m.shape[2], m.shape[3])
            for a in range(3):
                for r in range(5):
                    mu, sd = v[:, a, r].mean().item(), v[:, a, r].std().item()
                    row, b0 = a * no + r, d.bias[a * no + r].item()
                    k = (obj_std if r == 4 else box_std) / sd
                    d.weight[row].mul_(k)
                    d.bias[row] = (obj_mean if r == 4 else 0.0) - k * (mu - b0)
                d.bias[a * no + 5:(a + 1) * no] = cls_bias
    return net


@pytest.mark.gpu
def test_gpu_detector_boxes_agree_across_precisions_and_backends(native):
    """Box-level agreement on the surface yolo_smooth_tracking.py:13-23 hands its caller: seeded frames through (a) the float32 HIP
    detector = the reference's precision, (b) the same network on torch's float32 convolutions, (c) the float16 HIP fast mode.
    The trunk is initialised so that the image reaches the head (yolov5s.build(init_gain=1.4)) and the head is calibrated to fire on
    ~2 % of its boxes with confidences across (0.25, 1): 110-180 boxes per 320x320 frame after NMS.
      (a) vs (b): equal counts, every box has a partner with IoU >= 0.995 and |dconf| <= 1e-4, same class.
      (c) vs (a): every box with conf >= 0.27 of either has a partner in the other (taken at conf >= 0.20) with IoU >= 0.95 and
                  |dconf| <= 2e-2, up to 2 % of the boxes compared excepted: a greedy-NMS decision right at the 0.45 overlap threshold can
                  fall the other way in float16, which keeps one box more and may suppress a neighbour of it (seen: 0 / 0 / 4 / 0 boxes
                  of ~250 compared per frame); the counts at conf >= 0.25 differ by no more than those plus the boxes within 0.02 of 0.25."""
    import torch
    import image_detection.model.yolov5s as Y
    from image_detection.src.yolo_smooth_tracking import Detector
    g = torch.Generator(device="cpu").manual_seed(21)
    frames = torch.randint(0, 256, (4, 320, 320, 3), dtype=torch.uint8, generator=g).cuda()
    x = (frames.flip(-1).permute(0, 3, 1, 2).float() / 255).contiguous(memory_format=torch.channels_last)
    with torch.no_grad():
        maps = Y.build(half=False, init_gain=1.4)(x)
    dets = {}
    for name, half, backend in (("f32_hip", False, "hip"), ("f32_torch", False, "miopen"), ("f16_hip", True, "hip")):
        net = _calibrate_head(Y.build(half=False, init_gain=1.4), maps)
        if half:
            net = net.half()
        det = Detector(half=half, conv_backend="miopen")
        det.net, det.conv_backend = (Y.use_hip_convs(net) if backend == "hip" else net), backend
        out, n = det.detect(frames, conf_thres=0.20)
        out, n = out.cpu().numpy().copy(), n.cpu().numpy().copy()
        dets[name] = [out[b, : n[b]] for b in range(frames.shape[0])]
    at = lambda boxes, c: boxes[boxes[:, 4] >= c]
    for b in range(frames.shape[0]):
        ref, tor, f16 = dets["f32_hip"][b], dets["f32_torch"][b], dets["f16_hip"][b]
        assert 40 < len(at(ref, 0.25)) < 300, len(at(ref, 0.25))            # the calibration did what it is for (and max_det does not clip)
        assert len(at(ref, 0.25)) == len(at(tor, 0.25))
        for box in at(ref, 0.25):
            j, v = _best_partner(box, tor)
            assert v >= 0.995 and abs(box[4] - tor[j, 4]) <= 1e-4 and box[5] == tor[j, 5], (b, box, v)
        misses = compared = 0
        for mine, other in ((ref, f16), (f16, ref)):
            for box in at(mine, 0.27):
                j, v = _best_partner(box, other)
                compared += 1
                if not (v >= 0.95 and abs(box[4] - other[j, 4]) <= 2e-2 and box[5] == other[j, 5]):
                    misses += 1
        border = int(((ref[:, 4] > 0.23) & (ref[:, 4] < 0.27)).sum())
        assert misses <= 0.02 * compared, (b, misses, compared)
        assert abs(len(at(ref, 0.25)) - len(at(f16, 0.25))) <= misses + border, (b, len(at(ref, 0.25)), len(at(f16, 0.25)), border)


@pytest.mark.gpu
def test_gpu_postprocess_kernels_stay_inside_their_buffers(native):
    """Guard bands: decode -> candidate selection -> NMS at batch 64 on the full 25,200-box head, every buffer the
    three kernels are handed sitting between two canary blocks inside one allocation; all canaries are intact afterwards and the
    results equal a run on plain buffers."""
    import ctypes as C
    import torch
    from image_detection.model import yolov5s
    B, nc, K, max_det = 64, 1, 1024, 300
    g = torch.Generator(device="cpu").manual_seed(5)
    raw = [(torch.randn((B, 3 * (5 + nc), 640 // s, 640 // s), generator=g) * 1.5).cuda() for s in yolov5s.STRIDES]
    T = 3 * sum(r.shape[2] * r.shape[3] for r in raw)
    GUARD = 1024                                       # elements of 4 bytes on either side
    CANARY = 0x5A5AA5A5

    def guarded(n_elems, dtype):
        t = torch.full((n_elems + 2 * GUARD,), CANARY, dtype=torch.int32, device="cuda")
        return t, t[GUARD:GUARD + n_elems].view(dtype)

    spec = dict(boxes=(B * T * 4, torch.float32), scores=(B * T, torch.float32), cls=(B * T, torch.int32), top=(B * K, torch.float32),
                cb=(B * K * 4, torch.float32), cc=(B * K, torch.int32), counts=(B, torch.int32), mask=(B * K * (K // 64) * 2, torch.int32),
                out=(B * max_det * 6, torch.float32), n_out=(B, torch.int32))
    anchors = np.ascontiguousarray(np.asarray(yolov5s.ANCHORS, dtype=np.float32).reshape(3, 3, 2))
    hs = (C.c_int * 3)(*[int(r.shape[2]) for r in raw]); ws = (C.c_int * 3)(*[int(r.shape[3]) for r in raw]); st = (C.c_int * 3)(*yolov5s.STRIDES)
    ptrs = (C.c_void_p * 3)(*[r.data_ptr() for r in raw])
    results = []
    for use_guards in (True, False):
        bufs = {k: guarded(n, dt) if use_guards else (None, torch.zeros(n, dtype=torch.int32, device="cuda").view(dt)) for k, (n, dt) in spec.items()}
        v = {k: b[1] for k, b in bufs.items()}
        lib = native.lib
        assert lib.bf_yolo_decode_device(ptrs, hs, ws, st, native.fptr(anchors), B, nc, 0, 0.1, v["boxes"].data_ptr(), v["scores"].data_ptr(), v["cls"].data_ptr(), None) == 0, native.check()
        assert lib.bf_topk_candidates_device(v["scores"].data_ptr(), v["boxes"].data_ptr(), v["cls"].data_ptr(), B, T, K, v["top"].data_ptr(), v["cb"].data_ptr(),
                                             v["cc"].data_ptr(), v["counts"].data_ptr(), None) == 0, native.check()
        assert lib.bf_nms_device(v["cb"].data_ptr(), v["top"].data_ptr(), v["cc"].data_ptr(), v["counts"].data_ptr(), B, K, 0.45, max_det, v["mask"].data_ptr(),
                                 v["out"].data_ptr(), v["n_out"].data_ptr(), None) == 0, native.check()
        torch.cuda.synchronize()
        if use_guards:
            for k, (whole, _) in bufs.items():
                assert bool((whole[:GUARD] == CANARY).all()) and bool((whole[-GUARD:] == CANARY).all()), k
        results.append((v["out"].clone(), v["n_out"].clone()))
    assert torch.equal(results[0][0], results[1][0]) and torch.equal(results[0][1], results[1][1]) and int(results[0][1].min()) > 0


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
def test_gpu_hip_convolution_into_a_slice_with_residual(native, half):
    """The C3 block's use: one convolution writes the upper channel half of the concatenation buffer, another the lower half with the
    bottleneck's residual added -- equal to torch.cat((x + silu(conv_a(x)), silu(conv_b(x))), 1) computed from the same kernel's
    plain outputs (the add as torch adds two float16 tensors), bit for bit; untouched channels stay untouched."""
    import torch
    from image_detection.model import yolov5s
    g = torch.Generator(device="cpu").manual_seed(7)
    dt = torch.float16 if half else torch.float32
    B, C, H, W = 2, 64, 24, 20
    convs = []
    for k in (3, 1):
        c = torch.nn.Conv2d(C, C, k, 1, k // 2, bias=True)
        with torch.no_grad():
            c.weight.copy_(torch.randn(c.weight.shape, generator=g) / (C * k * k) ** 0.5)
            c.bias.copy_(torch.randn(c.bias.shape, generator=g) * 0.5)
        convs.append(yolov5s.HipConv(c.cuda().to(dt), True))
    x = (torch.randn((B, C, H, W), generator=g) * 1.5).cuda().to(dt).contiguous(memory_format=torch.channels_last)
    plain_a, plain_b = convs[0](x), convs[1](x)
    buf = torch.full((B, 2 * C + 8, H, W), 7.0, dtype=dt, device="cuda").contiguous(memory_format=torch.channels_last)
    ra = convs[0](x, out=buf[:, :C], residual=x)
    rb = convs[1](x, out=buf[:, C:2 * C])
    assert ra.data_ptr() == buf.data_ptr() and rb.data_ptr() == buf[:, C:].data_ptr()
    assert torch.equal(buf[:, :C], x + plain_a) and torch.equal(buf[:, C:2 * C], plain_b) and bool((buf[:, 2 * C:] == 7.0).all())
    with pytest.raises(Exception):
        convs[0](x, out=buf[:, :C].permute(0, 1, 3, 2))


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
def test_gpu_preprocess_kernel_matches_torch(native, half):
    """bf_preprocess_bgr8(_f32)_device against frames.flip(-1).half() / 255 (.float() / 255): RGB order, the same values, a zero fourth channel."""
    import torch
    from image_detection.src.yolo_smooth_tracking import Detector
    frames = torch.randint(0, 256, (3, 64, 96, 3), dtype=torch.uint8, device="cuda")
    x = Detector(half=half, conv_backend="hip").preprocess(frames)
    want = frames.flip(-1).permute(0, 3, 1, 2)
    want = (want.half() if half else want.float()) / 255
    assert tuple(x.shape) == (3, 4, 64, 96) and x.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(x[:, :3], want) and bool((x[:, 3] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
@pytest.mark.parametrize("B,C,H,W", [(2, 256, 20, 20), (1, 64, 12, 20), (3, 8, 5, 3), (1, 64, 40, 40), (2, 24, 9, 7)])
def test_gpu_sppf_pool_kernel_matches_torch(native, B, C, H, W, half):
    """bf_sppf_pool(_f32)_device against nn.MaxPool2d(5, 1, 2) applied once, twice, three times: exact; the first quarter untouched.
    (The shapes walk the kernel's chunks-per-workgroup choice: 4, 2 and 1 by channel count, 1 by the LDS bound at 40 x 40.)"""
    import torch
    from lib import _native as nat
    g = torch.Generator(device="cpu").manual_seed(C + H)
    dt = torch.float16 if half else torch.float32
    x = torch.randn((B, C, H, W), generator=g).cuda().to(dt)
    buf = torch.zeros((B, 4 * C, H, W), dtype=dt, device="cuda").contiguous(memory_format=torch.channels_last)
    buf[:, :C] = x
    fn = nat.lib.bf_sppf_pool_device if half else nat.lib.bf_sppf_pool_f32_device
    assert fn(buf.data_ptr(), B, H, W, C, torch.cuda.current_stream().cuda_stream) == 0, nat.check()
    m = torch.nn.MaxPool2d(5, 1, 2)
    y1 = m(x); y2 = m(y1); y3 = m(y2)
    assert torch.equal(buf, torch.cat((x, y1, y2, y3), 1))


@pytest.mark.gpu
@pytest.mark.parametrize("half", [True, False])
def test_gpu_upsample_concat_kernel_matches_torch(native, half):
    import torch
    from lib import _native as nat
    g = torch.Generator(device="cpu").manual_seed(3)
    cl = torch.channels_last
    dt = torch.float16 if half else torch.float32
    a = torch.randn((2, 24, 6, 10), generator=g).cuda().to(dt).contiguous(memory_format=cl)
    b = torch.randn((2, 16, 12, 20), generator=g).cuda().to(dt).contiguous(memory_format=cl)
    out = torch.empty((2, 40, 12, 20), dtype=dt, device="cuda").contiguous(memory_format=cl)
    fn = nat.lib.bf_upsample_concat_device if half else nat.lib.bf_upsample_concat_f32_device
    assert fn(a.data_ptr(), b.data_ptr(), out.data_ptr(), 2, 12, 20, 24, 16, torch.cuda.current_stream().cuda_stream) == 0, nat.check()
    assert torch.equal(out, torch.cat((torch.nn.Upsample(scale_factor=2, mode="nearest")(a), b), 1))
