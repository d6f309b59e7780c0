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
@pytest.mark.parametrize("half", [False, True])
def test_gpu_decode_and_nms_match_oracle(native, half):
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
    raw_q = [r.float().cpu().numpy() for r in raw]     # what the kernel really reads
    out, n = det.postprocess(raw, conf_thres=0.1, iou_thres=0.45, max_det=300)
    out, n = out.cpu().numpy(), n.cpu().numpy()
    boxes, scores, cls = D.decode(raw_q, yolov5s.ANCHORS, yolov5s.STRIDES, nc, 0.1)
    for b in range(B):
        order = np.argsort(-scores[b], kind="stable")[:1024]
        keep = D.nms(boxes[b][order], scores[b][order], 0.45, 300)
        want = np.concatenate([boxes[b][order][keep], scores[b][order][keep][:, None]], axis=1)
        assert n[b] == len(keep) and n[b] > 3
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
@pytest.mark.parametrize("B,T,K", [(3, 25200, 1024), (64, 25200, 1024), (2, 700, 1024), (5, 4097, 300), (1, 1024, 1024)])
def test_gpu_topk_candidates_match_numpy(native, B, T, K):
    """bf_topk_candidates_device (radix select + ordered tie admission + bitonic sort, one workgroup per image) against a
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

CONV_CASES = [  # (B, Cin, H, W, Cout, k, stride, pad, silu)      the layer shapes of yolov5s.py plus ragged tiles
    (2, 3, 64, 96, 32, 6, 2, 2, True),        # the stem: 3 channels padded to 16, 6x6 window, stride 2
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
    got = yolov5s.HipConv(conv, silu)(x)
    want = torch.nn.functional.conv2d(x.float(), conv.weight.float(), conv.bias.float(), s, p)
    if silu:
        want = torch.nn.functional.silu(want)
    assert got.shape == want.shape and got.dtype == torch.float16 and got.is_contiguous(memory_format=torch.channels_last)
    err = (got.float() - want).abs().max().item() / want.abs().max().item()
    assert err < 2e-3, err


@pytest.mark.gpu
def test_gpu_network_on_hip_convolutions(native):
    """The whole YOLOv5s-shaped network with every convolution on the library's kernel: head logits agree with the fp32 forward
    to fp16 accuracy (the same bound the MIOpen path is held to), and with the MIOpen fp16 forward."""
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
        assert (a.float() - c.float()).abs().max().item() / b.abs().max().item() < 3e-2


@pytest.mark.gpu
def test_gpu_detector_backends_agree(native):
    import torch
    from image_detection.src.yolo_smooth_tracking import Detector
    frames = torch.randint(0, 256, (2, 320, 320, 3), dtype=torch.uint8, device="cuda")
    a, na = Detector(conv_backend="miopen").detect(frames, conf_thres=0.001)
    b, nb = Detector(conv_backend="hip").detect(frames, conf_thres=0.001)
    assert a.shape == b.shape and torch.equal(na > 0, nb > 0)


@pytest.mark.gpu
def test_gpu_hip_convolution_into_a_slice_with_residual(native):
    """The C3 block's use: one convolution writes the upper channel half of the concatenation buffer, another the lower half with the
    bottleneck's residual added -- equal to torch.cat((x + silu(conv_a(x)), silu(conv_b(x))), 1) computed from the same kernel's
    plain outputs (the add as torch adds two float16 tensors), bit for bit; untouched channels stay untouched."""
    import torch
    from image_detection.model import yolov5s
    g = torch.Generator(device="cpu").manual_seed(7)
    B, C, H, W = 2, 64, 24, 20
    convs = []
    for k in (3, 1):
        c = torch.nn.Conv2d(C, C, k, 1, k // 2, bias=True)
        with torch.no_grad():
            c.weight.copy_(torch.randn(c.weight.shape, generator=g) / (C * k * k) ** 0.5)
            c.bias.copy_(torch.randn(c.bias.shape, generator=g) * 0.5)
        convs.append(yolov5s.HipConv(c.cuda().half(), True))
    x = (torch.randn((B, C, H, W), generator=g) * 1.5).cuda().half().contiguous(memory_format=torch.channels_last)
    plain_a, plain_b = convs[0](x), convs[1](x)
    buf = torch.full((B, 2 * C + 8, H, W), 7.0, dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    ra = convs[0](x, out=buf[:, :C], residual=x)
    rb = convs[1](x, out=buf[:, C:2 * C])
    assert ra.data_ptr() == buf.data_ptr() and rb.data_ptr() == buf[:, C:].data_ptr()
    assert torch.equal(buf[:, :C], x + plain_a) and torch.equal(buf[:, C:2 * C], plain_b) and bool((buf[:, 2 * C:] == 7.0).all())
    with pytest.raises(Exception):
        convs[0](x, out=buf[:, :C].permute(0, 1, 3, 2))


@pytest.mark.gpu
def test_gpu_preprocess_kernel_matches_torch(native):
    """bf_preprocess_bgr8_device against frames.flip(-1).half() / 255: RGB order, the same float16 values, a zero fourth channel."""
    import torch
    from image_detection.src.yolo_smooth_tracking import Detector
    frames = torch.randint(0, 256, (3, 64, 96, 3), dtype=torch.uint8, device="cuda")
    x = Detector(conv_backend="hip").preprocess(frames)
    want = frames.flip(-1).permute(0, 3, 1, 2).half() / 255
    assert tuple(x.shape) == (3, 4, 64, 96) and x.is_contiguous(memory_format=torch.channels_last)
    assert torch.equal(x[:, :3], want) and bool((x[:, 3] == 0).all())


@pytest.mark.gpu
@pytest.mark.parametrize("B,C,H,W", [(2, 256, 20, 20), (1, 64, 12, 20), (3, 8, 5, 3)])
def test_gpu_sppf_pool_kernel_matches_torch(native, B, C, H, W):
    """bf_sppf_pool_device against nn.MaxPool2d(5, 1, 2) applied once, twice, three times: exact; the first quarter untouched."""
    import torch
    from lib import _native as nat
    g = torch.Generator(device="cpu").manual_seed(C + H)
    x = torch.randn((B, C, H, W), generator=g).cuda().half()
    buf = torch.zeros((B, 4 * C, H, W), dtype=torch.float16, device="cuda").contiguous(memory_format=torch.channels_last)
    buf[:, :C] = x
    assert nat.lib.bf_sppf_pool_device(buf.data_ptr(), B, H, W, C, torch.cuda.current_stream().cuda_stream) == 0, nat.check()
    m = torch.nn.MaxPool2d(5, 1, 2)
    y1 = m(x); y2 = m(y1); y3 = m(y2)
    assert torch.equal(buf, torch.cat((x, y1, y2, y3), 1))


@pytest.mark.gpu
def test_gpu_upsample_concat_kernel_matches_torch(native):
    import torch
    from lib import _native as nat
    g = torch.Generator(device="cpu").manual_seed(3)
    cl = torch.channels_last
    a = torch.randn((2, 24, 6, 10), generator=g).cuda().half().contiguous(memory_format=cl)
    b = torch.randn((2, 16, 12, 20), generator=g).cuda().half().contiguous(memory_format=cl)
    out = torch.empty((2, 40, 12, 20), dtype=torch.float16, device="cuda").contiguous(memory_format=cl)
    assert nat.lib.bf_upsample_concat_device(a.data_ptr(), b.data_ptr(), out.data_ptr(), 2, 12, 20, 24, 16, torch.cuda.current_stream().cuda_stream) == 0, nat.check()
    assert torch.equal(out, torch.cat((torch.nn.Upsample(scale_factor=2, mode="nearest")(a), b), 1))
