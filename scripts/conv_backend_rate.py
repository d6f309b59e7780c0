#!/usr/bin/env python3
"""Detector forward rate with torch's (MIOpen) convolutions and with the library's implicit-GEMM kernel (GPU box).
usage: python3 scripts/conv_backend_rate.py [batch] [size]   -> frames/s of the network forward, per backend, and per-layer times of the HIP path"""
import os, sys, time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
import torch
from image_detection.model import yolov5s

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
S = int(sys.argv[2]) if len(sys.argv) > 2 else 640
x = torch.rand((B, 3, S, S), device="cuda").half().contiguous(memory_format=torch.channels_last)
FLOP = None
for backend in ("miopen", "hip"):
    net = yolov5s.build(half=True, conv_backend=backend)
    with torch.no_grad():
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            net(x)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / n
    print("%-7s %8.2f ms per batch of %d  = %8.0f frames/s" % (backend, dt * 1e3, B, B / dt), flush=True)

# per-layer: time every HipConv of the hip network, with its FLOP count
net = yolov5s.build(half=True, conv_backend="hip")
rows = []
def hook(name):
    def f(m, inp, out):
        b, c, h, w = inp[0].shape
        flop = 2.0 * out.numel() * m.kh * m.kw * m.c
        ev0, ev1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        xin = inp[0]
        for _ in range(2):
            m._forward_plain(xin)
        ev0.record()
        for _ in range(5):
            m._forward_plain(xin)
        ev1.record()
        torch.cuda.synchronize()
        rows.append((name, tuple(inp[0].shape), m.n, m.kh, m.stride, ev0.elapsed_time(ev1) / 5, flop))
    return f
for name, m in net.named_modules():
    if isinstance(m, yolov5s.HipConv):
        m._forward_plain = m.forward
        m.register_forward_hook(hook(name))
with torch.no_grad():
    net(x)
tot_ms = sum(r[5] for r in rows); tot_flop = sum(r[6] for r in rows)
for r in sorted(rows, key=lambda r: -r[5])[:12]:
    print("%-18s in %-22s -> %4d  k%d s%d  %7.3f ms  %6.1f TFLOP/s" % (r[0], r[1], r[2], r[3], r[4], r[5], r[6] / r[5] / 1e9))
print("all %d convolutions: %.2f ms, %.1f GFLOP per batch, %.1f TFLOP/s" % (len(rows), tot_ms, tot_flop / 1e9, tot_flop / tot_ms / 1e9))
