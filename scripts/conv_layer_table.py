#!/usr/bin/env python3
"""Per-layer table of the detector's 60 convolutions on the library's MFMA kernels (GPU box), plus the whole-forward rate per backend.

  python3 scripts/conv_layer_table.py [--batch 64] [--size 640] [--half] [--csv profiles/r03_conv_layers_f32.csv] [--no-miopen]

One forward of the HIP network is recorded (every HipConv call with its tensors), then each call is replayed alone and timed with
HIP events.  Per layer: FLOP (2 x outputs x KH KW C), compulsory bytes (inputs read once + weights + outputs written; a residual
counts as an input), time, TFLOP/s, GB/s, and the fraction of the roofline floor max(FLOP / MFMA peak, bytes / 6.3 TB/s) -- the HBM
figure is the guide's measured streaming rate, the MFMA peaks its dense f16 (2.5 PFLOP/s) and exact-f32 (157.3 TFLOP/s) figures."""
import argparse
import csv
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
import torch
from image_detection.model import yolov5s

ap = argparse.ArgumentParser()
ap.add_argument("--batch", type=int, default=64)
ap.add_argument("--size", type=int, default=640)
ap.add_argument("--half", action="store_true")
ap.add_argument("--csv", default=None)
ap.add_argument("--no-miopen", action="store_true")
ap.add_argument("--reps", type=int, default=10)
ap.add_argument("--f32-mode", choices=["split", "native"], default="split", help="float32 only: bf_conv2d_f32_mode (split = 3 bfloat16 parts per operand, 6 products: the library's default; native = v_mfma_f32_32x32x2_f32)")
a = ap.parse_args()
dt = torch.float16 if a.half else torch.float32
eb = 2 if a.half else 4
from lib import _native as nat
nat.lib.bf_conv2d_f32_mode(1 if a.f32_mode == "split" else 0)
peak = 2500e12 if a.half else (2500e12 / 6.0 if a.f32_mode == "split" else 157.3e12)       # matrix peak of the instruction mix the mode runs
HBM = 6.3e12

x3 = torch.rand((a.batch, 3, a.size, a.size), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
x4 = torch.zeros((a.batch, 4, a.size, a.size), device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
x4[:, :3] = x3


def rate(net, x, n=10):
    with torch.no_grad():
        for _ in range(3):
            net(x)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(n):
            net(x)
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


if not a.no_miopen:
    d = rate(yolov5s.build(half=a.half, conv_backend="miopen"), x3)
    print("miopen  %8.2f ms per batch of %d  = %8.0f frames/s" % (d * 1e3, a.batch, a.batch / d), flush=True)
net = yolov5s.build(half=a.half, conv_backend="hip")
d = rate(net, x4)
print("hip     %8.2f ms per batch of %d  = %8.0f frames/s  (%s)" % (d * 1e3, a.batch, a.batch / d, "float16" if a.half else "float32"), flush=True)

calls = []
names = {m: n for n, m in net.named_modules()}
plain = yolov5s.HipConv.forward


def recording(self, x, out=None, residual=None, x2=None, up=False):
    y = plain(self, x, out=out, residual=residual, x2=x2, up=up)
    calls.append((self, x, dict(out=y, residual=residual, x2=x2, up=up)))
    return y


yolov5s.HipConv.forward = recording
with torch.no_grad():
    net(x4)
yolov5s.HipConv.forward = plain
torch.cuda.synchronize()

rows = []
for m, x, kw in calls:
    y = kw["out"]
    b, n, ho, wo = (int(v) for v in y.shape)
    flop = 2.0 * b * n * ho * wo * m.kh * m.kw * m.c
    nbytes = eb * (x.numel() + (0 if kw["x2"] is None else kw["x2"].numel()) + (0 if kw["residual"] is None else kw["residual"].numel()) + y.numel() + m.n * m.kh * m.kw * m.c)
    for _ in range(2):
        plain(m, x, **kw)
    # `reps` launches captured into one HIP graph and replayed: a layer of 10-20 us is otherwise timed at the rate Python issues it
    graph = torch.cuda.CUDAGraph()
    with torch.cuda.graph(graph):
        for _ in range(a.reps):
            plain(m, x, **kw)
    graph.replay()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    graph.replay()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / a.reps
    del graph
    floor = max(flop / peak, nbytes / HBM) * 1e3
    rows.append(dict(layer=names[m], cin=m.c, cout=m.n, k=m.kh, stride=m.stride, out_hw="%dx%d" % (ho, wo), sources=1 + (kw["x2"] is not None), up=int(kw["up"]),
                     gflop=flop / 1e9, mbytes=nbytes / 1e6, ms=ms, tflops=flop / ms / 1e9, gbs=nbytes / ms / 1e6, floor_ms=floor,
                     bound="mfma" if flop / peak > nbytes / HBM else "hbm", frac=floor / ms))
tot_ms = sum(r["ms"] for r in rows); tot_flop = sum(r["gflop"] for r in rows); tot_floor = sum(r["floor_ms"] for r in rows); tot_mb = sum(r["mbytes"] for r in rows)
print("%-16s %5s %5s k s %9s  %8s %8s %8s %8s %7s %5s" % ("layer", "cin", "cout", "out", "GFLOP", "MB", "ms", "TFLOP/s", "GB/s", "frac"))
for r in rows:
    print("%-16s %5d %5d %d %d %9s  %8.1f %8.1f %8.4f %8.1f %7.0f %5.2f %s" % (r["layer"], r["cin"], r["cout"], r["k"], r["stride"], r["out_hw"], r["gflop"], r["mbytes"],
                                                                            r["ms"], r["tflops"], r["gbs"], r["frac"], r["bound"]))
print("all %d convolutions: %.3f ms, %.1f GFLOP and %.0f MB per batch of %d: %.1f TFLOP/s, %.0f GB/s; sum of the layers' roofline floors %.3f ms (%.2f of the time)"
      % (len(rows), tot_ms, tot_flop, tot_mb, a.batch, tot_flop / tot_ms, tot_mb / tot_ms * 1e3 / 1e3, tot_floor, tot_floor / tot_ms), flush=True)
if a.csv:
    with open(os.path.join(ROOT, a.csv) if not os.path.isabs(a.csv) else a.csv, "w", newline="") as f:
        w = csv.DictWriter(f, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in rows:
            w.writerow({k: ("%.6g" % v if isinstance(v, float) else v) for k, v in r.items()})
        w.writerow({"layer": "TOTAL batch %d %s" % (a.batch, "f16" if a.half else "f32 " + a.f32_mode), "gflop": "%.6g" % tot_flop, "mbytes": "%.6g" % tot_mb, "ms": "%.6g" % tot_ms,
                    "tflops": "%.6g" % (tot_flop / tot_ms), "floor_ms": "%.6g" % tot_floor, "frac": "%.6g" % (tot_floor / tot_ms)})
