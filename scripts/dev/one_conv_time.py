#!/usr/bin/env python3
"""Time ONE convolution layer shape through HipConv with HIP events (dev tool; GPU box).
usage: python3 scripts/dev/one_conv_time.py cin cout k stride hw [--half] [--reps 50]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
import torch
from image_detection.model import yolov5s
a = [x for x in sys.argv[1:] if not x.startswith("--")]
cin, cout, k, s, hw = (int(v) for v in a[:5])
half = "--half" in sys.argv
dt = torch.float16 if half else torch.float32
conv = torch.nn.Conv2d(cin, cout, k, s, k // 2).cuda().to(dt)
hc = yolov5s.HipConv(conv, True)
x = torch.randn((64, cin, hw, hw), device="cuda").to(dt).contiguous(memory_format=torch.channels_last)
y = hc(x)
g = torch.cuda.CUDAGraph()
with torch.cuda.graph(g):
    for _ in range(20):
        hc(x, out=y)
for _ in range(2):
    g.replay()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
us = e0.elapsed_time(e1) / 20 * 1e3
flop = 2.0 * 64 * cout * y.shape[2] * y.shape[3] * k * k * cin
print("%s  %d>%d k%d s%d %dx%d: %.1f us  %.1f TFLOP/s  [%s]" % ("f16" if half else "f32", cin, cout, k, s, hw, hw, us, flop / us / 1e6, os.environ.get("BF_NATIVE_LIB", "default").split("_")[-1]))
