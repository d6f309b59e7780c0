#!/usr/bin/env python3
"""Experiment (GPU box): the detector forward on a batch of 64 as one launch sequence, or as S batch slices on S streams (the slices' launches fill each
other's tails).  usage: python3 scripts/dev/two_streams.py [--half]"""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
import torch
from image_detection.model import yolov5s

half = "--half" in sys.argv
dt = torch.float16 if half else torch.float32
B = 64
x = torch.zeros((B, 4, 640, 640), device="cuda", dtype=dt).contiguous(memory_format=torch.channels_last)
x[:, :3] = torch.rand((B, 3, 640, 640), device="cuda").to(dt)
net = yolov5s.build(half=half, conv_backend="hip")


def run(S):
    streams = [torch.cuda.Stream() for _ in range(S)] if S > 1 else [torch.cuda.current_stream()]
    parts = x.chunk(S, 0)

    def step():
        cur = torch.cuda.current_stream()
        outs = []
        for st, p in zip(streams, parts):
            if S > 1:
                st.wait_stream(cur)
            with torch.cuda.stream(st):
                outs.append(net(p))
        if S > 1:
            for st in streams:
                cur.wait_stream(st)
        return outs
    with torch.no_grad():
        for _ in range(3):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        n = 10
        for _ in range(n):
            step()
        torch.cuda.synchronize()
    return (time.perf_counter() - t0) / n


for S in (1, 2, 4, 1, 2, 4):
    d = run(S)
    print("%s  %d stream(s): %.2f ms per batch of %d = %.0f frames/s" % ("f16" if half else "f32", S, d * 1e3, B, B / d), flush=True)
