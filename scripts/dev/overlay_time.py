#!/usr/bin/env python3
"""Time bf_heatmap_overlay_device at the fused step's shape (64 frames of 640 x 640, camera blend) with HIP events (dev tool; GPU box)."""
import os, sys
import numpy as np, torch
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
from interface import config
import visual
B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
st = visual.HeatmapStream(640, 640)
small = torch.randint(0, 256, (B, config.MAX_RES_Y, config.MAX_RES_X, 3), dtype=torch.uint8, device="cuda")
cam = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, device="cuda")
for _ in range(3):
    st.overlay(small, cam)
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
for _ in range(20):
    st.overlay(small, cam)
e1.record(); torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / 20
print("overlay %d frames: %.1f us  (%.0f GB/s of camera + output bytes)" % (B, ms * 1e3, 2 * B * 640 * 640 * 3 / ms / 1e6))
