#!/usr/bin/env python3
"""A few detector steps for the profiler (GPU box): python3 scripts/detector_forward.py [batch] [backend]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
import torch
from image_detection.src.yolo_smooth_tracking import Detector

B = int(sys.argv[1]) if len(sys.argv) > 1 else 64
det = Detector(conv_backend=sys.argv[2] if len(sys.argv) > 2 else None)
frames = torch.randint(0, 256, (B, 640, 640, 3), dtype=torch.uint8, device="cuda")
for _ in range(6):
    out, n = det.detect(frames)
torch.cuda.synchronize()
print(det.conv_backend, tuple(out.shape), int(n.sum()))
