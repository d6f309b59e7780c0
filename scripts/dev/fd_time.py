#!/usr/bin/env python3
"""Time the frequency-domain maps of config 3 (64 mics, 101 x 101, 190 windows, 94 bins): phase-steer DAS and MVDR (dev tool; GPU box)."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
import numpy as np, torch
import synth
from realtime_scripts import beam_forming_algorithm as B, config as C
C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = 64, 1, 101, 101
fb = B.FrequencyBeamformer()
frames = torch.from_numpy(synth.frame_batch(64, 256, 64)).cuda().repeat(3, 1, 1)[:190].contiguous()
def timed(fn, n=10):
    for _ in range(3): fn()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(n): fn()
    torch.cuda.synchronize(); return (time.perf_counter() - t0) / n
print("das %.3f ms   mvdr %.3f ms   [BF_GEMM_F32=%s BF_FD_RT=%s]" % (timed(lambda: fb.das_power(frames)) * 1e3, timed(lambda: fb.mvdr_power(frames, 1e-2)) * 1e3,
                                                                 os.environ.get("BF_GEMM_F32", "-"), os.environ.get("BF_FD_RT", "-")))
