"""Dev measurement: cfg2-sized launch (64 mics x 256 samples x 101x101, 190 frames) with an UNSTRUCTURED delay table
(independent random delays): the sweep kernel's worst case, every direction step reloads its quads."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
import torch
from interface import config
from lib import _native as nat
import synth

M, N, X, Y, F = 64, 256, 101, 101, 190
config.configure(N_MICROPHONES=M, ACTIVE_TILES=1, N_SAMPLES=N, MAX_RES_X=X, MAX_RES_Y=Y, N_TAPS=8)
rng = np.random.default_rng(5)
delays = rng.uniform(0, 28.0, size=(X, Y, M))
mics = np.arange(M, dtype=np.int32)
d_sig = torch.from_numpy(np.tile(synth.frame_batch(M, N, 64), (3, 1, 1))[:F]).cuda()
d_img = torch.zeros((F, X * Y), dtype=torch.float32, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for name, algo, table, load in (("pad", nat.PAD, np.ascontiguousarray(delays.astype(int).astype(np.int32)).ravel(), "load_coefficients_pad"),
                                ("lerp", nat.LERP, np.ascontiguousarray(np.float32(delays)).ravel(), "load_coefficients_lerp")):
    getattr(nat.lib, load)(nat.iptr(table) if name == "pad" else nat.fptr(table), table.size); nat.check()
    for _ in range(3):
        nat.lib.bf_das_device(algo, d_sig.data_ptr(), M, d_img.data_ptr(), X * Y, F, nat.iptr(mics), M, 0, X * Y, s)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        nat.lib.bf_das_device(algo, d_sig.data_ptr(), M, d_img.data_ptr(), X * Y, F, nat.iptr(mics), M, 0, X * Y, s)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    nat.check()
    print("%s, unstructured table: %.2f ms per 190 frames = %.0f frames/s" % (name, dt * 1e3, F / dt))
