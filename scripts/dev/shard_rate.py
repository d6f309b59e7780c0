"""Dev measurement: what ONE rank of `bench.py --gpus G` runs -- cfg2's direction shard [0, D/G) over 190 * G frames (weak
scaling: the global batch grows with G) -- so that the multi-GPU launch shapes can be timed on a one-GPU box.
usage: python scripts/dev/shard_rate.py [G ...]"""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
sys.path.insert(0, os.path.join(ROOT, "oracle"))
import torch
from interface import config
from lib import _native as nat
from lib import directions
import synth

M, N, X, Y = 64, 256, 101, 101
D = X * Y
config.configure(N_MICROPHONES=M, ACTIVE_TILES=1, N_SAMPLES=N, MAX_RES_X=X, MAX_RES_Y=Y, N_TAPS=8)
delays = directions.calculate_delays()
mics = np.arange(M, dtype=np.int32)
s = torch.cuda.current_stream().cuda_stream
for G in [int(x) for x in sys.argv[1:]] or [1, 2, 4, 8]:
    F = 190 * G
    hi = (D + G - 1) // G
    d_sig = torch.from_numpy(np.tile(synth.frame_batch(M, N, 64), ((F + 63) // 64, 1, 1))[:F]).cuda()
    d_img = torch.zeros((F, hi), dtype=torch.float32, device="cuda")
    for name, algo, table, load in (("pad", nat.PAD, np.ascontiguousarray(delays.astype(int).astype(np.int32)).ravel(), "load_coefficients_pad"),
                                    ("lerp", nat.LERP, np.ascontiguousarray(np.float32(delays)).ravel(), "load_coefficients_lerp")):
        getattr(nat.lib, load)(nat.iptr(table) if name == "pad" else nat.fptr(table), table.size); nat.check()
        for _ in range(3):
            nat.lib.bf_das_device(algo, d_sig.data_ptr(), M, d_img.data_ptr(), hi, F, nat.iptr(mics), M, 0, hi, s)
        torch.cuda.synchronize(); t0 = time.perf_counter()
        for _ in range(10):
            nat.lib.bf_das_device(algo, d_sig.data_ptr(), M, d_img.data_ptr(), hi, F, nat.iptr(mics), M, 0, hi, s)
        torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
        nat.check()
        # the rank's share of the job: F frames x hi directions; as whole-frame equivalents per second
        print("G=%d %s: shard of %d directions x %d frames: %.2f ms = %.0f whole-frame equivalents/s per rank (x%d ranks = %.0f)"
              % (G, name, hi, F, dt * 1e3, F * hi / D / dt, G, G * F * hi / D / dt))
