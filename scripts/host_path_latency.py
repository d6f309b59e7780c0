"""Time the drop-in, host-pointer entry points (one frame in, one image out, PCIe both ways): what a caller of the
reference's mimo_pad / mimo_lerp sees per call.  Prints frames/s; used for the PCIe-inclusive figure in DESIGN.md."""
import os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, os.path.join(ROOT, "zybo-rt-sampler-image-detection_amd"))
from interface import config
from lib import _native as nat, directions
import synth

config.configure(N_MICROPHONES=64, ACTIVE_TILES=1, N_SAMPLES=256, MAX_RES_X=101, MAX_RES_Y=101, N_TAPS=8)
d = directions.calculate_delays()
mics = np.arange(64, dtype=np.int32)
sig = synth.s2_noise(64, 256)
img = np.zeros(101 * 101, dtype=np.float32)
w = np.ascontiguousarray(d.astype(int).astype(np.int32)).ravel()
f = np.ascontiguousarray(np.float32(d)).ravel()
nat.lib.load_coefficients_pad(nat.iptr(w), w.size); nat.lib.load_coefficients_lerp(nat.fptr(f), f.size); nat.check()
# each entry point twice, interleaved: the first loop after start-up also pays for the clock ramp of an idle GPU
for name, fn in (("mimo_pad", nat.lib.mimo_pad), ("mimo_lerp", nat.lib.mimo_lerp)) * 2:
    for _ in range(20):
        fn(nat.fptr(sig), nat.fptr(img), nat.iptr(mics), 64)
    nat.check()
    t0 = time.perf_counter(); n = 500
    for _ in range(n):
        fn(nat.fptr(sig), nat.fptr(img), nat.iptr(mics), 64)
    dt = time.perf_counter() - t0
    print("%s host-pointer path, cfg2 (64x256x101x101), one frame per call: %.1f us/call = %.0f frames/s" % (name, dt / n * 1e6, n / dt))
