#!/usr/bin/env python3
"""Golden vectors for the two legacy helpers of the reference's lib.directions (directions.pyx:126-187):
`calculate_delays_()` and `calculate_delay_miso(azimuth, elevation)`.  Like gen_golden.py this runs the REAL reference
(built in a /tmp scratch directory, as-shipped sizes) and stores only data: tests/golden/legacy_directions.npz."""
import os
import shutil
import subprocess
import sys

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
import gen_golden as G  # noqa: E402

WORKER = r'''
import sys, hashlib
sys.path.insert(0, "")
import numpy as np
from lib.directions import calculate_delays_, calculate_delay_miso
d = calculate_delays_()
angles = [(0.0, 0.0), (10.0, -20.0), (-35.5, 62.0), (70.0, 70.0)]
np.savez_compressed(sys.argv[1], delays_sha256=hashlib.sha256(np.ascontiguousarray(d).tobytes()).hexdigest(), delays_shape=np.array(d.shape),
                    delays_dtype=str(d.dtype), delays_sample=d[::8, ::5, :].copy(), angles=np.array(angles),
                    miso=np.stack([calculate_delay_miso(a, e) for a, e in angles]))
print("legacy golden written", d.shape, d.dtype)
'''


def main():
    from configs import CONFIGS
    scratch = G.build_scratch("shipped", CONFIGS["shipped"])
    try:
        open(os.path.join(scratch, "legacy_worker.py"), "w").write(WORKER)
        out = os.path.join(os.path.dirname(HERE), "tests", "golden", "legacy_directions.npz")
        subprocess.check_call([sys.executable, "legacy_worker.py", out], cwd=scratch)
    finally:
        shutil.rmtree(scratch, ignore_errors=True)


if __name__ == "__main__":
    main()
