"""Shared helpers for the parity tests (sizes, golden vectors, oracle tables)."""
import functools
import hashlib
import os

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, "tests", "golden")

from configs import CONFIGS  # oracle/configs.py (on sys.path via conftest)

REL_TOL = 1e-5   # BASELINE.json north_star: "within 1e-5 relative for float32 beam power"

ALGOS = {"pad": 0, "lerp": 1, "hybrid": 2, "fir_naive": 3, "fir_vec": 4}


def sha(a):
    return hashlib.sha256(np.ascontiguousarray(a).tobytes()).hexdigest()


@functools.lru_cache(maxsize=None)
def golden(name):
    return np.load(os.path.join(GOLDEN, name + ".npz"))


@functools.lru_cache(maxsize=None)
def oracle_delays(name):
    """float64 [X, Y, M] from the NumPy restatement (itself pinned by test_oracle_golden)."""
    import directions_np as D
    c = CONFIGS[name]
    return D.calculate_delays(c["X"], c["Y"], arrays=c["arrays"])


def inputs(name):
    """The S1/S2/S3 blocks of a size, regenerated from their definitions and checked against the golden hashes."""
    import synth
    c = CONFIGS[name]
    g = golden(name)
    d = oracle_delays(name)
    x0, y0 = [int(v) for v in g["s3_dir"]]
    out = {"s1": synth.s1_tone(c["M"], c["N"]), "s2": synth.s2_noise(c["M"], c["N"]), "s3": synth.s3_plane_wave(d[x0, y0], c["N"])}
    return out


def table_for(algo, name):
    """The table each reference wrapper loads (benchmark.pyx): int32 whole for pad, float32 delays for lerp/hybrid,
    get_h2 taps of the full delay for the two FIR flavours."""
    import directions_np as D
    d = oracle_delays(name)
    if algo == "pad":
        return D.whole_samples(d)
    if algo in ("lerp", "hybrid"):
        return np.float32(d)
    g = golden(name)
    if "taps_get_h2" in g.files:
        return g["taps_get_h2"]
    return D.compute_convolve_h(d, CONFIGS[name]["T"])


def max_rel(got, want):
    got = np.asarray(got, dtype=np.float64).ravel()
    want = np.asarray(want, dtype=np.float64).ravel()
    denom = np.maximum(np.abs(want), np.finfo(np.float32).tiny)
    return float(np.max(np.abs(got - want) / denom))


def configure(name, **extra):
    """Switch the product package (interface.config + native library) to a size of oracle/configs.py."""
    from interface import config
    c = CONFIGS[name]
    kw = dict(N_MICROPHONES=c["M"], ACTIVE_TILES=c["arrays"], N_SAMPLES=c["N"], MAX_RES_X=c["X"], MAX_RES_Y=c["Y"], N_TAPS=c["T"])
    kw.update(extra)
    config.configure(**kw)
    return c
