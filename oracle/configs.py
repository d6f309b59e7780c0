"""Shared size definitions for the oracle, the golden-vector generator and the tests.

TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is imported by the product path.

The (M, N, X, Y) tuples are BASELINE.json `configs` 1, 2 and 5 plus the reference's
as-shipped PC/src/config.json:3-11 ("shipped").  `arrays` is the number of 8x8 tiles
(`_ACTIVE_MICS` in PC/src/directions.pyx:16, hard-coded to 4 there).
"""

CONFIGS = {
    # name     mics  samples grid_x grid_y taps tiles
    "cfg1":    dict(M=64,  N=256,  X=11,  Y=11,  T=8, arrays=1),
    "cfg2":    dict(M=64,  N=256,  X=101, Y=101, T=8, arrays=1),
    "shipped": dict(M=256, N=256,  X=57,  Y=32,  T=8, arrays=4),
    "cfg5":    dict(M=256, N=1024, X=361, Y=361, T=8, arrays=4),
}

# Constants of PC/src/config.json as shipped (the only values the fixtures were generated with).
SHIPPED_JSON = dict(
    SAMPLE_RATE=48828.0, PROPAGATION_SPEED=340.0, ELEMENT_DISTANCE=0.02,
    VIEW_ANGLE=59.0, Z=1.0, ROWS=8, COLUMNS=8, SKIP_N_MICS=1,
)
