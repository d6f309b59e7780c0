"""CPU: the oracle (oracle/directions_np.py + oracle/das_oracle.c) against the golden vectors that
oracle/gen_golden.py produced by running the compiled reference.  This is what pins the oracle."""
import numpy as np
import pytest

import util
from util import CONFIGS, golden, sha


@pytest.mark.parametrize("name", list(CONFIGS))
def test_tables_bit_exact(name):
    import directions_np as D
    c, g = CONFIGS[name], golden(name)
    act, n = D.active_microphones(arrays=c["arrays"])
    assert np.array_equal(act, g["active_mics"]) and n == c["M"]
    rp = D.calc_r_prime(float(np.float32(0.02)), arrays=c["arrays"])
    assert rp.tobytes() == g["r_prime"].tobytes()
    d = util.oracle_delays(name)
    assert sha(d) == str(g["delay_sha256"])
    assert np.array_equal(d.reshape(-1, c["M"])[g["delay_rows_idx"]], g["delay_rows"])
    assert sha(D.whole_samples(d)) == str(g["whole_sha256"])
    assert sha(np.float32(d)) == str(g["delay_f32_sha256"])
    assert [d.min(), d.max()] == list(g["delay_minmax"])


def test_tap_generators_bit_exact():
    import directions_np as D
    g = golden("cfg1")
    for x, h1, h2 in zip(g["tap_probe_delay"], g["tap_probe_get_h"], g["tap_probe_get_h2"]):
        assert np.array_equal(D.get_h(float(x)), h1)
        assert np.array_equal(D.get_h2(float(x)), h2)
    d = g["delay"]
    whole, h = D.calculate_coefficients(d)
    assert np.array_equal(h, g["taps_get_h"])
    assert np.array_equal(D.compute_convolve_h(d), g["taps_get_h2"])


def test_inputs_regenerate():
    for name in CONFIGS:
        g, ins = golden(name), util.inputs(name)
        for k, v in ins.items():
            assert sha(v) == str(g["in_sha256_" + k]), (name, k)
        assert np.array_equal(ins["s1"][0], g["s1_row"])


IMAGE_CASES = [(n, a, s) for n in ("cfg1", "cfg2", "shipped") for a in ("pad", "lerp", "hybrid", "fir_vec") for s in ("s1", "s2", "s3")
               if not (a == "fir_vec" and s == "s3")]


@pytest.mark.parametrize("name,algo,sig", IMAGE_CASES)
def test_images_bit_exact(oracle_lib, name, algo, sig):
    """Full images at cfg1; at the larger sizes 48 sampled directions (the full images take minutes on one core
    and are covered on the GPU side, where the HIP path is compared with these same golden images)."""
    c, g = CONFIGS[name], golden(name)
    key = "img_%s_%s" % ({"fir_vec": "convolve"}.get(algo, algo), sig)
    want = g[key].reshape(-1)
    orc = oracle_lib.Oracle(c["N"], c["X"], c["Y"], c["T"])
    s = util.inputs(name)[sig]
    mics = np.arange(c["M"], dtype=np.int32)
    a = util.ALGOS[algo]
    orc.load(a, util.table_for(algo, name))
    D = c["X"] * c["Y"]
    if name == "cfg1":
        got = orc.mimo_range(a, s, mics, 0, D)
        assert np.array_equal(got, want)
    else:
        for d in np.random.default_rng(3).choice(D, 48, replace=False):
            got = orc.mimo_range(a, s, mics, int(d), int(d) + 1)
            assert got[0] == want[d], (name, algo, sig, int(d))


def test_cfg5_sampled_directions(oracle_lib):
    """256 mics x 1024 samples x 361x361: 16 sampled directions of the golden lerp/hybrid images."""
    name = "cfg5"
    c, g = CONFIGS[name], golden(name)
    orc = oracle_lib.Oracle(c["N"], c["X"], c["Y"], c["T"])
    mics = np.arange(c["M"], dtype=np.int32)
    ins = util.inputs(name)
    picks = np.random.default_rng(5).choice(c["X"] * c["Y"], 16, replace=False)
    orc.load(1, util.table_for("lerp", name))
    for sig in ("s1", "s2"):
        want = g["img_lerp_" + sig].reshape(-1)
        for d in picks:
            assert orc.mimo_range(1, ins[sig], mics, int(d), int(d) + 1)[0] == want[d]
