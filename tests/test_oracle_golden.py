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


# ---- fixtures taken from the reference's C compiled where it lies (oracle/gen_golden_refc.py -> tests/golden/refc.npz)

def test_refc_cfg5_pad_sampled_directions(oracle_lib):
    """mimo_pad at 256 mics x 1024 x 361x361: 32 sampled directions of the compiled reference's images, bit for bit."""
    name = "cfg5"
    c, g = CONFIGS[name], golden("refc")
    orc = oracle_lib.Oracle(c["N"], c["X"], c["Y"], c["T"])
    mics = np.arange(c["M"], dtype=np.int32)
    ins = util.inputs(name)
    whole = util.table_for("pad", name)
    assert sha(whole) == str(g["cfg5/whole_sha256"])
    orc.load(0, whole)
    picks = np.random.default_rng(11).choice(c["X"] * c["Y"], 32, replace=False)
    for sig in ("s1", "s2"):
        assert sha(ins[sig]) == str(g["cfg5/in_sha256_" + sig])
        want = g["cfg5/img_pad_" + sig].reshape(-1)
        for d in picks:
            assert orc.mimo_range(0, ins[sig], mics, int(d), int(d) + 1)[0] == want[d]


@pytest.mark.parametrize("name", ["cfg1", "cfg2"])
def test_refc_convolve_naive(oracle_lib, name):
    """mimo_convolve_naive (convolve_and_sum.c:231-272), which no Cython wrapper of the reference calls: the oracle's naive FIR
    order against the compiled reference -- all of cfg1, 48 sampled directions of cfg2."""
    c, g = CONFIGS[name], golden("refc")
    orc = oracle_lib.Oracle(c["N"], c["X"], c["Y"], c["T"])
    mics = np.arange(c["M"], dtype=np.int32)
    taps = np.ascontiguousarray(util.table_for("fir_naive", name), dtype=np.float32)
    assert sha(taps) == str(g[name + "/taps_sha256"])
    orc.load(3, taps)
    D = c["X"] * c["Y"]
    picks = range(D) if name == "cfg1" else np.random.default_rng(13).choice(D, 48, replace=False)
    for sig in ("s1", "s2"):
        want = g["%s/img_naive_%s" % (name, sig)].reshape(-1)
        x = util.inputs(name)[sig]
        for d in picks:
            assert orc.mimo_range(3, x, mics, int(d), int(d) + 1)[0] == want[d], (name, sig, int(d))


# ---- the oracle against oracle/_ref directly ("equals it bit for bit")

@pytest.mark.parametrize("name", ["cfg1", "shipped"])
def test_oracle_equals_compiled_reference(oracle_lib, name):
    """Same random block through both CPU engines: every entry point the two share, whole images, bit for bit.
    Skipped where oracle/_ref has not been built (it needs /root/reference: oracle/build_ref.py)."""
    if not oracle_lib.RefLib.available(name):
        pytest.skip("oracle/_ref/libref_%s.so not built" % name)
    c = CONFIGS[name]
    ref = oracle_lib.RefLib(name)
    orc = oracle_lib.Oracle(c["N"], c["X"], c["Y"], c["T"])
    rng = np.random.default_rng(17)
    sig = (rng.standard_normal((c["M"], c["N"])) * 0.25).astype(np.float32)
    mics = np.arange(c["M"], dtype=np.int32)
    d = util.oracle_delays(name)
    whole, d32 = util.table_for("pad", name), np.float32(d)
    assert orc.mimo_pad(sig, whole, mics).tobytes() == ref.mimo_pad(sig, whole, mics).tobytes()
    assert orc.mimo_lerp(sig, d32, mics).tobytes() == ref.mimo_lerp(sig, d32, mics).tobytes()
    assert orc.mimo_hybrid(sig, d32, mics).tobytes() == ref.mimo_hybrid(sig, d32, mics).tobytes()
    ow, oh = orc.lerp_tables(d32.ravel())
    rw, rh = ref.lerp_tables(d32.ravel())
    assert np.array_equal(ow, rw) and oh.tobytes() == rh.tobytes()
    ow, ot = orc.hybrid_tables(d32.ravel())
    rw, rt = ref.hybrid_tables(d32.ravel())
    assert np.array_equal(ow, rw) and ot.tobytes() == rt.tobytes()
    off = (c["X"] * c["Y"] // 3) * c["M"]
    assert orc.miso_pad(sig, whole, mics, off).tobytes() == ref.miso_pad(sig, whole, mics, off).tobytes()
    assert orc.miso_lerp(sig, d32, mics, off).tobytes() == ref.miso_lerp(sig, d32, mics, off).tobytes()
    if name == "cfg1":
        taps = np.ascontiguousarray(util.table_for("fir_vec", name), dtype=np.float32)
        for vec in (False, True):
            assert orc.mimo_convolve(sig, taps, mics, vectorized=vec).tobytes() == ref.mimo_convolve(sig, taps, mics, vectorized=vec).tobytes()
