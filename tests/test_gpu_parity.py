"""GPU (-m gpu): the HIP path, called through the C-ABI, against the oracle and the golden vectors.

Bar (BASELINE.json north_star): steering tables bit-exact; float32 beam power within 1e-5 relative.
The raw steered blocks out_d[k] (miso_*) are additionally required to be BIT-IDENTICAL to the oracle, because the
kernels keep the reference's mic order and operation order; only the final sum over k is a tree."""
import ctypes as C

import numpy as np
import pytest

import util
from util import ALGOS, CONFIGS, REL_TOL, golden, max_rel

pytestmark = pytest.mark.gpu


@pytest.fixture(scope="module")
def nat(native):
    assert native.gpu_available(), "these tests need the MI355X"
    return native


def run_product(nat, algo, table, sig, mics):
    """load_coefficients_* + mimo_* of include/beamformer_hip.h with host pointers -> float32 [D]."""
    from interface import config
    a = ALGOS[algo]
    sig = nat.f32c(sig)
    mics = np.ascontiguousarray(mics, dtype=np.int32)
    img = np.zeros(config.MAX_RES_X * config.MAX_RES_Y, dtype=np.float32)
    if a == 0:
        t = np.ascontiguousarray(table, dtype=np.int32).ravel()
        nat.lib.load_coefficients_pad(nat.iptr(t), t.size)
    else:
        t = np.ascontiguousarray(table, dtype=np.float32).ravel()
        {1: nat.lib.load_coefficients_lerp, 2: nat.lib.load_coefficients_convolve_hybrid}.get(a, nat.lib.load_coefficients_convolve)(nat.fptr(t), t.size)
    nat.check()
    fn = [nat.lib.mimo_pad, nat.lib.mimo_lerp, nat.lib.mimo_convolve_hybrid, nat.lib.mimo_convolve_naive, nat.lib.mimo_convolve_vectorized][a]
    fn(nat.fptr(sig), nat.fptr(img), nat.iptr(mics), mics.size)
    nat.check()
    return img


# ------------------------------------------------------------------ golden images through the reference-named wrappers

GOLD_CASES = [(n, a, s) for n in ("cfg1", "cfg2", "shipped") for a in ("pad", "lerp", "hybrid", "convolve") for s in ("s1", "s2", "s3")
              if not (a == "convolve" and s == "s3")]
GOLD_CASES += [("cfg5", "lerp", "s1"), ("cfg5", "lerp", "s2"), ("cfg5", "hybrid", "s2")]

# Images taken from the reference's own C compiled where it lies (oracle/gen_golden_refc.py -> tests/golden/refc.npz): the sizes /
# flavours its Cython wrappers cannot reach (pad at cfg5: hours of Python tap loops; mimo_convolve_naive: no wrapper calls it).
REFC_CASES = [("cfg5", "pad", "s1"), ("cfg5", "pad", "s2")] + [(n, "naive", s) for n in ("cfg1", "cfg2") for s in ("s1", "s2")]


@pytest.mark.parametrize("name,algo,sig", REFC_CASES)
def test_matches_compiled_reference_c(nat, name, algo, sig):
    """mimo_pad at 256 mics x 1024 x 361x361 and mimo_convolve_naive, through the C-ABI with host pointers, against
    PC/src/algorithms/{pad_and_sum.c:100-143, convolve_and_sum.c:231-272} compiled by oracle/build_ref.py."""
    c = util.configure(name)
    g = golden("refc")
    x = util.inputs(name)[sig]
    assert util.sha(x) == str(g["%s/in_sha256_%s" % (name, sig)])
    mics = np.arange(c["M"], dtype=np.int32)
    if algo == "pad":
        table = util.table_for("pad", name)
        assert util.sha(table) == str(g[name + "/whole_sha256"])
        got = run_product(nat, "pad", table, x, mics)
    else:
        taps = np.ascontiguousarray(util.table_for("fir_naive", name), dtype=np.float32)
        assert util.sha(taps) == str(g[name + "/taps_sha256"])
        got = run_product(nat, "fir_naive", taps, x, mics)
    want = g["%s/img_%s_%s" % (name, algo, sig)]
    got = got.reshape(want.shape)
    assert np.isfinite(got).all()
    assert max_rel(got, want) <= REL_TOL
    if algo == "pad" or name == "cfg1":
        assert got.tobytes() == want.tobytes()          # same operation order end to end: bit-identical


@pytest.mark.parametrize("name,algo,sig", GOLD_CASES)
def test_wrappers_match_golden(nat, name, algo, sig):
    """lib.tests.mimo_*_wrapper (PC/src/benchmark.pyx names) vs images produced by the compiled reference."""
    from lib import tests as T
    util.configure(name)
    fn = {"pad": T.mimo_pad_wrapper, "lerp": T.mimo_lerp_wrapper, "hybrid": T.mimo_hybrid_convolve_wrapper, "convolve": T.mimo_convolve_wrapper}[algo]
    got = fn(util.inputs(name)[sig])
    want = golden(name)["img_%s_%s" % (algo, sig)]
    assert got.shape == want.shape and got.dtype == np.float32
    assert np.isfinite(got).all()
    assert max_rel(got, want) <= REL_TOL
    if algo != "convolve" or name == "cfg1":
        # stronger than the bar: the kernels keep the reference's operation order end to end (mic order, fma
        # placement, k-ordered power sum), so the maps are bit-identical to the compiled reference's
        assert got.tobytes() == want.tobytes()
    top = np.sort(want.ravel())[-2:]
    if top[1] - top[0] > 4 * REL_TOL * top[1]:        # the peak is unambiguous at the tolerance: it must be in the same pixel
        assert np.unravel_index(np.argmax(got), got.shape) == np.unravel_index(np.argmax(want), want.shape)


def test_known_answer_tone(nat):
    """PC/plot.py's input: identical 8 kHz tone in every mic -> maximum at the grid centre, peak ~0.5."""
    from lib import tests as T
    import synth
    util.configure("shipped")
    img = T.mimo_pad_wrapper(synth.s1_tone(256, 256))
    assert np.unravel_index(np.argmax(img), img.shape)[0] == 28
    assert abs(float(img.max()) - 0.50014573) < 1e-6


# ------------------------------------------------------------------ loaders

@pytest.mark.parametrize("name", ["cfg1", "cfg2", "shipped"])
def test_loader_tables_bit_exact(nat, oracle_lib, name):
    """What load_coefficients_lerp / _convolve_hybrid leave on the GPU == the oracle's (== the reference's) tables."""
    c = util.configure(name)
    d32 = np.ascontiguousarray(np.float32(util.oracle_delays(name))).ravel()
    orc = oracle_lib.Oracle(c["N"], c["X"], c["Y"], c["T"])
    nat.lib.load_coefficients_lerp(nat.fptr(d32), d32.size); nat.check()
    w = np.zeros(d32.size, dtype=np.int32); h = np.zeros(d32.size, dtype=np.float32)
    assert nat.lib.bf_get_lerp_tables(nat.iptr(w), nat.fptr(h), d32.size) == 0
    ow, oh = orc.lerp_tables(d32)
    assert np.array_equal(w, ow) and h.tobytes() == oh.tobytes()
    nat.lib.load_coefficients_convolve_hybrid(nat.fptr(d32), d32.size); nat.check()
    taps = np.zeros((d32.size, c["T"]), dtype=np.float32)
    assert nat.lib.bf_get_hybrid_tables(nat.iptr(w), nat.fptr(taps), d32.size) == 0
    ow, ot = orc.hybrid_tables(d32)
    assert np.array_equal(w, ow) and taps.tobytes() == ot.tobytes()


# ------------------------------------------------------------------ oracle comparisons on shapes the fixtures do not hold

def _random_case(rng, M_total, n_active, N, X, Y, T, pmax):
    sig = (rng.standard_normal((M_total, N)) * 0.25).astype(np.float32)
    mics = np.sort(rng.choice(M_total, n_active, replace=False)).astype(np.int32)
    delays = rng.uniform(0, pmax, size=(X, Y, n_active))
    taps = rng.uniform(-0.5, 0.5, size=(X, Y, n_active, T)).astype(np.float32)
    return sig, mics, delays, taps


EDGE = [
    # M_total n_active   N    X   Y   T  pmax   what it exercises
    (64,   64,  256, 11, 11,  8,  12.0),   # the reference CPU case
    (64,   37,  256,  7,  5,  8,  30.0),   # subset of mics, n not a power of two (true division by n)
    (16,   16,  200,  5,  3,  8,  20.0),   # N not a multiple of 64 (masked tail)
    (8,     5,   64,  3,  3,  8,  63.9),   # delays up to the block length
    (8,     8,   37,  2,  2,  8,  50.0),   # delays beyond the block: those mics contribute nothing
    (4,     3, 1000,  3,  2,  8, 100.0),   # long block, 16 segments
    (300, 280,  256,  4,  3,  8,  47.0),   # mic block larger than LDS: chunked staging, DPW accumulators
    (2,     1,  128,  1,  1,  8,   0.0),   # 1x1 grid, one mic, zero delay
    (32,   32,  256,  4,  4, 16,   9.0),   # 16 taps (two AVX blocks in the vectorized FIR order)
    (16,   16,  256,  3,  3,  8, 255.5),   # delays up to the block length at N = 256: zero prefix longer than one wave's quads
    (24,   20,  512,  5,  4,  8,  60.0),   # two 256-sample segments in the shifted-copies kernel
    (8,     8,  258,  2,  2,  8,  10.0),   # second segment almost empty, N not a multiple of 4 (scalar staging loads)
    (12,    9,  450,  3,  3,  8, 200.0),   # delays past the fixed-stride prefix: run-time row stride, partial chunk
    (40,   33, 1024,  3,  2,  8, 300.0),   # four segments, several chunks (pad stages two (mic, segment) pairs per wave)
    (48,   24, 1024,  5,  4,  8,  40.0),   # das_long_kernel: four segments, six halves of 4 mics, n not a power of two (true division)
    (32,   16,  512,  9,  8,  8,  60.0),   # das_long_kernel: two segments, 72 directions = one partial wave group
    (16,   16, 1000,  3,  3,  8, 300.0),   # long rows with delays past the fixed prefix: run-time row stride (pad: long kernel; lerp: its LDS image does not fit -> das_copies_kernel)
    (64,   64,  700, 17, 16,  8,  20.0),   # das_long_kernel: N not a multiple of 256 (third segment partial), 272 directions = five wave groups of 64
    (32,   32,  516, 10,  8,  8,  30.0),   # das_long_kernel with four segments of which two lie wholly beyond the block (k0 = 768 > N): every staging load stays inside the mic's row
]


@pytest.mark.parametrize("case", EDGE, ids=lambda c: "M%d_n%d_N%d_%dx%d_T%d" % c[:6])
@pytest.mark.parametrize("algo", list(ALGOS))
def test_edge_shapes_match_oracle(nat, oracle_lib, algo, case):
    from interface import config
    M_total, n_active, N, X, Y, T, pmax = case
    rng = np.random.default_rng(hash((algo,) + case) % (2 ** 32))
    sig, mics, delays, taps = _random_case(rng, M_total, n_active, N, X, Y, T, pmax)
    config.configure(N_MICROPHONES=M_total, N_SAMPLES=N, MAX_RES_X=X, MAX_RES_Y=Y, N_TAPS=T)
    table = {"pad": delays.astype(int).astype(np.int32), "lerp": np.float32(delays), "hybrid": np.float32(delays)}.get(algo, taps)
    orc = oracle_lib.Oracle(N, X, Y, T)
    orc.load(ALGOS[algo], table)
    want = orc.mimo_range(ALGOS[algo], sig, mics, 0, X * Y)
    got = run_product(nat, algo, table, sig, mics)
    assert np.isfinite(got).all()
    assert max_rel(got, want) <= REL_TOL, (got, want)


@pytest.mark.parametrize("algo", ["pad", "lerp", "hybrid"])
def test_reloading_a_table_of_the_same_shape_takes_effect(nat, oracle_lib, algo):
    """load_coefficients_* twice with different delays and unchanged sizes (a steering update): the second table must be
    the one used (the kernels work from a digest derived from the table, which has to be rebuilt)."""
    from interface import config
    M, N, X, Y, T = 64, 256, 9, 7, 8
    config.configure(N_MICROPHONES=M, N_SAMPLES=N, MAX_RES_X=X, MAX_RES_Y=Y, N_TAPS=T)
    rng = np.random.default_rng(11)
    sig = (rng.standard_normal((M, N)) * 0.25).astype(np.float32)
    mics = np.arange(M, dtype=np.int32)
    orc = oracle_lib.Oracle(N, X, Y, T)
    for scale in (10.0, 23.0):
        delays = rng.uniform(0, scale, size=(X, Y, M))
        table = delays.astype(int).astype(np.int32) if algo == "pad" else np.float32(delays)
        orc.load(ALGOS[algo], table)
        want = orc.mimo_range(ALGOS[algo], sig, mics, 0, X * Y)
        got = run_product(nat, algo, table, sig, mics)
        assert max_rel(got, want) <= REL_TOL, scale


def test_zero_signal_and_empty_contribution(nat):
    """All-zero block -> exactly zero power; every delay >= N -> exactly zero power."""
    from interface import config
    config.configure(N_MICROPHONES=8, N_SAMPLES=128, MAX_RES_X=3, MAX_RES_Y=3, N_TAPS=8)
    mics = np.arange(8, dtype=np.int32)
    z = np.zeros((8, 128), dtype=np.float32)
    table = np.full((3, 3, 8), 5, dtype=np.int32)
    assert not run_product(nat, "pad", table, z, mics).any()
    s = np.ones((8, 128), dtype=np.float32)
    assert not run_product(nat, "pad", np.full((3, 3, 8), 4000, dtype=np.int32), s, mics).any()
    assert not run_product(nat, "lerp", np.full((3, 3, 8), 127.5, dtype=np.float32), s, mics).any()


def test_bad_arguments_are_reported(nat):
    from interface import config
    config.configure(N_MICROPHONES=8, N_SAMPLES=128, MAX_RES_X=3, MAX_RES_Y=3, N_TAPS=8)
    s = np.ones((8, 128), dtype=np.float32)
    img = np.zeros(9, dtype=np.float32)
    mics = np.arange(8, dtype=np.int32)
    t = np.full(9 * 8, -1, dtype=np.int32)                      # negative delay: the reference would write out of bounds
    nat.lib.load_coefficients_pad(nat.iptr(t), t.size)
    with pytest.raises(nat.BeamformerError, match="negative delay"):
        nat.check()
    t = np.zeros(9 * 8, dtype=np.int32)
    nat.lib.load_coefficients_pad(nat.iptr(t), t.size); nat.check()
    nat.lib.mimo_pad(nat.fptr(s), nat.fptr(img), nat.iptr(mics), 5)   # table was loaded for n = 8
    with pytest.raises(nat.BeamformerError, match="loaded"):
        nat.check()
    assert np.isnan(img).all()
    nat.lib.unload_coefficients_pad()
    nat.lib.mimo_pad(nat.fptr(s), nat.fptr(img), nat.iptr(mics), 8)
    with pytest.raises(nat.BeamformerError, match="has not been called"):
        nat.check()


# ------------------------------------------------------------------ MISO and the single-signal helpers: bit-exact out[k]

@pytest.mark.parametrize("name", ["cfg1", "shipped"])
def test_miso_bit_exact(nat, oracle_lib, name):
    c = util.configure(name)
    M, N = c["M"], c["N"]
    sig = util.inputs(name)["s3"]
    mics = np.arange(M, dtype=np.int32)
    d = util.oracle_delays(name)
    whole, d32 = util.table_for("pad", name).ravel(), np.ascontiguousarray(np.float32(d)).ravel()
    orc = oracle_lib.Oracle(N, c["X"], c["Y"], c["T"])
    nat.lib.load_coefficients_pad(nat.iptr(whole), whole.size)
    nat.lib.load_coefficients_lerp(nat.fptr(d32), d32.size)
    nat.lib.load_coefficients_convolve_hybrid(nat.fptr(d32), d32.size)
    nat.check()
    out = np.zeros(N, dtype=np.float32)
    for direction in (0, 7, c["X"] * c["Y"] // 2, c["X"] * c["Y"] - 1):
        off = direction * M
        nat.lib.miso_pad(nat.fptr(sig), nat.fptr(out), nat.iptr(mics), M, off); nat.check()
        assert out.tobytes() == orc.miso_pad(sig, whole, mics, off).tobytes()
        nat.lib.miso_lerp(nat.fptr(sig), nat.fptr(out), nat.iptr(mics), M, off); nat.check()
        assert out.tobytes() == orc.miso_lerp(sig, d32, mics, off).tobytes()


def test_delay_helpers_bit_exact(nat, oracle_lib):
    from interface import config
    N, T = 256, 8
    config.configure(N_MICROPHONES=4, N_SAMPLES=N, MAX_RES_X=2, MAX_RES_Y=2, N_TAPS=T)
    rng = np.random.default_rng(11)
    sig = rng.standard_normal(N).astype(np.float32)
    base = rng.standard_normal(N).astype(np.float32)
    h = rng.uniform(-1, 1, T).astype(np.float32)
    orc = oracle_lib.Oracle(N, 2, 2, T)
    F, I = oracle_lib._f, oracle_lib._i

    def both(native_call, oracle_call):
        a, b = base.copy(), base.copy()
        native_call(a); nat.check()
        oracle_call(b)
        assert a.tobytes() == b.tobytes()

    both(lambda o: nat.lib.pad_delay(nat.fptr(sig), nat.fptr(o), 17), lambda o: orc._fn("pad_delay")(F(sig), F(o), 17))
    both(lambda o: nat.lib.lerp_delay(nat.fptr(sig), nat.fptr(o), C.c_float(0.3125), 9),
         lambda o: orc._fn("lerp_delay")(F(sig), F(o), C.c_float(0.3125), 9))
    both(lambda o: nat.lib.convolve_delay_naive(nat.fptr(sig), nat.fptr(o), nat.fptr(h)), lambda o: orc._fn("convolve_delay_naive")(F(sig), F(o), F(h)))
    both(lambda o: nat.lib.convolve_delay_vectorized_add(nat.fptr(sig), nat.fptr(h), nat.fptr(o)),
         lambda o: orc._fn("convolve_delay_vectorized_add")(F(sig), F(h), F(o)))
    both(lambda o: nat.lib.convolve_hybrid_delay_add(nat.fptr(sig), nat.fptr(h), 5, nat.fptr(o)),
         lambda o: orc._fn("convolve_hybrid_delay_add")(F(sig), F(h), 5, F(o)))


# ------------------------------------------------------------------ device-resident batched entry point

def _torch():
    import torch
    assert torch.cuda.is_available()
    return torch


def batch_variant(algo, frames):
    """bf_last_das_variant() of a batch on geometric tables at N <= 256, M % 16 == 0 (include/beamformer_hip.h)"""
    return 8 if algo == "lerp" else 5


@pytest.mark.parametrize("algo", ["pad", "lerp"])
def test_batched_device_path(nat, algo):
    """bf_das_device on HBM-resident frames: every frame equals the host-pointer call; direction shards equal slices."""
    torch = _torch()
    import synth
    c = util.configure("cfg2")
    M, N, D = c["M"], c["N"], c["X"] * c["Y"]
    F = 6
    frames = synth.frame_batch(M, N, F)
    mics = np.arange(M, dtype=np.int32)
    table = util.table_for(algo, "cfg2")
    ref_imgs = np.stack([run_product(nat, algo, table, frames[f], mics) for f in range(F)])
    d_sig = torch.from_numpy(frames).cuda()
    d_img = torch.full((F, D), float("nan"), dtype=torch.float32, device="cuda")
    stream = torch.cuda.current_stream().cuda_stream
    assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, d_img.data_ptr(), D, F, nat.iptr(mics), M, 0, D, stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(d_img.cpu().numpy(), ref_imgs)
    assert nat.lib.bf_last_das_variant() == batch_variant(algo, F)   # geometric tables, a batch: the two-frame sweep (lerp: frame-interleaved rows)
    # two direction shards, as two ranks would compute them
    cut = 5003
    lo = torch.full((F, cut), float("nan"), dtype=torch.float32, device="cuda")
    hi = torch.full((F, D - cut), float("nan"), dtype=torch.float32, device="cuda")
    assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, lo.data_ptr(), cut, F, nat.iptr(mics), M, 0, cut, stream) == 0
    assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, hi.data_ptr(), D - cut, F, nat.iptr(mics), M, cut, D, stream) == 0
    torch.cuda.synchronize()
    assert np.array_equal(torch.cat([lo, hi], dim=1).cpu().numpy(), ref_imgs)


@pytest.mark.parametrize("cfg,n_active,F", [("cfg2", 64, 5), ("cfg2", 64, 2), ("cfg2", 32, 3), ("shipped", 256, 3), ("cfg2", 64, 4), ("cfg2", 48, 7), ("shipped", 256, 6)])
@pytest.mark.parametrize("algo", ["pad", "lerp"])
def test_batched_frame_pairs(nat, oracle_lib, algo, cfg, n_active, F):
    """The two-frames-per-workgroup kernel: odd frame counts (the last workgroup row owns a single frame), a subset of the
    microphone rows (adaptive_array), the as-shipped 256-microphone array (16 chunks): bit-identical to the CPU oracle, frame by frame, and to the one-frame kernel."""
    torch = _torch()
    import synth
    c = util.configure(cfg)
    M, N, X, Y = c["M"], c["N"], c["X"], c["Y"]
    D = X * Y
    frames = synth.frame_batch(M, N, F)
    mics = (np.arange(n_active) * (M // n_active)).astype(np.int32)     # rows of the frame the active slots read
    table = np.ascontiguousarray(np.asarray(util.table_for(algo, cfg)).reshape(X, Y, M)[..., :n_active])
    orc = oracle_lib.Oracle(N, X, Y, 8)
    orc.load(ALGOS[algo], table)
    one = run_product(nat, algo, table, frames[F - 1], mics)   # loads the table; one frame: the one-frame kernel
    assert nat.lib.bf_last_das_variant() == 2
    d_sig = torch.from_numpy(frames).cuda()
    d_img = torch.full((F, D), float("nan"), dtype=torch.float32, device="cuda")
    assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, d_img.data_ptr(), D, F, nat.iptr(mics), n_active, 0, D,
                                 torch.cuda.current_stream().cuda_stream) == 0, nat.check()
    torch.cuda.synchronize()
    assert nat.lib.bf_last_das_variant() == batch_variant(algo, F)
    got = d_img.cpu().numpy()
    assert np.array_equal(got[F - 1], one.reshape(-1))
    for f in range(F):
        assert np.array_equal(got[f], orc.mimo_range(ALGOS[algo], frames[f], mics, 0, D).reshape(-1)), f


@pytest.mark.parametrize("cfg,n_active,F", [("cfg2", 64, 3), ("cfg2", 32, 2), ("shipped", 256, 3), ("cfg1", 64, 5)])
@pytest.mark.parametrize("algo", ["hybrid", "fir_naive", "fir_vec"])
def test_batched_hybrid_frame_pairs(nat, oracle_lib, algo, cfg, n_active, F):
    """The 8-tap FIR flavours' two-frame sweep on frame-interleaved rows (packed multiply-accumulates with scalar taps; hybrid: windows
    shared between neighbouring directions, guard masks rebuilt with the window; naive / vectorized-order: one window per mic):
    odd frame counts, a subset of the microphone rows, the as-shipped 256-microphone array, a grid smaller than one workgroup
    pass -- bit-identical to the CPU oracle, frame by frame, and to the one-frame kernel."""
    torch = _torch()
    import synth
    c = util.configure(cfg)
    M, N, X, Y = c["M"], c["N"], c["X"], c["Y"]
    D = X * Y
    frames = synth.frame_batch(M, N, F)
    mics = (np.arange(n_active) * (M // n_active)).astype(np.int32)
    if algo != "hybrid" and cfg == "shipped":
        # the full-FIR tap table of the 256-mic array: the oracle's Python loop (get_h2 per entry, directions.pyx:207-247) takes minutes
        # at this size, the library's batch form of the same function (bf_get_h2_batch, bit-exact against the golden taps in
        # tests/test_host_side.py) milliseconds -- both kernel and oracle are then fed the same table
        from lib import directions
        full = directions.compute_convolve_h()
    else:
        full = np.asarray(util.table_for(algo, cfg))
    table = np.ascontiguousarray(full.reshape(X, Y, M)[..., :n_active] if algo == "hybrid" else full.reshape(X, Y, M, 8)[:, :, :n_active, :])
    orc = oracle_lib.Oracle(N, X, Y, 8)
    orc.load(ALGOS[algo], table)
    one = run_product(nat, algo, table, frames[F - 1], mics)   # loads the table; one frame: the direction-outer kernel
    assert nat.lib.bf_last_das_variant() == 4
    d_sig = torch.from_numpy(frames).cuda()
    d_img = torch.full((F, D), float("nan"), dtype=torch.float32, device="cuda")
    assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, d_img.data_ptr(), D, F, nat.iptr(mics), n_active, 0, D,
                                 torch.cuda.current_stream().cuda_stream) == 0, nat.check()
    torch.cuda.synchronize()
    assert nat.lib.bf_last_das_variant() == 7
    got = d_img.cpu().numpy()
    assert np.array_equal(got[F - 1], one.reshape(-1))
    for f in range(F):
        assert np.array_equal(got[f], orc.mimo_range(ALGOS[algo], frames[f], mics, 0, D).reshape(-1)), f
    # a direction shard, as a rank of the multi-GPU launch computes it
    lo, hi = D // 3, D // 3 + max(1, D // 5)
    part = torch.full((F, hi - lo), float("nan"), dtype=torch.float32, device="cuda")
    assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, part.data_ptr(), hi - lo, F, nat.iptr(mics), n_active, lo, hi,
                                 torch.cuda.current_stream().cuda_stream) == 0, nat.check()
    torch.cuda.synchronize()
    assert np.array_equal(part.cpu().numpy(), got[:, lo:hi])


@pytest.mark.parametrize("algo", ["pad", "lerp"])
def test_batched_frame_pairs_short_block(nat, oracle_lib, algo):
    """The pair kernel on a block that does not fill its 256-sample rows (N = 200) and a grid that is not a multiple of the
    128 directions a workgroup pass carries: bit-identical to the CPU oracle."""
    torch = _torch()
    from interface import config
    from lib import directions
    M, N, X, Y, T, F = 64, 200, 37, 31, 8, 3
    config.configure(N_MICROPHONES=M, ACTIVE_TILES=1, N_SAMPLES=N, MAX_RES_X=X, MAX_RES_Y=Y, N_TAPS=T)
    try:
        delays = directions.calculate_delays()
        table = delays.astype(int).astype(np.int32) if algo == "pad" else np.float32(delays)
        rng = np.random.default_rng(11)
        frames = (rng.standard_normal((F, M, N)) * 0.25).astype(np.float32)
        mics = np.arange(M, dtype=np.int32)
        orc = oracle_lib.Oracle(N, X, Y, T)
        orc.load(ALGOS[algo], table)
        run_product(nat, algo, table, frames[0], mics)          # loads the table into the library
        D = X * Y
        d_sig = torch.from_numpy(frames).cuda()
        d_img = torch.full((F, D), float("nan"), dtype=torch.float32, device="cuda")
        assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, d_img.data_ptr(), D, F, nat.iptr(mics), M, 0, D,
                                     torch.cuda.current_stream().cuda_stream) == 0, nat.check()
        torch.cuda.synchronize()
        assert nat.lib.bf_last_das_variant() == batch_variant(algo, F)
        got = d_img.cpu().numpy()
        for f in range(F):
            assert np.array_equal(got[f], orc.mimo_range(ALGOS[algo], frames[f], mics, 0, D).reshape(-1)), f
    finally:
        util.configure("cfg1")


@pytest.mark.parametrize("cfg,algo,F", [("cfg2", "hybrid", 3), ("cfg5", "lerp", 2), ("cfg5", "pad", 3)])
def test_batched_large_tables_frame_inner_mapping(nat, cfg, algo, F):
    """Tables beyond the L2s (cfg2's 21 MB of hybrid taps, cfg5's 268 MB digest): the launch maps workgroup ids so that an
    XCD runs all frames of a direction tile back to back.  Every frame of the batch must equal the one-frame call."""
    torch = _torch()
    import synth
    c = util.configure(cfg)
    try:
        M, N, D = c["M"], c["N"], c["X"] * c["Y"]
        frames = synth.frame_batch(M, N, F)
        mics = np.arange(M, dtype=np.int32)
        table = util.table_for(algo, cfg)
        ref = np.stack([run_product(nat, algo, table, frames[f], mics) for f in range(F)])
        d_sig = torch.from_numpy(frames).cuda()
        d_img = torch.full((F, D), float("nan"), dtype=torch.float32, device="cuda")
        assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, d_img.data_ptr(), D, F, nat.iptr(mics), M, 0, D,
                                     torch.cuda.current_stream().cuda_stream) == 0, nat.check()
        torch.cuda.synchronize()
        assert np.array_equal(d_img.cpu().numpy(), ref.reshape(F, D))
    finally:
        util.configure("cfg1")


def test_full_size_properties(nat):
    """BASELINE cfg2 at bench size (190 frames): size-independent properties of the power map.
       * scaling a block by 2 scales its map by exactly 4 (power-of-two scaling is exact in float32);
       * frames are independent: permuting the batch permutes the maps;
       * a batch of identical frames gives identical maps."""
    torch = _torch()
    import synth
    c = util.configure("cfg2")
    M, N, D, F = c["M"], c["N"], c["X"] * c["Y"], 190
    mics = np.arange(M, dtype=np.int32)
    d32 = np.ascontiguousarray(util.table_for("lerp", "cfg2")).ravel()
    nat.lib.load_coefficients_lerp(nat.fptr(d32), d32.size); nat.check()
    frames = torch.from_numpy(synth.frame_batch(M, N, F)).cuda()
    stream = torch.cuda.current_stream().cuda_stream

    def run(x):
        out = torch.empty((x.shape[0], D), dtype=torch.float32, device="cuda")
        assert nat.lib.bf_das_device(ALGOS["lerp"], x.data_ptr(), M, out.data_ptr(), D, x.shape[0], nat.iptr(mics), M, 0, D, stream) == 0
        torch.cuda.synchronize()
        return out

    base = run(frames)
    assert torch.isfinite(base).all() and (base >= 0).all()
    assert torch.equal(run(frames * 2.0), base * 4.0)
    perm = torch.randperm(F, device="cuda")
    assert torch.equal(run(frames[perm].contiguous()), base[perm])
    same = run(frames[:1].expand(F, M, N).contiguous())
    assert torch.equal(same, base[:1].expand(F, D))


def test_api_shims(nat):
    """pad_mimo / mimo_truncated / miso_steer_listen (PC/src/api.c:951-1104) on a frame handed over with bf_publish_frame:
    pad_mimo sees the frame with the reference's 122 dead-microphone rows zeroed (api.c:835-858), mimo_truncated does not."""
    c = util.configure("shipped")
    M, N, D = c["M"], c["N"], c["X"] * c["Y"]
    sig = util.inputs("shipped")["s2"]
    mics = np.arange(M, dtype=np.int32)
    whole = util.table_for("pad", "shipped").ravel()
    nat.lib.load_coefficients_pad(nat.iptr(whole), whole.size)
    nat.lib.load_coefficients2(nat.iptr(whole), whole.size)
    nat.lib.bf_publish_frame(nat.fptr(sig)); nat.check()
    seen = np.zeros_like(sig)
    nat.lib.get_data(nat.fptr(seen)); nat.check()
    dead = np.flatnonzero(~seen.any(axis=1))
    assert dead.size == 122 and dead[0] == 0 and dead[-1] == 201
    img_a, img_b, img_c = (np.zeros(D, dtype=np.float32) for _ in range(3))
    nat.lib.pad_mimo(nat.fptr(img_a), nat.iptr(mics), M); nat.check()
    nat.lib.mimo_pad(nat.fptr(seen), nat.fptr(img_b), nat.iptr(mics), M); nat.check()
    assert np.array_equal(img_a, img_b)
    nat.lib.mimo_truncated(nat.fptr(img_c), nat.iptr(mics), M); nat.check()
    nat.lib.mimo_pad(nat.fptr(sig), nat.fptr(img_b), nat.iptr(mics), M); nat.check()
    assert np.array_equal(img_c, img_b)


def test_beamformer_module_surface(nat, oracle_lib):
    """lib.beamformer (PC/src/main.pyx names): publish -> receive, producer loop `b`, steering offset, MISO listen."""
    import queue as pyqueue
    from lib import beamformer as B
    c = util.configure("shipped")
    M, N, X, Y = c["M"], c["N"], c["X"], c["Y"]
    sig = util.inputs("shipped")["s3"]
    B.connect(replay_mode=True, verbose=False)
    B.publish(sig)
    seen = np.empty((M, N), dtype=np.float32)
    B.receive(seen)
    q = pyqueue.Queue()
    B.b(q, True, max_frames=2)
    (img, nr), (img2, nr2) = q.get(), q.get()
    assert (nr, nr2) == (1, 2) and img.shape == (X, Y) and np.array_equal(img, img2)
    whole = util.table_for("pad", "shipped")
    orc = oracle_lib.Oracle(N, X, Y, c["T"])
    want = orc.mimo_pad(seen, whole, np.arange(M, dtype=np.int32))
    assert img.tobytes() == want.tobytes()
    # steering: (azimuth, elevation) degrees -> flat offset, main.pyx:498-515
    off = B.steer_cartesian_degree(0, 0)
    assert off == int((Y // 2) * X * M + int(0.5 * X) * M)
    nat.lib.load_coefficients_pad(nat.iptr(whole.ravel()), whole.size); nat.check()
    out = B.listen()
    assert out.tobytes() == orc.miso_pad(seen, whole, np.arange(M, dtype=np.int32), off).tobytes()
    # the playback loop's body (api.c:505-531) on the state that steer() / load_pa() keep: miso_pad, then / n * MIC_GAIN
    sub = np.array([3, 43, 44, 45, 46, 65, 66, 67], dtype=np.int32)       # microphones the dead-row mask leaves alive
    assert nat.lib.load_miso() == 0
    nat.lib.load_pa(nat.iptr(sub), sub.size)
    B.steer(off)
    blk = np.zeros(N, dtype=np.float32)
    assert nat.lib.bf_miso_listen_block(nat.fptr(blk), C.c_float(128.0)) == 0; nat.check()
    raw = orc.miso_pad(seen, whole, sub, off)
    want_blk = raw / np.float32(sub.size) * np.float32(128.0)
    assert blk.tobytes() == want_blk.astype(np.float32).tobytes() and np.abs(blk).max() > 0
    nat.lib.stop_miso()
    B.disconnect()


def test_beamformer_producer_loops(nat, oracle_lib):
    """The remaining lib.beamformer entry points (main.pyx:554-572, 810-820): map producers with steering requests,
    the MISO block producer, the audio hook that stands in for PortAudio."""
    import queue as pyqueue
    from lib import beamformer as B
    c = util.configure("cfg1")
    M, N, X, Y = c["M"], c["N"], c["X"], c["Y"]
    sig = util.inputs("cfg1")["s3"]
    mics = np.arange(M, dtype=np.int32)
    B.connect(replay_mode=True, verbose=False)
    B.publish(sig)
    sig = np.empty((M, N), dtype=np.float32)
    B.receive(sig)                      # what the loops see: the published window with the dead-microphone rows zeroed (api.c:835-858)
    whole = util.table_for("pad", "cfg1")
    orc = oracle_lib.Oracle(N, X, Y, c["T"])
    want_map = orc.mimo_pad(sig, whole, mics)
    blocks = []
    B.audio_sink = lambda blk: blocks.append(blk.copy())
    try:
        q_steer, q_out = pyqueue.Queue(), pyqueue.Queue()
        q_steer.put((0.25, 0.75))
        B.multi_pad(q_steer, q_out, True, max_frames=2)
        (img, _), (img2, _) = q_out.get(), q_out.get()
        assert img.tobytes() == want_map.tobytes() and img2.tobytes() == want_map.tobytes()
        off = int(int(0.75 * Y) * X * M + int(0.25 * X) * M)           # main.pyx:517-528
        assert B._steer_offset == off and len(blocks) == 2
        nat.lib.load_coefficients_pad(nat.iptr(whole.ravel()), whole.size); nat.check()
        assert blocks[1].tobytes() == orc.miso_pad(sig, whole, mics, off).tobytes()
        q = pyqueue.Queue()
        B.uti_api_with_miso(q, True, max_frames=1)
        assert q.get()[0].tobytes() == want_map.tobytes() and B._steer_offset == B.steer_cartesian_degree(0, 0)
        q = pyqueue.Queue()
        B.miso_api(q, True, max_frames=1)
        assert q.get().tobytes() == orc.miso_pad(sig, whole, mics, B._steer_offset).tobytes()
        q = pyqueue.Queue()
        q.put((0.5, 0.5))
        B.pure_miso_pad(q, True, max_requests=1)
        assert B._steer_offset == int(int(0.5 * Y) * X * M + int(0.5 * X) * M)
        B.just_miso_api(pyqueue.Queue(), True, max_frames=1)
    finally:
        B.audio_sink = None
        B.disconnect()


@pytest.mark.parametrize("algo", ["pad", "lerp"])
def test_batched_random_tables_every_step_reloads(nat, oracle_lib, algo):
    """A table with no structure (independent random delays per direction and mic): the sweep kernel's worst case, every
    direction step re-reads its quads.  Batched device path, 5 frames, against the oracle frame by frame."""
    torch = _torch()
    from interface import config
    M, N, X, Y, T, F = 64, 256, 24, 23, 8, 5
    config.configure(N_MICROPHONES=M, N_SAMPLES=N, MAX_RES_X=X, MAX_RES_Y=Y, N_TAPS=T)
    rng = np.random.default_rng(77)
    delays = rng.uniform(0, 40.0, size=(X, Y, M))
    table = delays.astype(int).astype(np.int32) if algo == "pad" else np.float32(delays)
    frames = (rng.standard_normal((F, M, N)) * 0.25).astype(np.float32)
    mics = np.arange(M, dtype=np.int32)
    orc = oracle_lib.Oracle(N, X, Y, T)
    orc.load(ALGOS[algo], table)
    run_product(nat, algo, table, frames[0], mics)          # loads the table into the library
    D = X * Y
    d_sig = torch.from_numpy(frames).cuda()
    d_img = torch.full((F, D), float("nan"), dtype=torch.float32, device="cuda")
    assert nat.lib.bf_das_device(ALGOS[algo], d_sig.data_ptr(), M, d_img.data_ptr(), D, F, nat.iptr(mics), M, 0, D,
                                 torch.cuda.current_stream().cuda_stream) == 0, nat.check()
    torch.cuda.synchronize()
    assert nat.lib.bf_last_das_variant() == 3          # the digest build found no structure: direction-outer variant
    got = d_img.cpu().numpy()
    for f in range(F):
        want = orc.mimo_range(ALGOS[algo], frames[f], mics, 0, D)
        assert max_rel(got[f], want) <= REL_TOL, f
