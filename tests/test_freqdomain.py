"""Frequency-domain beamformers.

CPU: the NumPy oracle against the golden vectors produced by importing the reference's realtime_scripts (bit-exact).
GPU: the MFMA path against the golden vectors (phase-steer DAS) and against the builder-defined float64 oracle (MVDR).
The reference computes in float64/complex128; the GPU path is float32 (exact-f32 MFMA), so the tolerance is relative to
the map's peak: 2e-5 of the maximum per pixel (the maps are normalised by their maximum before display)."""
import hashlib

import numpy as np
import pytest

import util

TOL_OF_PEAK = 2e-5


def gold():
    return np.load(util.GOLDEN + "/fft_backend.npz")


def test_oracle_matches_reference_bit_exact():
    import freq_np as F
    g = gold()
    t = F.tables()
    assert np.array_equal(t["r_prime_all"], g["r_prime_all"]) and np.array_equal(t["freq"], g["freq"])
    assert (t["bin_lo"], t["bin_hi"]) == (int(g["bin_lo"]), int(g["bin_hi"]))
    assert hashlib.sha256(np.ascontiguousarray(t["phase_shift"]).tobytes()).hexdigest() == str(g["phase_shift_sha256"])
    for k in ("f1", "f2", "f3"):
        assert np.array_equal(F.das_power(g["in_" + k], t["phase_shift"], t["bin_lo"], t["bin_hi"]), g["power_" + k])
        assert np.array_equal(F.das_heatmap(g["in_" + k], t["phase_shift"], t["bin_lo"], t["bin_hi"]), g["heatmap_" + k])
    assert not g["heatmap_f3"].any()                      # quiet input: the reference blanks the map


def _scene(t, ix, iy, n_frames, rng, M=64, N=256):
    xs, ys = t["x_scan"][ix], t["y_scan"][iy]
    r = np.sqrt(xs * xs + ys * ys + 1.0)
    tau = (xs * t["r_prime_all"][0] + ys * t["r_prime_all"][1]) / r / 343.0
    tt = np.arange(N) / 48828.0
    return np.stack([sum(np.sin(2 * np.pi * f * (tt[:, None] + tau[None, :]) + rng.uniform(0, 6.28)) for f in (2500.0, 5200.0, 8100.0)) +
                     0.3 * rng.standard_normal((N, M)) for _ in range(n_frames)])


def test_shard_oracle_equals_full_grid_oracle():
    """tables_dirs / das_power_dirs / mvdr_power_dirs (bin by bin, no [K, M, X, Y] table) == the full-grid oracle on the shard."""
    import freq_np as F
    t = F.tables(res_x=9, res_y=7, arrays=1)
    td = F.tables_dirs(10, 40, res_x=9, res_y=7, arrays=1)
    frames = np.random.default_rng(1).standard_normal((70, 256, 64))
    full = F.mvdr_power(frames, t["phase_shift"], t["bin_lo"], t["bin_hi"]).ravel()
    assert np.max(np.abs(F.mvdr_power_dirs(frames, td) - full[10:40]) / full[10:40]) <= 1e-12
    d = F.das_power(frames[0], t["phase_shift"], t["bin_lo"], t["bin_hi"]).ravel()
    assert np.max(np.abs(F.das_power_dirs(frames[0], td) - d[10:40])) <= 1e-12 * d.max()


def test_mvdr_oracle_properties():
    """Builder-defined MVDR: peak at the source, main lobe no wider than delay-and-sum on the same data."""
    import freq_np as F
    t = F.tables(res_x=9, res_y=7, arrays=1)
    frames = _scene(t, 6, 2, 96, np.random.default_rng(3))
    p = F.mvdr_power(frames, t["phase_shift"], t["bin_lo"], t["bin_hi"])
    assert np.unravel_index(np.argmax(p), p.shape) == (6, 2)
    d = sum(F.das_power(fr, t["phase_shift"], t["bin_lo"], t["bin_hi"]) for fr in frames)
    assert (p / p.max() > 0.5).sum() <= (d / d.max() > 0.5).sum()


@pytest.mark.gpu
def test_gpu_phase_steer_matches_reference(native):
    import torch
    from realtime_scripts import beam_forming_algorithm as B
    g = gold()
    for k in ("f1", "f2", "f3"):
        got = B.main(g["in_" + k])
        want = g["heatmap_" + k]
        assert got.shape == want.shape
        assert np.max(np.abs(got - want)) <= TOL_OF_PEAK * max(1.0, float(want.max())), k
    fb = B._default
    frames = np.zeros((3, 256, 256), dtype=np.float32)
    for i, k in enumerate(("f1", "f2", "f3")):
        frames[i, fb.active] = g["in_" + k].T
    p = fb.das_power(torch.from_numpy(frames).cuda()).double().cpu().numpy()
    for i, k in enumerate(("f1", "f2", "f3")):
        want = g["power_" + k].ravel()
        assert np.max(np.abs(p[i] - want)) <= TOL_OF_PEAK * want.max(), k
    assert np.argmax(p[1]) == np.argmax(g["power_f2"])


@pytest.mark.gpu
def test_gpu_mvdr_matches_float64_oracle(native):
    import torch
    import freq_np as F
    from realtime_scripts import beam_forming_algorithm as B, config as C
    old = (C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y)
    C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = 64, 1, 31, 23
    try:
        fb = B.FrequencyBeamformer()
        t = F.tables(res_x=31, res_y=23, arrays=1)
        frames = _scene(t, 20, 7, 128, np.random.default_rng(5)).astype(np.float32)
        want = F.mvdr_power(frames.astype(np.float64), t["phase_shift"], t["bin_lo"], t["bin_hi"], loading=1e-2).ravel()
        d_frames = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 2, 1))).cuda()      # [F, M, N] mic-major
        got = fb.mvdr_power(d_frames, loading=1e-2).double().cpu().numpy()
        assert np.argmax(got) == np.argmax(want) == 20 * 23 + 7
        assert np.max(np.abs(got - want) / want) <= 2e-4          # conditioning of R^-1 in float32 (delta = 1e-2)
        dwant = np.stack([F.das_power(fr, t["phase_shift"], t["bin_lo"], t["bin_hi"]).ravel() for fr in frames[:8].astype(np.float64)])
        dgot = fb.das_power(d_frames[:8].contiguous()).double().cpu().numpy()
        assert np.max(np.abs(dgot - dwant)) <= TOL_OF_PEAK * dwant.max()
    finally:
        C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = old


def _with_config(**kw):
    from realtime_scripts import config as C
    old = {k: getattr(C, k) for k in kw}
    for k, v in kw.items():
        setattr(C, k, v)
    return C, old


@pytest.mark.gpu
def test_gpu_config3_full_size_mvdr_and_das(native):
    """BASELINE config 3 at its stated size: 64 mics, 256-sample windows, 101 x 101 directions -- MVDR (builder-defined) and
    phase-steer delay-and-sum over the whole grid against the float64 oracle."""
    import torch
    import freq_np as F
    from realtime_scripts import beam_forming_algorithm as B
    C, old = _with_config(N_MICROPHONES=64, ACTIVE_ARRAYS=1, MAX_RES_X=101, MAX_RES_Y=101)
    try:
        fb = B.FrequencyBeamformer()
        assert (fb.M, fb.D, fb.K) == (64, 101 * 101, 94)
        t = F.tables_dirs(0, 101 * 101, res_x=101, res_y=101, arrays=1)
        ix, iy = 70, 33
        frames = _scene(t, ix, iy, 190, np.random.default_rng(31)).astype(np.float32)
        want = F.mvdr_power_dirs(frames.astype(np.float64), t, loading=1e-2)
        d_frames = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 2, 1))).cuda()      # [F, M, N] mic-major
        got = fb.mvdr_power(d_frames, loading=1e-2).double().cpu().numpy()
        assert np.argmax(got) == np.argmax(want) == ix * 101 + iy
        assert np.max(np.abs(got - want) / want) <= 2e-4
        dwant = np.stack([F.das_power_dirs(fr, t) for fr in frames[:4].astype(np.float64)])
        dgot = fb.das_power(d_frames[:4].contiguous()).double().cpu().numpy()
        assert np.max(np.abs(dgot - dwant)) <= TOL_OF_PEAK * dwant.max()
    finally:
        for k, v in old.items():
            setattr(C, k, v)


@pytest.mark.gpu
def test_gpu_config5_one_rank_shard_mvdr_and_das(native):
    """BASELINE config 5 at its stated size, as ONE of the 8 ranks sees it: 256 mics, 1024-sample windows (377 bins), the
    rank's shard of the 361 x 361 grid (16,291 directions; the full steering table would be ~100 GB).  The GPU computes the whole
    shard; the float64 oracle checks every 16th direction of it (the oracle's solves are what takes the time)."""
    import torch
    import freq_np as F
    import multi_gpu
    from realtime_scripts import beam_forming_algorithm as B
    C, old = _with_config(N_MICROPHONES=256, ACTIVE_ARRAYS=4, MAX_RES_X=361, MAX_RES_Y=361, N_SAMPLES=1024)
    try:
        D = 361 * 361
        lo, hi = multi_gpu.shard_range(D, 8, 3)
        assert hi - lo == 16291
        fb = B.FrequencyBeamformer(dir_range=(lo, hi))
        assert (fb.M, fb.D, fb.D_full, fb.K) == (256, 16291, D, 377)
        src = lo + 5000                                             # a source inside the shard, on a checked direction
        pick = np.arange(lo + (src - lo) % 16, hi, 16)
        t_src = F.tables_dirs(src, src + 1, n_samples=1024, res_x=361, res_y=361, arrays=4)
        rng = np.random.default_rng(41)
        tt = np.arange(1024) / 48828.0
        tau = t_src["g"][:, 0] / 343.0
        frames = np.stack([sum(np.sin(2 * np.pi * f * (tt[:, None] + tau[None, :]) + rng.uniform(0, 6.28)) for f in (2500.0, 5200.0, 8100.0)) +
                           0.3 * rng.standard_normal((1024, 256)) for _ in range(320)]).astype(np.float32)
        d_frames = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 2, 1))).cuda()      # [F, M, N] mic-major
        got = fb.mvdr_power(d_frames, loading=1e-2).double().cpu().numpy()
        dgot = fb.das_power(d_frames[:2].contiguous()).double().cpu().numpy()
        del fb
        # the oracle on the picked directions: g of those columns only
        t = F.tables_dirs(lo, hi, n_samples=1024, res_x=361, res_y=361, arrays=4)
        t["g"] = np.ascontiguousarray(t["g"][:, pick - lo])
        want = F.mvdr_power_dirs(frames.astype(np.float64), t, loading=1e-2)
        assert lo + int(np.argmax(got)) == src and pick[int(np.argmax(want))] == src
        assert np.max(np.abs(got[pick - lo] - want) / want) <= 5e-4     # conditioning of R^-1 in float32 at 256 mics
        dwant = np.stack([F.das_power_dirs(fr, t) for fr in frames[:2].astype(np.float64)])
        assert np.max(np.abs(dgot[:, pick - lo] - dwant)) <= TOL_OF_PEAK * dwant.max()
    finally:
        for k, v in old.items():
            setattr(C, k, v)
        util.configure("cfg1")


GEMM_SHAPES = [
    # rows(I) mics(K) dirs(J) bins   what it exercises
    (150,  37,  45,  9),    # 5 row tiles -> 2 row groups of 3, odd K (half-empty last MFMA step), ragged column tile
    (33,  100, 200,  5),    # 2 panels of K, second partial; 2 row tiles, second almost empty
    (190,  64, 333, 23),    # the bench shape in miniature: several bin groups + the plane reduction
    (8,   130,  64,  3),    # K > 128: three panels
]


@pytest.fixture(params=[0, 1], ids=["f32_mfma", "bf16_split"])
def gemm_mode(request, native):
    """bf_fd_gemm_f32_mode for the test: the float32 matrix instruction, and the exact three-way bfloat16 split (same bounds for both)."""
    initial = native.lib.bf_fd_gemm_f32_mode(request.param)
    yield request.param
    native.lib.bf_fd_gemm_f32_mode(initial)


@pytest.mark.gpu
@pytest.mark.parametrize("I,K,J,B", GEMM_SHAPES)
def test_gpu_bin_reducing_gemms_match_numpy(native, gemm_mode, I, K, J, B):
    """bf_fd_das_power_device / bf_fd_mvdr_power_device on random complex operands against complex128 NumPy:
    P[f, d] = sum_b |sum_k X[b,k,f] A[b,k,d]|^2   and   P[d] = sum_b 1 / sum_i |sum_k L[b,k,i] conj(A[b,k,d])|^2."""
    import torch
    rng = np.random.default_rng(I * 1000 + K)
    x = (rng.standard_normal((B, K, I)) + 1j * rng.standard_normal((B, K, I))).astype(np.complex64)
    a = np.exp(1j * rng.uniform(0, 2 * np.pi, (B, K, J))).astype(np.complex64)
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    xr, xi, ar, ai = dev(x.real), dev(x.imag), dev(a.real), dev(a.imag)
    p = torch.full((I, J), float("nan"), dtype=torch.float32, device="cuda")
    assert native.lib.bf_fd_das_power_device(xr.data_ptr(), xi.data_ptr(), ar.data_ptr(), ai.data_ptr(), I, K, J, B, p.data_ptr(), None) == 0, native.check()
    want = (np.abs(np.einsum("bki,bkj->bij", x.astype(np.complex128), a.astype(np.complex128))) ** 2).sum(0)
    got = p.double().cpu().numpy()
    assert np.max(np.abs(got - want)) <= TOL_OF_PEAK * want.max()


@pytest.mark.gpu
@pytest.mark.parametrize("M,J,B", [(37, 45, 9), (64, 333, 23), (100, 70, 4), (128, 200, 6)])
def test_gpu_mvdr_quadratic_form_matches_numpy(native, gemm_mode, M, J, B):
    """bf_fd_mvdr_power_device on a random (transposed) triangular-inverse stand-in L[b, k, i] = Linv[i][k], zero for k > i as the planes of
    bf_fd_cholesky_inverse_device are (the kernel does not multiply the all-zero 32 x 32 blocks):
    P[d] = sum_b 1 / sum_i |sum_k L[b,k,i] conj(A[b,k,d])|^2  against complex128 NumPy."""
    import torch
    rng = np.random.default_rng(M)
    l = np.triu(rng.standard_normal((B, M, M)) + 1j * rng.standard_normal((B, M, M))).astype(np.complex64)
    a = np.exp(1j * rng.uniform(0, 2 * np.pi, (B, M, J))).astype(np.complex64)
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v)).cuda()
    lr, li, ar, ai = dev(l.real), dev(l.imag), dev(a.real), dev(a.imag)
    q = torch.full((J,), float("nan"), dtype=torch.float32, device="cuda")
    assert native.lib.bf_fd_mvdr_power_device(lr.data_ptr(), li.data_ptr(), ar.data_ptr(), ai.data_ptr(), M, J, B, q.data_ptr(), None) == 0, native.check()
    y = np.einsum("bki,bkj->bij", l.astype(np.complex128), np.conj(a.astype(np.complex128)))
    want = (1.0 / (np.abs(y) ** 2).sum(1)).sum(0)
    assert np.max(np.abs(q.double().cpu().numpy() - want) / want) <= 1e-4


@pytest.mark.gpu
@pytest.mark.parametrize("M,B", [(192, 3), (256, 4), (129, 2)])
def test_gpu_blocked_cholesky_inverse_matches_numpy(native, M, B):
    """bf_fd_cholesky_inverse_device above 128 mics (two blocks of 128 + strided GEMMs): the transposed planes must be
    inverse(cholesky(R + loading tr(R)/M I)) of random Hermitian positive-definite matrices."""
    import torch
    rng = np.random.default_rng(M)
    x = rng.standard_normal((B, M, 2 * M)) + 1j * rng.standard_normal((B, M, 2 * M))
    r = (x @ x.conj().transpose(0, 2, 1)) / (2 * M)
    loading = 1e-2
    dev = lambda v: torch.from_numpy(np.ascontiguousarray(v.astype(np.float32))).cuda()
    rr, ri = dev(r.real), dev(r.imag)
    lr = torch.full((B, M, M), float("nan"), dtype=torch.float32, device="cuda")
    li = torch.full((B, M, M), float("nan"), dtype=torch.float32, device="cuda")
    st = torch.zeros((B,), dtype=torch.int32, device="cuda")
    assert native.lib.bf_fd_cholesky_inverse_device(rr.data_ptr(), ri.data_ptr(), M, B, loading, lr.data_ptr(), li.data_ptr(), st.data_ptr(), None) == 0, native.check()
    assert int(st.abs().max().item()) == 0
    got_t = lr.double().cpu().numpy() + 1j * li.double().cpu().numpy()          # [b][c][r] = Linv[r][c]
    r32 = rr.double().cpu().numpy() + 1j * ri.double().cpu().numpy()
    for b in range(B):
        rl = r32[b] + loading * np.trace(r32[b]).real / M * np.eye(M)
        want = np.linalg.inv(np.linalg.cholesky(rl))
        got = got_t[b].T
        assert np.max(np.abs(got - want)) <= 2e-4 * np.max(np.abs(want)), (M, b)
        assert np.max(np.abs(np.triu(got, 1))) == 0.0


@pytest.mark.gpu
def test_gpu_mvdr_256_mics_matches_float64_oracle(native):
    """BASELINE config 5's array (4 tiles = 256 mics) through the whole MVDR path against the float64 oracle."""
    import torch
    import freq_np as F
    from realtime_scripts import beam_forming_algorithm as B, config as C
    old = (C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y)
    C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = 256, 4, 9, 7
    try:
        fb = B.FrequencyBeamformer()
        assert fb.M == 256
        t = F.tables(res_x=9, res_y=7, arrays=4)
        frames = _scene(t, 6, 2, 320, np.random.default_rng(9), M=256).astype(np.float32)
        want = F.mvdr_power(frames.astype(np.float64), t["phase_shift"], t["bin_lo"], t["bin_hi"], loading=1e-2).ravel()
        d_frames = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 2, 1))).cuda()      # [F, M, N] mic-major
        got = fb.mvdr_power(d_frames, loading=1e-2).double().cpu().numpy()
        assert np.argmax(got) == np.argmax(want) == 6 * 7 + 2
        assert np.max(np.abs(got - want) / want) <= 5e-4          # conditioning of R^-1 in float32 at 256 mics
    finally:
        C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = old


@pytest.mark.gpu
@pytest.mark.parametrize("N,M,F,lo,nb", [(1024, 8, 3, 5, 300), (256, 5, 7, 0, 129), (512, 33, 2, 17, 97), (200, 4, 2, 3, 40)])
def test_gpu_dft_matches_numpy_rfft(native, N, M, F, lo, nb):
    """bf_fd_dft_device (twiddle GEMM on the matrix cores, bin groups beyond 128 bins) against numpy.fft.rfft, both layouts."""
    import torch
    from interface import config
    config.configure(N_MICROPHONES=M, N_SAMPLES=N, MAX_RES_X=3, MAX_RES_Y=3, N_TAPS=8)
    try:
        rng = np.random.default_rng(N + nb)
        sig = rng.standard_normal((F, M, N)).astype(np.float32)
        mics = np.arange(M, dtype=np.int32)
        d = torch.from_numpy(sig).cuda()
        mk = lambda *s: torch.full(s, float("nan"), dtype=torch.float32, device="cuda")
        re_mf, im_mf, re_fm, im_fm = mk(nb, M, F), mk(nb, M, F), mk(nb, F, M), mk(nb, F, M)
        assert native.lib.bf_fd_dft_device(d.data_ptr(), M, F, native.iptr(mics), M, lo, nb, re_mf.data_ptr(), im_mf.data_ptr(), re_fm.data_ptr(),
                                           im_fm.data_ptr(), None) == 0, native.check()
        want = np.fft.rfft(sig.astype(np.float64), axis=2)[:, :, lo:lo + nb]          # [F, M, nb]
        got_mf = re_mf.double().cpu().numpy() + 1j * im_mf.double().cpu().numpy()     # [nb, M, F]
        got_fm = re_fm.double().cpu().numpy() + 1j * im_fm.double().cpu().numpy()     # [nb, F, M]
        tol = 2e-6 * np.abs(want).max() * np.sqrt(N)
        assert np.max(np.abs(got_mf - want.transpose(2, 1, 0))) <= tol
        assert np.max(np.abs(got_fm - want.transpose(2, 0, 1))) <= tol
    finally:
        util.configure("cfg1")


@pytest.mark.gpu
def test_gpu_direction_shards_of_the_frequency_domain_maps(native):
    """FrequencyBeamformer(dir_range=...) holds one shard of the steering phasors; shards put side by side (what
    multi_gpu.sharded_heatmaps' all-gather does across ranks) equal the full-grid maps, DAS and MVDR."""
    import torch
    import freq_np as F
    import multi_gpu
    from realtime_scripts import beam_forming_algorithm as B, config as C
    old = (C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y)
    C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = 64, 1, 13, 11
    try:
        t = F.tables(res_x=13, res_y=11, arrays=1)
        frames = _scene(t, 4, 7, 96, np.random.default_rng(21)).astype(np.float32)
        d_frames = torch.from_numpy(np.ascontiguousarray(frames.transpose(0, 2, 1))).cuda()
        full = B.FrequencyBeamformer()
        D = full.D
        want_mvdr, want_das = full.mvdr_power(d_frames), full.das_power(d_frames[:5].contiguous())
        parts_m, parts_d = [], []
        for rank in range(3):                                   # the ranges three ranks would own
            lo, hi = multi_gpu.shard_range(D, 3, rank)
            fb = B.FrequencyBeamformer(dir_range=(lo, hi))
            assert fb.D == hi - lo and fb.D_full == D
            parts_m.append(fb.mvdr_power(d_frames))
            parts_d.append(fb.das_power(d_frames[:5].contiguous()))
        assert torch.allclose(torch.cat(parts_m), want_mvdr, rtol=1e-6, atol=0)
        assert torch.allclose(torch.cat(parts_d, dim=1), want_das, rtol=1e-6, atol=0)
        # world size 1: sharded_heatmaps degenerates to the compute function on the whole grid
        one = multi_gpu.sharded_heatmaps(lambda lo, hi: B.FrequencyBeamformer(dir_range=(lo, hi)).mvdr_power(d_frames)[None, :], 1, D, device="cuda")
        assert torch.allclose(one[0], want_mvdr, rtol=1e-6, atol=0)
    finally:
        C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = old


@pytest.mark.gpu
def test_gpu_mvdr_deferred_status_check(native):
    """mvdr_power(defer_check=True) returns the same map without the per-map read-back, and check_deferred() raises what the immediate check raises:
    a covariance made indefinite (negative loading on rank-deficient data) is reported by both."""
    import torch
    from realtime_scripts import beam_forming_algorithm as B, config as C
    old = (C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y)
    C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = 64, 1, 11, 11
    try:
        fb = B.FrequencyBeamformer()
        rng = np.random.default_rng(5)
        frames = torch.from_numpy(rng.standard_normal((96, 64, C.N_SAMPLES)).astype(np.float32)).cuda()
        now = fb.mvdr_power(frames, 1e-2)
        later = fb.mvdr_power(frames, 1e-2, defer_check=True)
        fb.check_deferred()
        assert torch.equal(now, later)
        few = frames[:8].contiguous()                       # 8 windows, 64 mics: rank 8, and a negative loading makes it indefinite
        with pytest.raises(Exception):
            fb.mvdr_power(few, -1e-3)
        fb.mvdr_power(few, -1e-3, defer_check=True)
        fb.mvdr_power(frames, 1e-2, defer_check=True)       # a good map after the bad one does not clear the report
        with pytest.raises(Exception):
            fb.check_deferred()
        fb.check_deferred()                                 # cleared
    finally:
        C.N_MICROPHONES, C.ACTIVE_ARRAYS, C.MAX_RES_X, C.MAX_RES_Y = old
