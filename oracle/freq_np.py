"""NumPy (float64 / complex128) restatement of the frequency-domain beamformers.  TEST INFRASTRUCTURE ONLY.

Phase-steer delay-and-sum follows PC/application/realtime_scripts: calc_r_prime.py:9-24 (mic positions incl. the 0.11 m
camera offset), calc_phase_shift_cartesian.py:17-50 (scan window, bins, phase shifts), beam_forming_algorithm.py:30-70
(rfft, steer, |sum|^2, sum over bins, normalise / threshold).  PINNED by tests/golden/fft_backend.npz, which
oracle/gen_golden_fft.py produced by importing the reference itself.

MVDR has no counterpart in the reference: `mvdr_power` below is the builder's definition and the
only oracle for it -- PARITY UNPINNED."""
import numpy as np

CAMERA_OFFSET = 0.11   # calc_r_prime.py:7


def r_prime_all(d=0.02, rows=8, columns=8, arrays=4, separation=0.0):
    """calc_r_prime.py:9-24 -> float64 [2, rows*columns*arrays]."""
    half = d / 2
    pos = np.zeros((2, rows * columns * arrays))
    e = 0
    for a in range(arrays):
        a = -a
        for row in range(rows):
            for col in range(columns):
                pos[0, e] = -col * d - half + a * columns * d + a * separation + columns * arrays * half
                pos[1, e] = row * d - rows * half + half - CAMERA_OFFSET
                e += 1
    pos[0, :] += (arrays - 1) * separation / 2
    return pos


def tables(n_samples=256, res_x=13, res_y=13, fs=48828, c=343.0, d=0.02, view_angle=68.0, z=1.0, aspect=16 / 9, arrays=4,
           f_lo=0, f_hi=18000, active=None):
    """calc_phase_shift_cartesian.py:17-50 -> dict(freq [K], bin_lo, bin_hi, g [X, Y, M] metres (path difference),
    phase_shift complex128 [K, M, X, Y])."""
    rp = r_prime_all(d, arrays=arrays)
    if active is None:
        active = np.arange(rp.shape[1])
    x_i = rp[0, :].reshape(1, -1, 1, 1)
    y_i = rp[1, :].reshape(1, -1, 1, 1)
    x_max = z * np.tan(np.deg2rad(view_angle / 2))
    y_max = x_max / aspect
    xs = np.linspace(-x_max, x_max, res_x).reshape(1, 1, res_x, 1)
    ys = np.linspace(-y_max, y_max, res_y).reshape(1, 1, 1, res_y)
    r = np.sqrt(xs ** 2 + ys ** 2 + z ** 2)
    f = np.linspace(0, int(int(fs) / 2), int(n_samples / 2) + 1).reshape(-1, 1, 1, 1)
    lo = int((np.abs(f - f_lo)).argmin())
    hi = int((np.abs(f - f_hi)).argmin())
    f = f[lo:hi]
    k = 2 * np.pi * f / c
    g = (xs * x_i + ys * y_i) / r                        # [1, M, X, Y]
    phase = np.exp(1j * (-k * g))
    return dict(freq=f.ravel(), bin_lo=lo, bin_hi=hi, g=g[0][active], phase_shift=phase[:, active], x_scan=xs.ravel(), y_scan=ys.ravel(),
                r_prime_all=rp)


def das_power(signal, phase_shift, bin_lo, bin_hi):
    """beam_forming_algorithm.py:30-36,52-57: signal float [N, M] -> float64 [X, Y] (before normalisation)."""
    X = np.fft.rfft(signal, axis=0)[bin_lo:bin_hi, :]
    X = X.reshape(X.shape[0], X.shape[1], 1, 1)
    return np.sum(np.abs(np.sum(X * phase_shift, axis=1)) ** 2, axis=0)


def das_heatmap(signal, phase_shift, bin_lo, bin_hi, threshold=0.2):
    """beam_forming_algorithm.py:50-70 (`main`)."""
    h = das_power(signal, phase_shift, bin_lo, bin_hi)
    if np.max(h) < threshold:
        h[:, :] = 0
    else:
        h = h / np.max(h)
    return h


def mvdr_power(frames, phase_shift, bin_lo, bin_hi, loading=1e-2):
    """Builder-defined MVDR (Capon) map.  frames float [F, N, M] (F windows of the same scene).
    Per bin k: R = (1/F) sum_f x x^H + loading * tr(R)/M * I;  P[x, y] = sum_k 1 / real(v^H R^-1 v),  v = conj(phase_shift[k, :, x, y]).
    (The reference's delay-and-sum output is sum_m phase_shift_m x_m = v^H x, so v is the steering vector in the usual w^H x sense.)"""
    F, N, M = frames.shape
    Xf = np.fft.rfft(frames, axis=1)[:, bin_lo:bin_hi, :]          # [F, K, M]
    K = Xf.shape[1]
    out = np.zeros(phase_shift.shape[2:])
    for k in range(K):
        x = Xf[:, k, :]                                            # [F, M]
        R = (x.T @ x.conj()) / F
        R = R + loading * (np.trace(R).real / M) * np.eye(M)
        a = np.conj(phase_shift[k].reshape(M, -1))                 # [M, D]
        Ria = np.linalg.solve(R, a)
        out += (1.0 / np.real(np.sum(a.conj() * Ria, axis=0))).reshape(out.shape)
    return out


# ---- the same two maps on a SHARD of the flat direction grid, without the [K, M, X, Y] phase table (at BASELINE config 5's
# size -- 256 mics x 361 x 361 directions x 377 bins -- that table is 200 GB in complex128; one rank's 1/8 shard is processed
# bin by bin instead).  Same arithmetic as tables() / das_power() / mvdr_power(): equal to them on the shard.

def tables_dirs(dir_lo, dir_hi, n_samples=256, res_x=13, res_y=13, fs=48828, c=343.0, d=0.02, view_angle=68.0, z=1.0, aspect=16 / 9,
                arrays=4, f_lo=0, f_hi=18000):
    """-> dict(freq [K], bin_lo, bin_hi, g float64 [M, hi - lo] metres, c): calc_phase_shift_cartesian.py:17-45 for the flat
    directions d = x * res_y + y in [dir_lo, dir_hi)."""
    rp = r_prime_all(d, arrays=arrays)
    x_max = z * np.tan(np.deg2rad(view_angle / 2))
    y_max = x_max / aspect
    xs_all = np.linspace(-x_max, x_max, res_x)
    ys_all = np.linspace(-y_max, y_max, res_y)
    flat = np.arange(dir_lo, dir_hi)
    xs, ys = xs_all[flat // res_y], ys_all[flat % res_y]
    r = np.sqrt(xs ** 2 + ys ** 2 + z ** 2)
    f = np.linspace(0, int(int(fs) / 2), int(n_samples / 2) + 1)
    lo = int((np.abs(f - f_lo)).argmin())
    hi = int((np.abs(f - f_hi)).argmin())
    g = (xs[None, :] * rp[0][:, None] + ys[None, :] * rp[1][:, None]) / r[None, :]
    return dict(freq=f[lo:hi].copy(), bin_lo=lo, bin_hi=hi, g=g, c=c, x_scan=xs_all, y_scan=ys_all, r_prime_all=rp)


def das_power_dirs(signal, t):
    """das_power() on the shard described by tables_dirs(): signal float [N, M] -> float64 [hi - lo]."""
    X = np.fft.rfft(signal, axis=0)[t["bin_lo"]:t["bin_hi"], :]
    out = np.zeros(t["g"].shape[1])
    for k, f in enumerate(t["freq"]):
        phase = np.exp(1j * (-(2 * np.pi * f / t["c"]) * t["g"]))    # [M, D]
        out += np.abs(X[k] @ phase) ** 2
    return out


def mvdr_power_dirs(frames, t, loading=1e-2):
    """mvdr_power() on the shard described by tables_dirs(): frames float [F, N, M] -> float64 [hi - lo]."""
    F, N, M = frames.shape
    Xf = np.fft.rfft(frames, axis=1)[:, t["bin_lo"]:t["bin_hi"], :]
    out = np.zeros(t["g"].shape[1])
    for k, f in enumerate(t["freq"]):
        x = Xf[:, k, :]
        R = (x.T @ x.conj()) / F
        R = R + loading * (np.trace(R).real / M) * np.eye(M)
        a = np.exp(1j * ((2 * np.pi * f / t["c"]) * t["g"]))         # conj(phase_shift[k]): [M, D]
        Ria = np.linalg.solve(R, a)
        out += 1.0 / np.real(np.sum(a.conj() * Ria, axis=0))
    return out
