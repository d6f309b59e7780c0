"""Frequency-domain beamformers on the MI355X, behind the reference's module / function names.

`main(signal)` is PC/application/realtime_scripts/beam_forming_algorithm.py:50-70: one window [N_SAMPLES, n_mics] in,
normalised heat-map [MAX_RES_X, MAX_RES_Y] out.  `FrequencyBeamformer` is the batched, device-resident form (many
windows per launch) and adds MVDR, which the reference does not have (BASELINE.json config 3; defined in DESIGN.md).

Host side (this file, NumPy): geometry, scan window, bin selection and the path-difference table tau[d][m] -- the
reference computes the same tables in NumPy at import (calc_r_prime.py:9-24, calc_phase_shift_cartesian.py:17-50).
Device side (csrc/freq_kernels.hip through the C-ABI): steering phasors, DFT of the selected bins, and the f32-MFMA
complex GEMMs with fused power / covariance / MVDR epilogues.  No NumPy fallback for the arithmetic."""
import numpy as np

from lib import _native as nat
from . import config

threshold_heatmap = 0.2     # beam_forming_algorithm.py:22


def calc_r_prime(d):
    """calc_r_prime.py:9-24 -> (r_prime_all float64 [2, N_MICROPHONES], r_prime of the active mics)."""
    half = d / 2
    pos = np.zeros((2, config.N_MICROPHONES))
    e = 0
    for a in range(config.ACTIVE_ARRAYS):
        a = -a
        for row in range(config.rows):
            for col in range(config.columns):
                pos[0, e] = -col * d - half + a * config.columns * d + a * config.ARRAY_SEPARATION + config.columns * config.ACTIVE_ARRAYS * half
                pos[1, e] = row * d - config.rows * half + half - config.CAMERA_OFFSET
                e += 1
    pos[0, :] += (config.ACTIVE_ARRAYS - 1) * config.ARRAY_SEPARATION / 2
    act = active_microphones()
    return pos, pos[:, act]


def active_microphones(unused=()):
    """active_microphones.py:4-46 (mode = every n-th mic; `unused` = contents of unused_mics.npy)."""
    per = config.rows * config.columns
    grid = np.hstack([np.arange(a * per, (a + 1) * per).reshape(config.rows, config.columns) for a in range(config.ACTIVE_ARRAYS)])
    out = [int(grid[r, c]) for r in range(0, config.rows, config.mode) for c in range(0, config.columns * config.ACTIVE_ARRAYS, config.mode)
           if int(grid[r, c]) not in set(unused)]
    return np.sort(np.asarray(out, dtype=np.int64))


def scan_tables(active=None):
    """calc_phase_shift_cartesian.py:17-45 -> (freq [K] Hz, bin_lo, bin_hi, tau float64 [X*Y, M] seconds, x_scan, y_scan).
    The reference's phase is -k * g with k = 2 pi f / c and g the path difference in metres; tau = g / c."""
    rp, _ = calc_r_prime(config.ELEMENT_DISTANCE)
    if active is None:
        active = active_microphones()
    x_i, y_i = rp[0, active], rp[1, active]
    x_max = config.Z * np.tan(np.deg2rad(config.VIEW_ANGLE / 2))
    y_max = x_max / config.ASPECT_RATIO
    xs = np.linspace(-x_max, x_max, config.MAX_RES_X).reshape(-1, 1, 1)
    ys = np.linspace(-y_max, y_max, config.MAX_RES_Y).reshape(1, -1, 1)
    r = np.sqrt(xs ** 2 + ys ** 2 + config.Z ** 2)
    f = np.linspace(0, int(int(config.fs) / 2), int(config.N_SAMPLES / 2) + 1)
    lo = int(np.abs(f - config.threshold_freq_lower).argmin())
    hi = int(np.abs(f - config.threshold_freq_upper).argmin())
    g = (xs * x_i + ys * y_i) / r                                   # [X, Y, M]
    tau = (g / config.PROPAGATION_SPEED).reshape(-1, len(active))
    return f[lo:hi].copy(), lo, hi, np.ascontiguousarray(tau), xs.ravel(), ys.ravel()


class FrequencyBeamformer:
    """Steering phasors resident in HBM; batched phase-steer DAS and MVDR over windows that are already on the device."""

    def __init__(self, active=None, device="cuda", dir_range=None):
        """`dir_range=(lo, hi)`: keep the steering phasors of the flat directions [lo, hi) only -- one rank's shard when the
        grid is split over GPUs (multi_gpu.sharded_heatmaps); the maps returned then have hi - lo columns."""
        import torch
        if not torch.cuda.is_available():
            raise nat.BeamformerError("no usable HIP device; the frequency-domain beamformers have no CPU fallback")
        self.torch = torch
        self.device = device
        self.active = np.ascontiguousarray(active_microphones() if active is None else active, dtype=np.int32)
        self.freq, self.bin_lo, self.bin_hi, tau, self.x_scan, self.y_scan = scan_tables(self.active)
        self.K, self.M, self.D = len(self.freq), len(self.active), config.MAX_RES_X * config.MAX_RES_Y
        self.dir_lo, self.dir_hi = (0, self.D) if dir_range is None else (int(dir_range[0]), int(dir_range[1]))
        if not 0 <= self.dir_lo < self.dir_hi <= self.D:
            raise ValueError("dir_range %r outside the %d-direction grid" % (dir_range, self.D))
        tau = np.ascontiguousarray(tau[self.dir_lo:self.dir_hi])
        self.D_full, self.D = self.D, self.dir_hi - self.dir_lo
        nat.lib.bf_configure(config.N_MICROPHONES, config.N_SAMPLES, config.MAX_RES_X, config.MAX_RES_Y, 8)
        nat.check()
        d_tau = torch.from_numpy(tau).to(device)
        d_freq = torch.from_numpy(np.ascontiguousarray(self.freq, dtype=np.float64)).to(device)
        self.a_re = torch.empty((self.K, self.M, self.D), dtype=torch.float32, device=device)
        self.a_im = torch.empty_like(self.a_re)
        self._call(nat.lib.bf_fd_steering_device, d_tau.data_ptr(), d_freq.data_ptr(), self.D, self.M, self.K, self.a_re.data_ptr(), self.a_im.data_ptr())
        torch.cuda.synchronize()

    def _call(self, fn, *args):
        if fn(*args, self.torch.cuda.current_stream().cuda_stream) != 0:
            nat.check()

    def spectra(self, d_frames):
        """float32 cuda tensor [F, rows, N_SAMPLES] (mic-major windows) -> the two operand layouts of X."""
        t = self.torch
        F, rows, N = d_frames.shape
        assert N == config.N_SAMPLES and d_frames.is_contiguous()
        mk = lambda *s: t.empty(s, dtype=t.float32, device=self.device)
        x = dict(re_mf=mk(self.K, self.M, F), im_mf=mk(self.K, self.M, F), re_fm=mk(self.K, F, self.M), im_fm=mk(self.K, F, self.M), F=F)
        self._call(nat.lib.bf_fd_dft_device, d_frames.data_ptr(), rows, F, nat.iptr(self.active), self.M, self.bin_lo, self.K,
                   x["re_mf"].data_ptr(), x["im_mf"].data_ptr(), x["re_fm"].data_ptr(), x["im_fm"].data_ptr())
        return x

    def das_power(self, d_frames):
        """beam_forming_algorithm.py:30-36,52-57 for every window: float32 [F, MAX_RES_X*MAX_RES_Y] (not normalised)."""
        x = self.spectra(d_frames)
        p = self.torch.empty((x["F"], self.D), dtype=self.torch.float32, device=self.device)
        self._call(nat.lib.bf_fd_das_power_device, x["re_mf"].data_ptr(), x["im_mf"].data_ptr(), self.a_re.data_ptr(), self.a_im.data_ptr(),
                   x["F"], self.M, self.D, self.K, p.data_ptr())
        return p

    def das_heatmap(self, d_frames):
        """beam_forming_algorithm.py:58-63: zero when the maximum is under the threshold, else divided by the maximum.
        (With a `dir_range` the maximum is the shard's: normalise after assembling the shards instead.)"""
        p = self.das_power(d_frames)
        mx = p.amax(dim=1, keepdim=True)
        return self.torch.where(mx < threshold_heatmap, self.torch.zeros_like(p), p / mx)

    def mvdr_power(self, d_frames, loading=1e-2, defer_check=False):
        """One MVDR (Capon) map from F windows of the same scene: float32 [MAX_RES_X*MAX_RES_Y].  Not in the reference.
        The factorisation reports bins whose covariance is not positive definite; reading that report back is a host synchronisation per map.
        `defer_check=True` folds it into a device-side running maximum instead (one tiny kernel) and returns at once: a stream of maps then
        pipelines, and `check_deferred()` -- to be called before the maps are trusted -- raises what the immediate check would have raised."""
        t = self.torch
        x = self.spectra(d_frames)
        mk = lambda *s: t.empty(s, dtype=t.float32, device=self.device)
        r_re, r_im, l_re, l_im = mk(self.K, self.M, self.M), mk(self.K, self.M, self.M), mk(self.K, self.M, self.M), mk(self.K, self.M, self.M)
        status = t.zeros((self.K,), dtype=t.int32, device=self.device)
        p = mk(self.D)
        self._call(nat.lib.bf_fd_covariance_device, x["re_fm"].data_ptr(), x["im_fm"].data_ptr(), x["F"], self.M, self.K, r_re.data_ptr(), r_im.data_ptr())
        self._call(nat.lib.bf_fd_cholesky_inverse_device, r_re.data_ptr(), r_im.data_ptr(), self.M, self.K, float(loading), l_re.data_ptr(),
                   l_im.data_ptr(), status.data_ptr())
        self._call(nat.lib.bf_fd_mvdr_power_device, l_re.data_ptr(), l_im.data_ptr(), self.a_re.data_ptr(), self.a_im.data_ptr(), self.M, self.D, self.K,
                   p.data_ptr())
        if defer_check:
            worst = getattr(self, "_deferred_status", None)
            self._deferred_status = status if worst is None else t.maximum(worst, status)
            return p
        if int(status.max().item()) != 0:
            raise nat.BeamformerError("MVDR covariance is not positive definite in %d bin(s); raise `loading`" % int((status != 0).sum().item()))
        return p

    def check_deferred(self):
        """The positive-definiteness report of every `mvdr_power(..., defer_check=True)` call since the last check (synchronises)."""
        worst, self._deferred_status = getattr(self, "_deferred_status", None), None
        if worst is not None and int(worst.max().item()) != 0:
            raise nat.BeamformerError("MVDR covariance was not positive definite in %d bin(s) of at least one map; raise `loading`" % int((worst != 0).sum().item()))


_default = None


def main(signal):
    """beam_forming_algorithm.py:50-70: one window float [N_SAMPLES, n_active_mics] -> heat-map float [MAX_RES_X, MAX_RES_Y]."""
    global _default
    import torch
    if _default is None:
        _default = FrequencyBeamformer()
    fb = _default
    sig = np.ascontiguousarray(np.asarray(signal, dtype=np.float32).T)          # -> mic-major [M, N]
    assert sig.shape == (fb.M, config.N_SAMPLES), "signal must be [N_SAMPLES, n_active_mics]"
    frame = np.zeros((1, config.N_MICROPHONES, config.N_SAMPLES), dtype=np.float32)
    frame[0, fb.active] = sig
    h = fb.das_heatmap(torch.from_numpy(frame).to(fb.device))
    return h[0].double().cpu().numpy().reshape(config.MAX_RES_X, config.MAX_RES_Y)
