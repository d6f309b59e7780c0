"""Synthetic microphone blocks S1-S3 used by tests and bench.py.

All blocks are float32 [M, N], C-contiguous, mic-major: the layout the reference's receiver
produces (`signals[mic * N_SAMPLES + t]`, PC/src/receiver.c:94-151).
Pure NumPy; no dependence on the oracle or on the HIP extension.
"""
import numpy as np

FS = 48828.0  # PC/src/config.json:16


def s1_tone(M, N, frequency=8000.0, fs=FS):
    """8 kHz unit sine, identical in every mic: the reference's own smoke input (PC/plot.py:8-21)."""
    t = np.arange(0, 1, 1 / int(fs))[:N]
    row = np.sin(2 * np.pi * frequency * t + 0)
    return np.ascontiguousarray(np.repeat(row[None, :], M, axis=0).astype(np.float32))


def s2_noise(M, N, seed=0):
    """White gaussian block, sigma = 1/8."""
    rng = np.random.default_rng(seed)
    return np.ascontiguousarray((rng.standard_normal((M, N)).astype(np.float32) * np.float32(2.0 ** -3)))


def s3_plane_wave(delay_row, N, fs=FS, seed=1):
    """Plane wave arriving from the direction whose delay row (in samples, float64 [M]) is given.

    Mic m LEADS by delay_row[m] samples so that delay-and-sum with that row re-aligns it:
    sig_m[k] = s((k + delay_m) / fs), s = three tones, plus noise 20 dB below the tone power.
    """
    delay_row = np.asarray(delay_row, dtype=np.float64)
    M = delay_row.shape[0]
    k = np.arange(N, dtype=np.float64)[None, :]
    t = (k + delay_row[:, None]) / fs
    s = np.zeros_like(t)
    for f in (1500.0, 3000.0, 6000.0):
        s += np.sin(2 * np.pi * f * t)
    s /= 3.0
    rng = np.random.default_rng(seed)
    p_sig = float(np.mean(s ** 2))
    noise = rng.standard_normal((M, N)) * np.sqrt(p_sig * 10 ** (-20 / 10))
    return np.ascontiguousarray((s + noise).astype(np.float32))


def frame_batch(M, N, frames, seed0=100):
    """`frames` independent S2 blocks (seed0 + i): the batched-stream bench input, float32 [F, M, N]."""
    out = np.empty((frames, M, N), dtype=np.float32)
    for i in range(frames):
        out[i] = s2_noise(M, N, seed=seed0 + i)
    return out
