"""Mirror of the reference's `lib.directions` module (PC/src/directions.pyx): same function names, arguments and
return shapes/dtypes, computed by the C++ generators in csrc/directions.cpp through the C-ABI."""
import ctypes as C
import os

import numpy as np

from interface import config
from . import _native as nat


def _unused():
    """directions.pyx:58-63: `unused_mics.npy` in the working directory, indices offset by +64."""
    try:
        u = np.load("unused_mics.npy").astype(np.int64) + 64
    except Exception:
        return None, 0
    u = np.ascontiguousarray(u, dtype=np.int32)
    return u, int(u.size)


def active_microphones():
    """directions.pyx:35-87 -> (sorted int array of active mic indices, count)."""
    g = nat.geometry()
    u, nu = _unused()
    out = np.zeros(g.rows * g.columns * g.arrays, dtype=np.int32)
    n = nat.lib.bf_active_microphones(C.byref(g), nat.iptr(u) if nu else None, nu, nat.iptr(out))
    return out[:n].astype(np.int64), n


def calc_r_prime(d):
    """directions.pyx:17-32 -> float64 [2, n_active]; `d` is the element pitch in metres."""
    g = nat.geometry()
    g.element_distance = d          # the reference passes its float32-typed `d` here
    u, nu = _unused()
    out = np.zeros((2, g.rows * g.columns * g.arrays))
    n = nat.lib.bf_calc_r_prime(C.byref(g), nat.iptr(u) if nu else None, nu, nat.dptr(out))
    return np.ascontiguousarray(out.reshape(-1)[:2 * n].reshape(2, n))


def calculate_delays():
    """directions.pyx:90-124 -> float64 [MAX_RES_X, MAX_RES_Y, n_active], delays in samples."""
    g = nat.geometry()
    u, nu = _unused()
    n = nat.lib.bf_active_microphones(C.byref(g), nat.iptr(u) if nu else None, nu, None)
    out = np.zeros((config.MAX_RES_X, config.MAX_RES_Y, n))
    got = nat.lib.bf_calculate_delays(C.byref(g), config.MAX_RES_X, config.MAX_RES_Y, nat.iptr(u) if nu else None, nu, nat.dptr(out))
    if got != n:
        raise nat.BeamformerError("bf_calculate_delays returned %d" % got)
    return out


def get_h(delay, N=8):
    """directions.pyx:189-205 -> float64[8] sinc*Blackman taps of a fractional delay (N is ignored there too)."""
    out = np.zeros(8)
    nat.lib.bf_get_h(float(delay), nat.dptr(out))
    return out


def get_h2(delay, N=64):
    """directions.pyx:207-226 -> float32[N]."""
    out = np.zeros(int(N), dtype=np.float32)
    nat.lib.bf_get_h2(float(delay), int(N), nat.fptr(out))
    return out


def compute_convolve_h():
    """directions.pyx:229-247 -> float32 [X, Y, n_active, N_TAPS]: get_h2 of the full delay."""
    d = calculate_delays()
    T = config.N_TAPS
    h = np.zeros(d.shape + (T,), dtype=np.float32)
    nat.lib.bf_get_h2_batch(nat.dptr(d), d.size, T, nat.fptr(h))
    return h


def calculate_coefficients():
    """directions.pyx:260-277 -> (int whole-sample delays [X,Y,M], float32 get_h taps [X,Y,M,8])."""
    d = calculate_delays()
    whole = d.astype(int)
    frac = d - whole
    h = np.zeros(d.shape + (8,), dtype=np.float32)
    frac = np.ascontiguousarray(frac)
    nat.lib.bf_get_h_batch(nat.dptr(frac), frac.size, nat.fptr(h))
    return whole, h


def _tile_factors():
    """Element coordinates of ONE 8x8 tile about its centre, as directions.pyx:143-147 / :174-177 compute them
    (distance 0.02 hard-coded there): float64 [ROWS*COLUMNS] column and row offsets, row-major."""
    distance = 0.02
    half = distance / 2.0
    col = np.arange(config.COLUMNS) * distance - config.COLUMNS * half + half
    row = np.arange(config.ROWS) * distance - config.ROWS * half + half
    return np.tile(col, config.ROWS), np.repeat(row, config.COLUMNS)


def _samples_per_metre():
    # SAMPLE_RATE and PROPAGATION_SPEED are C floats in the reference (config.pxd:14-23): a float32 division
    return float(np.float32(config.SAMPLE_RATE) / np.float32(config.PROPAGATION_SPEED))


def calculate_delays_():
    """directions.pyx:126-157 (legacy, angle grid): float32 [MAX_RES_X, MAX_RES_Y, COLUMNS*ROWS*ACTIVE_TILES]; only the
    first tile's entries are filled there, the minimum over them (or 0) is subtracted from every entry."""
    tc, tr = _tile_factors()
    n = config.COLUMNS * config.ROWS
    out = np.zeros((config.MAX_RES_X, config.MAX_RES_Y, n * config.ACTIVE_TILES), dtype=np.float32)
    xf = np.sin(np.linspace(-config.MAX_ANGLE, config.MAX_ANGLE, config.MAX_RES_X) * -np.pi / 180.0)
    yf = np.sin(np.linspace(-config.MAX_ANGLE, config.MAX_ANGLE, config.MAX_RES_Y) * -np.pi / 180.0)
    d = tc[None, None, :] * xf[:, None, None] + tr[None, None, :] * yf[None, :, None]          # float64
    smallest = np.minimum(d.min(axis=2), 0.0)
    out[:, :, :n] = d                                                                             # rounded to float32 on store
    out -= smallest[:, :, None]                                                                   # float32 arithmetic, as in-place ops on the array
    out *= _samples_per_metre()
    return out


def calculate_delay_miso(azimuth, elevation):
    """directions.pyx:159-187 (legacy): whole-sample delays of one (azimuth, elevation) in degrees, int array."""
    tc, tr = _tile_factors()
    n = config.COLUMNS * config.ROWS
    out = np.zeros(n * config.ACTIVE_TILES, dtype=np.float32)
    d = tc * np.sin(azimuth * -np.pi / 180.0) + tr * np.sin(elevation * -np.pi / 180.0)
    out[:n] = d
    out -= min(float(d.min()), 0.0)
    out *= _samples_per_metre()
    return out.astype(int)
