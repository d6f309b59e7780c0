"""Mirror of the reference's `lib.tests` module (PC/src/benchmark.pyx:71-195): the offline wrappers
`signals -> image` that PC/plot.py drives, same names and semantics, running on the MI355X.

`signals` is float32 [N_MICROPHONES, N_SAMPLES]; every wrapper returns float32 [MAX_RES_X, MAX_RES_Y].
Unlike the reference, which recomputes its Python tap tables on every call, the steering tables are computed
once per configuration and cached; the tables themselves are identical."""
import numpy as np

from interface import config
from . import _native as nat
from .directions import active_microphones, calculate_delays, compute_convolve_h

_cache = {}


def _key():
    return (config.N_MICROPHONES, config.N_SAMPLES, config.MAX_RES_X, config.MAX_RES_Y, config.N_TAPS, config.ACTIVE_TILES,
            config.SKIP_N_MICS, config.ELEMENT_DISTANCE, config.VIEW_ANGLE, config.Z, config.SAMPLE_RATE, config.PROPAGATION_SPEED)


def _cached(name, fn):
    k = (name,) + _key()
    if k not in _cache:
        _cache[k] = fn()
    return _cache[k]


def _mics():
    active, n = active_microphones()
    return np.ascontiguousarray(active.astype(np.int32)), int(n)


def _image():
    return np.zeros((config.MAX_RES_X, config.MAX_RES_Y), dtype=np.float32)


def _signals(signals):
    s = nat.f32c(signals, 2)
    if s.shape[1] != config.N_SAMPLES:
        raise ValueError("signals must be [mics, N_SAMPLES=%d], got %s" % (config.N_SAMPLES, s.shape))
    return s


def pad_coefficients_load(whole_samples, n=None):
    """benchmark.pyx:57-72"""
    w = np.ascontiguousarray(np.asarray(whole_samples).astype(np.int32))
    nat.lib.load_coefficients_pad(nat.iptr(w), int(w.size))
    nat.check()


def convolve_coefficients_load(h):
    """benchmark.pyx:112-117"""
    h = nat.f32c(h)
    nat.lib.load_coefficients_convolve(nat.fptr(h), int(h.size))
    nat.check()


def pad_delay_wrapper(signal, out, pos_pad):
    """benchmark.pyx:74-82: out += signal delayed by pos_pad samples; returns the updated copy of `out`."""
    s = nat.f32c(signal, 1)
    o = nat.f32c(out, 1).copy()
    nat.lib.pad_delay(nat.fptr(s), nat.fptr(o), int(pos_pad))
    nat.check()
    return o


def mimo_pad_wrapper(signals):
    """benchmark.pyx:84-110"""
    s, img = _signals(signals), _image()
    mics, n = _mics()
    whole = _cached("whole", lambda: np.ascontiguousarray(calculate_delays().astype(int).astype(np.int32)))
    nat.lib.load_coefficients_pad(nat.iptr(whole), int(whole.size))
    nat.check()
    nat.lib.mimo_pad(nat.fptr(s), nat.fptr(img), nat.iptr(mics), n)
    nat.check()
    return img


def mimo_convolve_wrapper(signals):
    """benchmark.pyx:119-138"""
    s, img = _signals(signals), _image()
    mics, n = _mics()
    h = _cached("conv_h", compute_convolve_h)
    convolve_coefficients_load(h)
    nat.lib.mimo_convolve_vectorized(nat.fptr(s), nat.fptr(img), nat.iptr(mics), n)
    nat.check()
    return img


def mimo_lerp_wrapper(signals):
    """benchmark.pyx:141-162"""
    s, img = _signals(signals), _image()
    mics, n = _mics()
    d32 = _cached("delay_f32", lambda: np.ascontiguousarray(np.float32(calculate_delays())))
    nat.lib.load_coefficients_lerp(nat.fptr(d32), int(d32.size))
    nat.check()
    nat.lib.mimo_lerp(nat.fptr(s), nat.fptr(img), nat.iptr(mics), n)
    nat.check()
    nat.lib.unload_coefficients_lerp()
    return img


def mimo_hybrid_convolve_wrapper(signals):
    """benchmark.pyx:164-186"""
    s, img = _signals(signals), _image()
    mics, n = _mics()
    d32 = _cached("delay_f32", lambda: np.ascontiguousarray(np.float32(calculate_delays())))
    nat.lib.load_coefficients_convolve_hybrid(nat.fptr(d32), int(d32.size))
    nat.check()
    nat.lib.mimo_convolve_hybrid(nat.fptr(s), nat.fptr(img), nat.iptr(mics), n)
    nat.check()
    return img
