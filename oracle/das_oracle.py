"""ctypes front-ends for the two CPU checkers.  TEST INFRASTRUCTURE ONLY (see oracle/das_oracle.c header).

  Oracle(N, X, Y, T)   -- oracle/libdas_oracle.so, the run-time-sized C restatement ("port")
  RefLib(cfg_name)     -- oracle/_ref/libref_<cfg>.so, the reference's own C compiled by oracle/build_ref.py
                          for one fixed size ("reference")

Both expose the same methods so tests can run one against the other:
  mimo_pad(signals, whole_i32, mics) / mimo_lerp(signals, delays_f32, mics) /
  mimo_convolve(signals, taps_f32, mics, vectorized) / mimo_hybrid(signals, delays_f32, mics)
returning float32 [X, Y] images (the reference views image[D] as (MAX_RES_X, MAX_RES_Y), main.pyx:190).
"""
import ctypes as C
import os

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int)


def _f(a):
    return a.ctypes.data_as(FP)


def _i(a):
    return a.ctypes.data_as(IP)


def _prep(signals, mics):
    s = np.ascontiguousarray(signals, dtype=np.float32)
    m = np.ascontiguousarray(mics, dtype=np.int32)
    return s, m


class _Base:
    pre = ""

    def _fn(self, name):
        return getattr(self.lib, self.pre + name)

    def mimo_pad(self, signals, whole, mics):
        s, m = _prep(signals, mics)
        w = np.ascontiguousarray(whole, dtype=np.int32).ravel()
        img = np.zeros((self.X, self.Y), dtype=np.float32)
        self._fn("load_coefficients_pad")(_i(w), C.c_int(w.size))
        self._fn("mimo_pad")(_f(s), _f(img), _i(m), C.c_int(m.size))
        return img

    def mimo_lerp(self, signals, delays_f32, mics):
        s, m = _prep(signals, mics)
        d = np.ascontiguousarray(delays_f32, dtype=np.float32).ravel()
        img = np.zeros((self.X, self.Y), dtype=np.float32)
        self._fn("load_coefficients_lerp")(_f(d), C.c_int(d.size))
        self._fn("mimo_lerp")(_f(s), _f(img), _i(m), C.c_int(m.size))
        return img

    def mimo_convolve(self, signals, taps, mics, vectorized=True):
        s, m = _prep(signals, mics)
        h = np.ascontiguousarray(taps, dtype=np.float32).ravel()
        img = np.zeros((self.X, self.Y), dtype=np.float32)
        self._fn("load_coefficients_convolve")(_f(h), C.c_int(h.size))
        self._fn("mimo_convolve_vectorized" if vectorized else "mimo_convolve_naive")(_f(s), _f(img), _i(m), C.c_int(m.size))
        return img

    def mimo_hybrid(self, signals, delays_f32, mics):
        s, m = _prep(signals, mics)
        d = np.ascontiguousarray(delays_f32, dtype=np.float32).ravel()
        img = np.zeros((self.X, self.Y), dtype=np.float32)
        self._fn("load_coefficients_convolve_hybrid")(_f(d), C.c_int(d.size))
        self._fn("mimo_convolve_hybrid")(_f(s), _f(img), _i(m), C.c_int(m.size))
        return img

    def miso_pad(self, signals, whole, mics, offset):
        s, m = _prep(signals, mics)
        w = np.ascontiguousarray(whole, dtype=np.int32).ravel()
        out = np.zeros(self.N, dtype=np.float32)
        self._fn("load_coefficients_pad")(_i(w), C.c_int(w.size))
        self._fn("miso_pad")(_f(s), _f(out), _i(m), C.c_int(m.size), C.c_int(offset))
        return out

    def miso_lerp(self, signals, delays_f32, mics, offset):
        s, m = _prep(signals, mics)
        d = np.ascontiguousarray(delays_f32, dtype=np.float32).ravel()
        out = np.zeros(self.N, dtype=np.float32)
        self._fn("load_coefficients_lerp")(_f(d), C.c_int(d.size))
        self._fn("miso_lerp")(_f(s), _f(out), _i(m), C.c_int(m.size), C.c_int(offset))
        return out


class Oracle(_Base):
    """The C restatement, any size."""
    pre = "oracle_"
    kind = "port"

    def __init__(self, N, X, Y, T=8):
        path = os.path.join(HERE, "libdas_oracle.so")
        if not os.path.exists(path):
            raise FileNotFoundError("%s missing: run `make -C oracle` (or __graft_entry__.build())" % path)
        self.lib = C.CDLL(path)
        self.N, self.X, self.Y, self.T = N, X, Y, T
        self.lib.oracle_configure(N, X, Y, T)

    def _fn(self, name):
        self.lib.oracle_configure(self.N, self.X, self.Y, self.T)   # the .so is shared between instances
        return getattr(self.lib, self.pre + name)

    def load(self, algo, table):
        """Load the coefficient table of `algo` (0 pad: int32 whole; 1 lerp / 2 hybrid: float32 delays; 3/4 fir: taps)."""
        if algo == 0:
            t = np.ascontiguousarray(table, dtype=np.int32).ravel()
            self._fn("load_coefficients_pad")(_i(t), C.c_int(t.size))
        else:
            t = np.ascontiguousarray(table, dtype=np.float32).ravel()
            name = {1: "load_coefficients_lerp", 2: "load_coefficients_convolve_hybrid"}.get(algo, "load_coefficients_convolve")
            self._fn(name)(_f(t), C.c_int(t.size))

    def mimo_range(self, algo, signals, mics, d0, d1):
        """Images of flat directions [d0, d1) with the table loaded by `load` (float32 [d1-d0])."""
        s, m = _prep(signals, mics)
        img = np.zeros(d1 - d0, dtype=np.float32)
        self._fn("mimo_range")(C.c_int(algo), _f(s), _f(img), _i(m), C.c_int(m.size), C.c_int(d0), C.c_int(d1))
        return img

    def lerp_tables(self, delays_f32):
        d = np.ascontiguousarray(delays_f32, dtype=np.float32).ravel()
        self._fn("load_coefficients_lerp")(_f(d), C.c_int(d.size))
        whole = np.zeros(d.size, dtype=np.int32)
        h = np.zeros(d.size, dtype=np.float32)
        self.lib.oracle_get_lerp_tables(_i(whole), _f(h), C.c_int(d.size))
        return whole, h

    def hybrid_tables(self, delays_f32):
        d = np.ascontiguousarray(delays_f32, dtype=np.float32).ravel()
        self._fn("load_coefficients_convolve_hybrid")(_f(d), C.c_int(d.size))
        whole = np.zeros(d.size, dtype=np.int32)
        taps = np.zeros(d.size * self.T, dtype=np.float32)
        self.lib.oracle_get_hybrid_tables(_i(whole), _f(taps), C.c_int(d.size))
        return whole, taps.reshape(d.size, self.T)


class RefLib(_Base):
    """The reference's own C, compiled for one fixed size by oracle/build_ref.py."""
    pre = ""
    kind = "reference"

    def __init__(self, cfg_name):
        import sys
        sys.path.insert(0, HERE)
        from configs import CONFIGS
        cfg = CONFIGS[cfg_name]
        path = os.path.join(HERE, "_ref", "libref_%s.so" % cfg_name)
        if not os.path.exists(path):
            raise FileNotFoundError(path)
        self.lib = C.CDLL(path)
        self.N, self.X, self.Y, self.T = cfg["N"], cfg["X"], cfg["Y"], cfg["T"]

    @staticmethod
    def available(cfg_name):
        return os.path.exists(os.path.join(HERE, "_ref", "libref_%s.so" % cfg_name))

    def _global(self, name, ctype, n):
        ptr = C.POINTER(ctype).in_dll(self.lib, name)
        return np.ctypeslib.as_array(ptr, shape=(n,)).copy()

    def lerp_tables(self, delays_f32):
        d = np.ascontiguousarray(delays_f32, dtype=np.float32).ravel()
        self.lib.load_coefficients_lerp(_f(d), C.c_int(d.size))
        return self._global("whole_samples_lerp", C.c_int, d.size), self._global("fractional_samples_lerp", C.c_float, d.size)

    def hybrid_tables(self, delays_f32):
        d = np.ascontiguousarray(delays_f32, dtype=np.float32).ravel()
        self.lib.load_coefficients_convolve_hybrid(_f(d), C.c_int(d.size))
        return (self._global("whole_samples_convolve", C.c_int, d.size),
                self._global("convolve_coefficients_fractional", C.c_float, d.size * self.T).reshape(d.size, self.T))
