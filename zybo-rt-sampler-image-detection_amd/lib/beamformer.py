"""Mirror of the reference's `lib.beamformer` module (PC/src/main.pyx) for the hot path: stream hand-off, steering
index arithmetic and the producer loops that turn the newest frame into a power map -- same names and arguments.

Out of scope here (kept in the reference): the UDP receiver child process, PortAudio playback, camera / pcap capture
and the OpenCV demo entry points.  `connect()` therefore does not fork a receiver; a frame source (replay file, test,
or a real receiver process) hands each window over with `publish(signals)`, which is what `get_data()` returns."""
import queue

import numpy as np

from interface import config
from . import _native as nat
from .directions import active_microphones, calculate_coefficients, calculate_delays, compute_convolve_h

_connected = False
_steer_offset = 0


def connect(replay_mode: bool = False, verbose=True) -> None:
    """main.pyx:95-120.  Marks the stream open; frames arrive through publish()."""
    global _connected
    assert isinstance(replay_mode, bool), "Replay mode must be either True or False"
    _connected = True
    if verbose:
        print("Frame hand-off ready (bf_publish_frame).\nContinue your program!\n")


def disconnect() -> None:
    """main.pyx:122-131"""
    global _connected
    _connected = False


def publish(signals) -> None:
    """Hand the newest window [N_MICROPHONES, N_SAMPLES] float32 to the library (what the receiver child does through
    shared memory in the reference, api.c:903-930)."""
    assert signals.shape == (config.N_MICROPHONES, config.N_SAMPLES), "Arrays do not match shape"
    s = nat.f32c(signals)
    nat.lib.bf_publish_frame(nat.fptr(s))
    nat.check()


def receive(signals) -> None:
    """main.pyx:133-159: fill `signals` with the newest window (dead-microphone rows zeroed as api.c:835-858 does)."""
    assert signals.shape == (config.N_MICROPHONES, config.N_SAMPLES), "Arrays do not match shape"
    assert signals.dtype == np.float32, "Arrays dtype do not match"
    assert signals.flags["C_CONTIGUOUS"]
    nat.lib.get_data(nat.fptr(signals))
    nat.check()


def steer_cartesian_degree(azimuth: float, elevation: float):
    """main.pyx:498-515: degrees -> flat table offset of the steered direction (and select it for MISO listening)."""
    assert -90 <= azimuth <= 90, "Invalid range"
    assert -90 <= elevation <= 90, "Invalid range"
    azimuth = int((azimuth + 90) / 180 * config.MAX_RES_X)
    elevation = int((elevation + 90) / 180 * config.MAX_RES_Y)
    _, n = active_microphones()
    return steer(int(elevation * config.MAX_RES_X * n + azimuth * n))


def stear_miso_beam(azimuth: float, elevation: float):
    """main.pyx:517-528: normalised image coordinates in [0, 1) -> flat table offset."""
    azimuth = int(azimuth * config.MAX_RES_X)
    elevation = int(elevation * config.MAX_RES_Y)
    _, n = active_microphones()
    return steer(int(elevation * config.MAX_RES_X * n + azimuth * n))


def steer(offset: int) -> int:
    """api.c:576-581"""
    global _steer_offset
    _steer_offset = int(offset)
    return _steer_offset


def listen(out=None):
    """One block of the steered beam (api.c:1097-1104 miso_steer_listen): raw sum over the active mics, float32 [N_SAMPLES]."""
    if out is None:
        out = np.zeros(config.N_SAMPLES, dtype=np.float32)
    mics, n = _mics()
    nat.lib.miso_steer_listen(nat.fptr(out), nat.iptr(mics), n, _steer_offset)
    nat.check()
    return out


def _mics():
    active, n = active_microphones()
    return np.ascontiguousarray(active.astype(np.int32)), int(n)


def _running(flag):
    return flag.value if hasattr(flag, "value") else bool(flag)


def _produce(q, running, load, step, unload, max_frames=None):
    """Common body of the producer loops (main.pyx:172-202, 383-404): tables once, then one map per newest frame."""
    load()
    mics, n = _mics()
    frame_nr = 0
    while _running(running):
        power_map = np.zeros((config.MAX_RES_X, config.MAX_RES_Y), dtype=np.float32)
        try:
            step(power_map, mics, n)
            nat.check()
        except nat.BeamformerError:
            break
        frame_nr += 1
        try:
            q.put((power_map, frame_nr), timeout=1.0)
        except queue.Full:
            pass
        if max_frames is not None and frame_nr >= max_frames:
            break
    unload()


def _load_pad():
    whole, _ = calculate_coefficients()
    w = np.ascontiguousarray(whole.astype(np.int32))
    nat.lib.load_coefficients_pad(nat.iptr(w), int(w.size))
    nat.check()


def _load_lerp():
    d = np.ascontiguousarray(calculate_delays().astype(np.float32))
    nat.lib.load_coefficients_lerp(nat.fptr(d), int(d.size))
    nat.check()


def _load_convolve():
    h = np.ascontiguousarray(compute_convolve_h())
    nat.lib.load_coefficients_convolve(nat.fptr(h), int(h.size))
    nat.check()


def b(q, running, max_frames=None):
    """main.pyx:569-570 -> _loop_mimo_pad (:172-202): (power_map, frame_nr) per frame with the pad beamformer."""
    _produce(q, running, _load_pad, lambda img, m, n: nat.lib.pad_mimo(nat.fptr(img), nat.iptr(m), n), nat.lib.unload_coefficients_pad, max_frames)


def uti_api(q, running, max_frames=None):
    """main.pyx:554-555 -> api (:383-404)"""
    b(q, running, max_frames)


def multi_lerp(q_steer, q_out, running, max_frames=None):
    """main.pyx:819-820 -> _loop_mimo_and_miso_lerp (:330-380): lerp maps out, steering requests in."""
    def step(img, m, n):
        nat.lib.lerp_mimo(nat.fptr(img), nat.iptr(m), n)
        try:
            x, y = q_steer.get(block=False)
            stear_miso_beam(x, y)
        except queue.Empty:
            pass
    _produce(q_out, running, _load_lerp, step, nat.lib.unload_coefficients_lerp, max_frames)


def conv_api(q, running, max_frames=None):
    """main.pyx:560-561 -> api_convolve (:477-495)"""
    _produce(q, running, _load_convolve, lambda img, m, n: nat.lib.convolve_mimo_vectorized(nat.fptr(img), nat.iptr(m), n),
             nat.lib.unload_coefficients_convolve, max_frames)
