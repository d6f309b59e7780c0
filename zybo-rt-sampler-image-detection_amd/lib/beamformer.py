"""Mirror of the reference's `lib.beamformer` module (PC/src/main.pyx) for the hot path: stream hand-off, steering
index arithmetic and the producer loops that turn the newest frame into a power map -- same names and arguments.

Out of scope here (kept in the reference): the UDP receiver child process, PortAudio playback, camera / pcap capture
and the OpenCV demo entry points.  `connect()` therefore does not fork a receiver; a frame source (replay file, test,
or a real receiver process) hands each window over with `publish(signals)`, which is what `get_data()` returns.
Where the reference's loops start audio playback (`load_miso` / `load_pa`, api.c:491-575), this module offers the steered
block instead: set `audio_sink` to a callable and the MISO loops call it with every `listen()` block."""
import queue

import numpy as np

from interface import config
from . import _native as nat
from .directions import active_microphones, calculate_coefficients, calculate_delays, compute_convolve_h

_connected = False
_steer_offset = 0
audio_sink = None      # optional callable(float32 [N_SAMPLES]): receives the steered beam where the reference plays it


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
    nat.lib.steer(_steer_offset)
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
        _take_steering(q_steer, block=False)
    _produce(q_out, running, _load_lerp, step, nat.lib.unload_coefficients_lerp, max_frames)


def conv_api(q, running, max_frames=None):
    """main.pyx:560-561 -> api_convolve (:477-495)"""
    _produce(q, running, _load_convolve, lambda img, m, n: nat.lib.convolve_mimo_vectorized(nat.fptr(img), nat.iptr(m), n),
             nat.lib.unload_coefficients_convolve, max_frames)


def multi_pad(q_steer, q_out, running, max_frames=None):
    """main.pyx:816-817 -> _loop_mimo_and_miso_pad (:279-328): pad maps out, steering requests (x, y) in [0, 1) in."""
    steer_cartesian_degree(0, 0)

    def step(img, m, n):
        nat.lib.pad_mimo(nat.fptr(img), nat.iptr(m), n)
        _take_steering(q_steer, block=False)
        _play()
    _produce(q_out, running, _load_pad, step, nat.lib.unload_coefficients_pad, max_frames)


def uti_api_with_miso(q, running, max_frames=None):
    """main.pyx:557-558 -> api_with_miso (:417-446): the pad maps of uti_api while the beam listens at zero bearing."""
    steer_cartesian_degree(0, 0)

    def step(img, m, n):
        nat.lib.pad_mimo(nat.fptr(img), nat.iptr(m), n)
        _play()
    _produce(q, running, _load_pad, step, nat.lib.unload_coefficients_pad, max_frames)


def miso_api(q, running, max_frames=None):
    """main.pyx:563-564 -> api_miso (:531-550): one steered block [N_SAMPLES] per newest frame."""
    _load_pad()
    n_out = 0
    while _running(running):
        try:
            q.put(listen().copy(), timeout=1.0)
        except queue.Full:
            pass
        except nat.BeamformerError:
            break
        n_out += 1
        if max_frames is not None and n_out >= max_frames:
            break


def just_miso_api(q, running, max_frames=None, poll_s=0.1):
    """main.pyx:566-567 -> just_miso (:448-475): tables loaded, beam at zero bearing, nothing queued; the reference's
    PortAudio callback does the listening -- here `audio_sink`, if set, is fed once per poll."""
    import time
    _load_pad()
    steer_cartesian_degree(0, 0)
    n_out = 0
    while _running(running):
        if audio_sink is not None:
            _play()
        else:
            time.sleep(poll_s)
        n_out += 1
        if max_frames is not None and n_out >= max_frames:
            break
    nat.lib.unload_coefficients_pad()


def _pure_miso(q, running, load, unload, max_requests):
    load()
    steer_cartesian_degree(0, 0)
    n_req = 0
    while _running(running):
        if not _take_steering(q, block=True, timeout=0.5):
            continue
        _play()
        n_req += 1
        if max_requests is not None and n_req >= max_requests:
            break
    unload()


def pure_miso_pad(q, running, max_requests=None):
    """main.pyx:810-811 -> _loop_miso_pad (:204-241): steering requests in, nothing out (audio only)."""
    _pure_miso(q, running, _load_pad, nat.lib.unload_coefficients_pad, max_requests)


def pure_miso_lerp(q, running, max_requests=None):
    """main.pyx:813-814 -> _loop_miso_lerp (:243-277)"""
    _pure_miso(q, running, _load_lerp, nat.lib.unload_coefficients_lerp, max_requests)


def just_miso_loop(q, running):
    """main.pyx:572-580: idle until `running` clears."""
    import time
    while _running(running):
        time.sleep(0.1)


def _take_steering(q, block, timeout=None):
    """One (x, y) request off a steering queue -> stear_miso_beam; False when none was waiting."""
    try:
        x, y = q.get(block=block, timeout=timeout)
    except queue.Empty:
        return False
    if hasattr(q, "task_done"):
        q.task_done()
    stear_miso_beam(x, y)
    return True


def _play():
    if audio_sink is not None:
        audio_sink(listen())
