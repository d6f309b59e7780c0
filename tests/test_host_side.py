"""CPU: the product's host side through the C-ABI -- steering tables (csrc/directions.cpp) bit-exact against the
golden vectors, exported symbols, launch planning, and loud failure when no GPU is present."""
import ctypes as C
import os
import re

import numpy as np
import pytest

import util
from util import CONFIGS, golden, sha


@pytest.mark.parametrize("name", list(CONFIGS))
def test_product_tables_bit_exact(native, name):
    from lib import directions
    c, g = util.configure(name), golden(name)
    act, n = directions.active_microphones()
    assert n == c["M"] and np.array_equal(act, g["active_mics"])
    assert directions.calc_r_prime(float(np.float32(0.02))).tobytes() == g["r_prime"].tobytes()
    d = directions.calculate_delays()
    assert d.shape == (c["X"], c["Y"], c["M"]) and d.dtype == np.float64
    assert sha(d) == str(g["delay_sha256"])
    assert sha(d.astype(int).astype(np.int32)) == str(g["whole_sha256"])
    assert sha(np.float32(d)) == str(g["delay_f32_sha256"])


def test_product_taps_bit_exact(native):
    from lib import directions
    util.configure("cfg1")
    g = golden("cfg1")
    for x, h1, h2 in zip(g["tap_probe_delay"], g["tap_probe_get_h"], g["tap_probe_get_h2"]):
        assert np.array_equal(directions.get_h(float(x)), h1)
        assert np.array_equal(directions.get_h2(float(x), N=8), h2)
    whole, h = directions.calculate_coefficients()
    assert np.array_equal(whole, g["delay"].astype(int)) and np.array_equal(h, g["taps_get_h"])
    assert np.array_equal(directions.compute_convolve_h(), g["taps_get_h2"])


def test_product_tables_match_oracle_on_other_geometry(native):
    """A geometry no fixture covers (2 tiles, every 2nd mic, other pitch/angle/distance): product C++ == NumPy oracle."""
    import directions_np as D
    from interface import config
    from lib import directions
    config.configure(N_MICROPHONES=128, ACTIVE_TILES=2, SKIP_N_MICS=2, MAX_RES_X=23, MAX_RES_Y=17, ELEMENT_DISTANCE=0.025,
                     VIEW_ANGLE=59.0, Z=1.5, SAMPLE_RATE=44100.0, PROPAGATION_SPEED=343.0)
    try:
        want = D.calculate_delays(23, 17, arrays=2, fs=44100.0, c=343.0, d=0.025, view_angle=59.0, z=1.5, skip=2)
        got = directions.calculate_delays()
        assert got.shape == want.shape and got.tobytes() == want.tobytes()
        assert np.array_equal(directions.active_microphones()[0], D.active_microphones(arrays=2, skip=2)[0])
    finally:
        config.configure(SKIP_N_MICS=1, ELEMENT_DISTANCE=0.02, Z=1.0, SAMPLE_RATE=48828.0, PROPAGATION_SPEED=340.0)


def test_every_declared_symbol_is_exported(native):
    text = open(os.path.join(util.ROOT, "include", "beamformer_hip.h")).read()
    text = re.sub(r"/\*.*?\*/", "", text, flags=re.S)
    names = re.findall(r"\b([A-Za-z_][A-Za-z0-9_]*)\s*\([^;{]*\)\s*;", text)
    assert len(names) > 50
    missing = [n for n in names if not hasattr(native.lib, n)]
    assert not missing, missing


def test_api_h_management_symbols(native):
    """PC/src/api.h:6-9,41-45: exported so that the reference's main.pyx links unchanged.  steer / load_pa / load_miso keep the
    listen state (api.c:461-489, 553-581); the receiver child is out of scope, so load() fails loudly."""
    lib = native.lib
    n = C.c_int(-1)
    assert lib.load_miso() == 0                       # miso_init_shared_memory: n = 1, adaptive_array zero, steer(0)
    assert lib.bf_get_steer(C.byref(n)) == 0 and n.value == 1
    lib.steer(12345)
    mics = np.arange(7, dtype=np.int32)
    lib.load_pa(native.iptr(mics), 7)
    assert lib.bf_get_steer(C.byref(n)) == 12345 and n.value == 7
    lib.stop_miso()
    assert lib.bf_get_steer(C.byref(n)) == 0 and n.value == 0
    lib.signal_handler()
    lib.stop_receiving()
    lib.bf_clear_error()
    assert lib.load(True) == -1 and b"out of scope" in lib.bf_last_error()
    lib.bf_clear_error()
    # no listen state -> the playback-loop body refuses (and never silently computes on the CPU)
    out = np.zeros(256, dtype=np.float32)
    assert lib.bf_miso_listen_block(native.fptr(out), C.c_float(128.0)) == -1 and np.isnan(out).all()
    lib.bf_clear_error()


def test_plan_geometry(native):
    """LDS sizing / tiling chosen on the host for the BASELINE sizes (no GPU needed)."""
    out = (C.c_longlong * 10)()
    util.configure("cfg2")
    # lerp at N = 256, one frame: shifted-copies layout, 32 mics x (2 sample copies + 2 difference copies) per chunk, one
    # 1024-thread workgroup per CU, 8 directions per wave; small delays get the compile-time row stride (lead 56)
    assert native.lib.bf_plan_das(native.LERP, 64, 1, 0, 101 * 101, 12, 256, out) == 0
    nc, lead, rs, mc, nch, waves, dpw, tile, ntiles, lds = list(out)
    assert (nc, mc, nch, waves, dpw) == (4, 32, 2, 16, 8) and lds <= 160 * 1024 and ntiles % 8 == 0
    assert (lead, rs) == (56, 312)
    # a batch: two frames per workgroup, 16 mics x 2 frames per chunk
    assert native.lib.bf_plan_das(native.LERP, 64, 190, 0, 101 * 101, 12, 256, out) == 0
    nc, lead, rs, mc, nch, waves, dpw, tile, ntiles, lds = list(out)
    assert (nc, mc, nch, waves, dpw) == (4, 16, 4, 16, 8) and lds == 16 * 2 * 4 * 312 * 4 and tile % 128 == 0 and ntiles % 8 == 0
    # tile size from the per-XCD round count: 40 tiles of 2 wave groups = 5 per XCD, 5 * 95 frame pairs on 32 CUs = 15 rounds
    assert (tile, ntiles) == (256, 40)
    assert native.lib.bf_plan_das(native.PAD, 64, 190, 0, 101 * 101, 12, 256, out) == 0
    assert list(out)[3:5] == [16, 4] and out[9] == 128 * 260 * 4      # the parked power rows outgrow the pad chunk
    # a mic count that is not a multiple of 16 keeps one frame per workgroup
    assert native.lib.bf_plan_das(native.PAD, 40, 190, 0, 101 * 101, 12, 256, out) == 0
    assert list(out)[3:5] == [32, 2]
    # larger delays: run-time row stride (the two-copy rows still leave room for 16 mics per chunk)
    assert native.lib.bf_plan_das(native.LERP, 64, 190, 0, 101 * 101, 95, 256, out) == 0
    nc, lead, rs, mc, nch, waves, dpw, tile, ntiles, lds = list(out)
    assert (lead, rs, mc, nch) == (100, 356, 16, 4) and lds <= 160 * 1024
    # N = 1024: one strided copy per mic row, chunked over the mics, 4 directions carried per wave
    util.configure("cfg5")
    assert native.lib.bf_plan_das(native.LERP, 256, 1, 0, 361 * 361, 47, 256, out) == 0
    nc, lead, rs, mc, nch, waves, dpw, tile, ntiles, lds = list(out)
    assert nc == 16 and nch * mc >= 256 and nch > 1 and dpw == 4 and lds <= 160 * 1024
    # unsupported shapes are refused with a message, not launched
    assert native.lib.bf_plan_das(native.FIR_VEC, 64, 1, 0, 1, 0, 256, out) == 0
    assert native.lib.bf_configure(64, 4096, 11, 11, 8) == -1
    with pytest.raises(native.BeamformerError):
        native.check()
    util.configure("cfg1")


def test_config_json_roundtrip(native, tmp_path):
    p = tmp_path / "config.json"
    p.write_text('{"general": {"N_MICROPHONES": 64, "N_SAMPLES": 128, "MAX_RES_X": 9, "MAX_RES_Y": 7, "N_TAPS": 16}}')
    assert native.lib.bf_configure_from_json(str(p).encode()) == 0
    out = (C.c_int * 5)()
    native.lib.bf_get_config(out)
    assert list(out) == [64, 128, 9, 7, 16]
    assert native.lib.bf_configure_from_json(b"/nonexistent.json") == -1
    with pytest.raises(native.BeamformerError):
        native.check()
    util.configure("cfg1")


def test_fails_loudly_without_gpu(native):
    """No silent CPU path: without a GPU every compute entry point reports an error and returns NaN."""
    if native.gpu_available():
        pytest.skip("a GPU is present")
    from lib import tests as T
    import synth
    util.configure("cfg1")
    with pytest.raises(native.BeamformerError, match="no usable HIP device"):
        T.mimo_pad_wrapper(synth.s2_noise(64, 256))
    img = np.zeros(121, dtype=np.float32)
    sig = synth.s2_noise(64, 256)
    mics = np.arange(64, dtype=np.int32)
    native.lib.mimo_pad(native.fptr(sig), native.fptr(img), native.iptr(mics), 64)
    assert np.isnan(img).all()
    native.lib.bf_clear_error()


def test_legacy_direction_helpers_match_the_reference(native):
    """lib.directions.calculate_delays_ / calculate_delay_miso (directions.pyx:126-187) against vectors produced by the
    real reference (oracle/gen_golden_legacy.py), bit for bit; plus the pass-through constants of interface.config."""
    import hashlib
    from interface import config
    from lib import directions as D
    util.configure("shipped")
    g = np.load(os.path.join(util.GOLDEN, "legacy_directions.npz"))
    d = D.calculate_delays_()
    assert d.dtype == np.float32 and tuple(d.shape) == tuple(g["delays_shape"])
    assert np.array_equal(d[::8, ::5, :], g["delays_sample"])
    assert hashlib.sha256(np.ascontiguousarray(d).tobytes()).hexdigest() == str(g["delays_sha256"])
    for (az, el), want in zip(g["angles"], g["miso"]):
        assert np.array_equal(D.calculate_delay_miso(float(az), float(el)), want)
    assert config.MAX_ANGLE == 70.0 and abs(config.ASPECT_RATIO - 4 / 3) < 1e-15 and config.WINDOW_SIZE == (720, 480)
    util.configure("cfg1")


def test_visual_host_helpers(native):
    """visual.generate_color_map / local_max (PC/src/visual.py:26-63): the colour table equals the matplotlib-derived
    fixture the colourise kernel is tested against; local_max marks >=-neighbour peaks above the threshold."""
    import visual
    assert np.array_equal(visual.generate_color_map(), np.load(os.path.join(util.GOLDEN, "jet_lut.npy")))
    img = np.array([[0, 1, 0, 0], [1, 5, 1, 0], [0, 1, 6.0, 0.4]])
    want = np.zeros_like(img, dtype=bool)
    want[1, 1] = want[2, 2] = True
    assert np.array_equal(visual.local_max(img, 0.5), want)
