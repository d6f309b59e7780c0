"""Run-time counterpart of the reference's generated `interface/config.py` (PC/src/build_config.py:18-60).

The reference bakes these constants in at build time; here they are read from a JSON file laid out like
PC/src/config.json -- $BF_CONFIG if set, else <package>/src/config.json -- and can be changed at run time with
`configure(...)`, which also re-sizes the native library (bf_configure).  Attribute names are the reference's.
"""
import json
import os

_HERE = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
_PATH = os.environ.get("BF_CONFIG", os.path.join(_HERE, "src", "config.json"))

N_MICROPHONES = 256
N_SAMPLES = 256
N_TAPS = 8
COLUMNS = 8
ROWS = 8
MAX_RES_X = 57
MAX_RES_Y = 32
Z = 1.0
VIEW_ANGLE = 59.0
SAMPLE_RATE = 48828.0
ELEMENT_DISTANCE = 0.02
SKIP_N_MICS = 1
PROPAGATION_SPEED = 340.0
ACTIVE_TILES = 4          # `_ACTIVE_MICS` of PC/src/directions.pyx:16 (hard-coded there)
fs = 48828                # PC/src/config.json:47 (python section)
# The rest of the reference's generated constants (PC/src/config.json "general" / "python"): not used by the beamforming
# path, carried so that `from interface.config import ...` lines of the reference's scripts resolve.
EVERY_N_SAMPLES = 1
MAX_RES = 20
MAX_ANGLE = 70.0
UDP_PORT = 21844
ARRAY_SEPARATION = 0.0
ACTIVE_ARRAYS = 3
APPLICATION_WINDOW_WIDTH = 720
APPLICATION_WINDOW_HEIGHT = 480
CAMERA_SOURCE = 2
FLIP_IMAGE = True
APPLICATION_NAME = "BEEEEEAAAAAAM FOOOOOOORMING"
UDP_IP = "10.0.0.1"
UDP_REPLAY_IP = "127.0.0.1"
FPGA_PROTOCOL_VERSION = 2
ASPECT_RATIO = 4 / 3
USE_COMPUTER_VISION = True
azimuth = 0.0
elevation = 0.0
columns = 8
rows = 8
distance = 0.02
propagation_speed = 340.0
TIMEOUT = 30
mode = 1
modes = 7
WINDOW_SIZE = (720, 480)

_SIZE_KEYS = ("N_MICROPHONES", "N_SAMPLES", "N_TAPS", "MAX_RES_X", "MAX_RES_Y", "COLUMNS", "ROWS", "SKIP_N_MICS", "ACTIVE_TILES")
_FLOAT_KEYS = ("Z", "VIEW_ANGLE", "SAMPLE_RATE", "ELEMENT_DISTANCE", "PROPAGATION_SPEED")


def _load(path):
    with open(path) as f:
        general = json.load(f).get("general", {})
    g = globals()
    for k in _SIZE_KEYS:
        if k in general:
            g[k] = int(general[k])
    for k in _FLOAT_KEYS:
        if k in general:
            g[k] = float(general[k])
    for k, v in general.items():            # pass-through constants (no expressions are evaluated)
        if k not in _SIZE_KEYS and k not in _FLOAT_KEYS and k != "expression" and k in g and not k.startswith("_"):
            g[k] = v
    g["BUFFER_LENGTH"] = g["N_SAMPLES"] * g["N_MICROPHONES"]


def configure(**kw):
    """Change sizes/geometry at run time, e.g. configure(N_MICROPHONES=64, ACTIVE_TILES=1, MAX_RES_X=101, MAX_RES_Y=101).
    Drops every table loaded in the native library (the sizes are part of their shape)."""
    g = globals()
    for k, v in kw.items():
        if k in _SIZE_KEYS:
            g[k] = int(v)
        elif k in _FLOAT_KEYS:
            g[k] = float(v)
        else:
            raise KeyError("unknown config key %r" % k)
    g["BUFFER_LENGTH"] = g["N_SAMPLES"] * g["N_MICROPHONES"]
    from lib import _native
    _native.apply_config()


if os.path.exists(_PATH):
    _load(_PATH)
BUFFER_LENGTH = N_SAMPLES * N_MICROPHONES
