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
