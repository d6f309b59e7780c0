"""MI355X-native delay-and-sum beamforming hot path (drop-in for the reference's PC/src CPU/Cython path).

The directory name contains '-' (it mirrors the reference repository's name), so it is not importable with a
plain `import`.  Use it the way the reference's PC/ directory is used: put this directory on sys.path and
import the reference's module names,

    sys.path.insert(0, "<repo>/zybo-rt-sampler-image-detection_amd")
    from lib.tests import mimo_pad_wrapper, mimo_lerp_wrapper      # PC/plot.py:5
    from lib.directions import calculate_delays                     # PC/plot.py:6
    from interface import config

or call `__graft_entry__.load_package()` which does that and returns the three modules.
All numerical work happens in lib/libbeamformer_hip.so (HIP, gfx950); importing `lib._native` raises if that
library has not been built -- there is no Python or CPU fallback.
"""
