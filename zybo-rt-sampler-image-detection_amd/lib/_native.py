"""ctypes binding of lib/libbeamformer_hip.so (C-ABI: include/beamformer_hip.h).

Importing this module loads the HIP library or raises -- the Python layer has no other compute path."""
import ctypes as C
import os

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
# $BF_NATIVE_LIB: another build of the same library (kernel A/B runs); the default is the in-tree build
LIB_PATH = os.environ.get("BF_NATIVE_LIB") or os.path.join(_HERE, "libbeamformer_hip.so")


class BeamformerError(RuntimeError):
    """An entry point of libbeamformer_hip.so reported a failure (bf_last_error)."""


if not os.path.exists(LIB_PATH):
    raise ImportError("%s is missing: build it with `python -c 'import __graft_entry__ as g; g.build()'` "
                      "(hipcc --offload-arch=gfx950).  There is no CPU fallback." % LIB_PATH)

# One HIP runtime per process: PyTorch bundles its own libamdhip64.so (same soname).  Loaded first, it also serves
# this library's DT_NEEDED; loaded second, the process would hold two runtimes whose streams cannot be mixed.
if os.environ.get("BF_NO_TORCH", "0") != "1":
    try:
        import torch  # noqa: F401
    except Exception:  # torch is optional for the host-pointer API
        pass

lib = C.CDLL(LIB_PATH)

FP = C.POINTER(C.c_float)
IP = C.POINTER(C.c_int)
DP = C.POINTER(C.c_double)


class Geometry(C.Structure):
    _fields_ = [("rows", C.c_int), ("columns", C.c_int), ("arrays", C.c_int), ("skip_n_mics", C.c_int),
                ("sample_rate", C.c_float), ("propagation_speed", C.c_float), ("element_distance", C.c_float),
                ("view_angle", C.c_float), ("z", C.c_float)]


def _sig(name, res, *args):
    try:
        fn = getattr(lib, name)
    except AttributeError:
        if os.environ.get("BF_NATIVE_LIB"):      # an older build loaded for a same-box A/B run: entry points added since are simply absent
            return None
        raise
    fn.restype = res
    fn.argtypes = list(args)
    return fn


# PART 1 (reference symbols)
for _n in ("mimo_pad", "mimo_lerp", "mimo_convolve_naive", "mimo_convolve_vectorized", "mimo_convolve_hybrid"):
    _sig(_n, None, FP, FP, IP, C.c_int)
for _n in ("miso_pad", "miso_pad2", "miso_lerp", "miso_convolve_vectorized", "miso_convolve_hybrid"):
    _sig(_n, None, FP, FP, IP, C.c_int, C.c_int)
for _n in ("load_coefficients_pad", "load_coefficients_pad2", "load_coefficients2"):
    _sig(_n, None, IP, C.c_int)
for _n in ("load_coefficients_lerp", "load_coefficients_convolve", "load_coefficients_convolve_hybrid"):
    _sig(_n, None, FP, C.c_int)
for _n in ("unload_coefficients_pad", "unload_coefficients_pad2", "unload_coefficients_lerp", "unload_coefficients_convolve",
           "unload_coefficients_convolve_hybrid"):
    _sig(_n, None)
_sig("pad_delay", None, FP, FP, C.c_int)
_sig("lerp_delay", None, FP, FP, C.c_float, C.c_int)
_sig("convolve_delay_naive_add", None, FP, FP, FP)
_sig("convolve_delay_vectorized", None, FP, FP, FP)
_sig("convolve_delay_vectorized_add", None, FP, FP, FP)
_sig("convolve_delay_naive", None, FP, FP, FP)
_sig("convolve_hybrid_delay_add", None, FP, FP, C.c_int, FP)
_sig("get_data", None, FP)
for _n in ("pad_mimo", "lerp_mimo", "convolve_mimo_naive", "convolve_mimo_vectorized", "mimo_truncated"):
    _sig(_n, None, FP, IP, C.c_int)
_sig("miso_steer_listen", None, FP, IP, C.c_int, C.c_int)
_sig("load", C.c_int, C.c_bool)
for _n in ("stop_receiving", "signal_handler", "stop_miso"):
    _sig(_n, None)
_sig("load_miso", C.c_int)
_sig("load_pa", None, IP, C.c_int)
_sig("steer", None, C.c_int)
# PART 2 (extensions)
_sig("bf_configure", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int)
_sig("bf_configure_from_json", C.c_int, C.c_char_p)
_sig("bf_get_config", None, IP)
_sig("bf_last_error", C.c_char_p)
_sig("bf_clear_error", None)
_sig("bf_gpu_available", C.c_int)
_sig("bf_last_das_variant", C.c_int)
_sig("bf_set_device", C.c_int, C.c_int)
_sig("bf_set_debug", None, C.c_int)
_sig("bf_read_phase_stamps", C.c_int, C.POINTER(C.c_ulonglong), C.c_int)
_sig("bf_publish_frame", None, FP)
_sig("bf_miso_listen_block", C.c_int, FP, C.c_float)
_sig("bf_get_steer", C.c_int, IP)
_sig("bf_das_device", C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, IP, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_plan_das", C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.POINTER(C.c_longlong))
_sig("bf_ingest", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, FP)
_sig("bf_ingest_device", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
_sig("bf_heatmap_colorize_device", C.c_int, C.c_void_p, C.c_int, C.c_float, C.c_float, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_heatmap_overlay_device", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_void_p)
_sig("bf_power_center_device", C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_fd_steering_device", C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_fd_dft_device", C.c_int, C.c_void_p, C.c_int, C.c_int, IP, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_fd_das_power_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
_sig("bf_fd_covariance_device", C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_fd_cholesky_inverse_device", C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_fd_mvdr_power_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p)
_sig("bf_yolo_decode_device", C.c_int, C.POINTER(C.c_void_p), IP, IP, IP, FP, C.c_int, C.c_int, C.c_int, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_upsample_concat_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_sppf_pool_device", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_preprocess_bgr8_device", C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_conv2d_weight_row", C.c_int, C.c_int, C.c_int, C.c_int)
_sig("bf_conv2d_nhwc_f16_into_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_conv2d_nhwc_f16_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_letterbox_bgr8_device", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_conv2d_use_dma_kernel", C.c_int, C.c_int)
_sig("bf_conv2d_f32_mode", C.c_int, C.c_int)
_sig("bf_fd_gemm_f32_mode", C.c_int, C.c_int)
_sig("bf_conv2d_weight_row_f32", C.c_int, C.c_int, C.c_int, C.c_int)
_sig("bf_conv2d_nhwc_f32_into_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_conv2d_nhwc_f32_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_upsample_concat_f32_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_sppf_pool_f32_device", C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_preprocess_bgr8_f32_device", C.c_int, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
for _n in ("bf_conv1x1_cat_nhwc_f16_device", "bf_conv1x1_cat_nhwc_f32_device"):
    _sig(_n, C.c_int, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_void_p, C.c_int,
         C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_int, C.c_void_p)
_sig("bf_topk_candidates_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_nms_device", C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int, C.c_int, C.c_float, C.c_int, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p)
_sig("bf_jet_lut", None, C.POINTER(C.c_ubyte))
_sig("bf_get_lerp_tables", C.c_int, IP, FP, C.c_int)
_sig("bf_get_hybrid_tables", C.c_int, IP, FP, C.c_int)
_sig("bf_default_geometry", None, C.POINTER(Geometry))
_sig("bf_active_microphones", C.c_int, C.POINTER(Geometry), IP, C.c_int, IP)
_sig("bf_calc_r_prime", C.c_int, C.POINTER(Geometry), IP, C.c_int, DP)
_sig("bf_calculate_delays", C.c_int, C.POINTER(Geometry), C.c_int, C.c_int, IP, C.c_int, DP)
_sig("bf_get_h", None, C.c_double, DP)
_sig("bf_get_h2", None, C.c_double, C.c_int, FP)
_sig("bf_get_h_batch", None, DP, C.c_longlong, FP)
_sig("bf_get_h2_batch", None, DP, C.c_longlong, C.c_int, FP)

PAD, LERP, HYBRID, FIR_NAIVE, FIR_VEC = range(5)


def fptr(a):
    return a.ctypes.data_as(FP)


def iptr(a):
    return a.ctypes.data_as(IP)


def dptr(a):
    return a.ctypes.data_as(DP)


def check():
    """Raise BeamformerError if the last native call recorded a failure."""
    msg = lib.bf_last_error()
    if msg:
        text = msg.decode()
        lib.bf_clear_error()
        raise BeamformerError(text)


def apply_config():
    """Push interface.config's sizes into the native library (bf_configure)."""
    from interface import config as cfg
    if lib.bf_configure(cfg.N_MICROPHONES, cfg.N_SAMPLES, cfg.MAX_RES_X, cfg.MAX_RES_Y, cfg.N_TAPS) != 0:
        check()


def geometry():
    from interface import config as cfg
    g = Geometry()
    lib.bf_default_geometry(C.byref(g))
    g.rows, g.columns, g.arrays, g.skip_n_mics = cfg.ROWS, cfg.COLUMNS, cfg.ACTIVE_TILES, cfg.SKIP_N_MICS
    g.sample_rate, g.propagation_speed = cfg.SAMPLE_RATE, cfg.PROPAGATION_SPEED
    g.element_distance, g.view_angle, g.z = cfg.ELEMENT_DISTANCE, cfg.VIEW_ANGLE, cfg.Z
    return g


def gpu_available():
    return bool(lib.bf_gpu_available())


def f32c(a, ndim=None):
    a = np.ascontiguousarray(a, dtype=np.float32)
    if ndim is not None and a.ndim != ndim:
        raise ValueError("expected a %d-d array, got shape %s" % (ndim, a.shape))
    return a


apply_config()
