"""ctypes binding of libpsvo_hip.so (the C ABI declared in include/psvo_hip.h).

The product path has NO fallback: if the shared library is missing or a call returns a
non-zero status this module raises.  Nothing here imports `oracle/`.
"""
import ctypes
import os

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.environ.get("PSVO_HIP_LIB") or os.path.join(_HERE, "csrc", "libpsvo_hip.so")

PSVO_OK = 0
PSVO_ERR_INVALID = -1
PSVO_ERR_UNSUPPORTED = -2
PSVO_ERR_HIP = -3
PSVO_TUNE_BSIM_BWD = 1      # psvo_set_tuning keys (include/psvo_hip.h)
PSVO_TUNE_ROWS_BWD = 2
PSVO_TUNE_L2_SPLIT = 3
PSVO_TUNE_SKEW = 4
PSVO_TUNE_WGRAD2 = 5
PSVO_TUNE_FILTER_BWD = 6


class PsvoHipError(RuntimeError):
    """A libpsvo_hip entry point returned a non-zero status."""


class psvo_desc(ctypes.Structure):
    _fields_ = [(n, ctypes.c_int32) for n in
                ("B", "T", "N", "M", "Dx", "Dy", "H", "resample", "two_q", "bootstrap", "emission", "layers")]


class psvo_mlp(ctypes.Structure):
    _fields_ = [(n, ctypes.c_void_p) for n in ("W1", "b1", "W2", "b2", "Wh", "bh")]


_P = ctypes.c_void_p
_DESC = ctypes.POINTER(psvo_desc)
_MLP = ctypes.POINTER(psvo_mlp)

# name -> (restype, argtypes); must list every symbol include/psvo_hip.h declares
SIGNATURES = {
    "psvo_abi_version": (ctypes.c_int, []),
    "psvo_status_string": (ctypes.c_char_p, [ctypes.c_int]),
    "psvo_last_hip_error": (ctypes.c_char_p, []),
    "psvo_filter_forward": (ctypes.c_int, [_DESC, _MLP, _MLP, _MLP] + [_P] * 20 + [_P]),
    "psvo_filter_acc_size": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "psvo_filter_ws_floats": (ctypes.c_longlong, [ctypes.c_int] * 5),
    "psvo_filter_backward": (ctypes.c_int, [_DESC, _MLP, _MLP, _MLP] + [_P] * 18 + [ctypes.c_int] + [_P] * 16),
    "psvo_filter_forward_cov": (ctypes.c_int, [_DESC, _MLP, _MLP, _MLP] + [_P] * 22 + [_P]),
    "psvo_filter_cov_ws_floats": (ctypes.c_longlong, [ctypes.c_int] * 5),
    "psvo_filter_backward_cov": (ctypes.c_int, [_DESC, _MLP, _MLP, _MLP] + [_P] * 39 + [_P]),
    "psvo_bsim_forward_cov": (ctypes.c_int, [_DESC] + [_P] * 4 + [_MLP, _MLP, _MLP] + [_P] * 23 + [_P]),
    "psvo_bsim_backward_cov": (ctypes.c_int, [_DESC] + [_P] * 4 + [_MLP, _MLP, _MLP] + [_P] * 38 + [_P]),
    "psvo_bsimwr_forward_cov": (ctypes.c_int, [_DESC] + [_P] * 4 + [_MLP, _MLP, _MLP] + [_P] * 26 + [_P]),
    "psvo_bsimwr_backward_cov": (ctypes.c_int, [_DESC] + [_P] * 4 + [_MLP, _MLP, _MLP] + [_P] * 42 + [_P]),
    "psvo_mlp_wgrad_blocks": (ctypes.c_int, [ctypes.c_longlong]),
    "psvo_mlp_wgrad": (ctypes.c_int, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                      _P, _P, _MLP, _P, _P, ctypes.c_int, _P]),
    "psvo_mlp2_wgrad_blocks": (ctypes.c_int, [ctypes.c_longlong]),
    "psvo_mlp2_wgrad": (ctypes.c_int, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, ctypes.c_int,
                                       _P, _P, _MLP, _P, _P, ctypes.c_int, _P]),
    "psvo_bsim_forward": (ctypes.c_int, [_DESC] + [_P] * 4 + [_MLP, _MLP, _MLP] + [_P] * 22 + [_P]),
    "psvo_bsim_blocks": (ctypes.c_int, [_DESC]),
    "psvo_set_tuning": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "psvo_get_tuning": (ctypes.c_int, [ctypes.c_int]),
    "psvo_bsim_acc_size": (ctypes.c_int, [ctypes.c_int, ctypes.c_int]),
    "psvo_bsim_backward": (ctypes.c_int, [_DESC] + [_P] * 3 + [_MLP, _MLP, _MLP] + [_P] * 28),
    "psvo_bsim_backward_fold": (ctypes.c_int, [_DESC] + [_P] * 15),
    "psvo_bsimwr_blocks": (ctypes.c_int, [ctypes.c_int] * 3),
    "psvo_bsimwr_ws_floats": (ctypes.c_longlong, [ctypes.c_int] * 3),
    "psvo_bsimwr_forward": (ctypes.c_int, [_DESC] + [_P] * 3 + [_MLP, _MLP, _MLP] + [_P] * 25 + [_P]),
    "psvo_bsimwr_bwd_ws_floats": (ctypes.c_longlong, [ctypes.c_int] * 4),
    "psvo_bsimwr_backward": (ctypes.c_int, [_DESC] + [_P] * 3 + [_MLP, _MLP, _MLP] + [_P] * 39),
    "psvo_rows_mlp_blocks": (ctypes.c_int, [ctypes.c_longlong]),
    "psvo_rows_mlp_forward": (ctypes.c_int, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _MLP, _P, _P]),
    "psvo_rows_mlp_backward": (ctypes.c_int, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, ctypes.c_int, _P, _P, _MLP,
                                              _P, _P, _P, ctypes.c_int, _P]),
    "psvo_dense_wgrad_slices": (ctypes.c_int, [ctypes.c_longlong]),
    "psvo_dense_forward": (ctypes.c_int, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, _P, _P, _P, ctypes.c_int, _P, _P]),
    "psvo_dense_backward": (ctypes.c_int, [ctypes.c_longlong, ctypes.c_int, ctypes.c_int, _P, _P, _P, _P, ctypes.c_int,
                                           _P, _P, _P, ctypes.c_int, _P]),
    "psvo_bilstm_forward": (ctypes.c_int, [ctypes.c_int] * 4 + [_P] * 8 + [_P]),
    "psvo_bilstm_backward": (ctypes.c_int, [ctypes.c_int] * 4 + [_P] * 10 + [_P]),
    "psvo_bilstm_wgrad_fold": (ctypes.c_int, [ctypes.c_int] * 3 + [_P] * 4 + [ctypes.c_int, _P]),
    "psvo_adam_step": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_longlong, ctypes.c_float, ctypes.c_float, ctypes.c_float,
                                      ctypes.c_float, ctypes.c_longlong, ctypes.c_float, _P]),
    "psvo_reduce_rows": (ctypes.c_int, [_P, ctypes.c_int, ctypes.c_longlong, ctypes.c_int, _P, ctypes.c_int, _P]),
    "psvo_sigma_forward": (ctypes.c_int, [_P, _P, _P, ctypes.c_int, _P]),
    "psvo_sigma_backward": (ctypes.c_int, [_P, _P, _P, _P, ctypes.c_int, ctypes.c_int, _P]),
    "psvo_debug_stamp": (ctypes.c_int, [_P, _P]),
    "psvo_selftest_lanes": (ctypes.c_int, [_P, _P, _P]),
    "psvo_selftest_lanes2": (ctypes.c_int, [_P, _P, _P]),
    "psvo_elbo_filter": (ctypes.c_int, [_DESC, _P, _P, _P]),
    "psvo_elbo_bsim": (ctypes.c_int, [_DESC, _P, _P, _P]),
    "psvo_elbo_bsim_mean": (ctypes.c_int, [_DESC, _P, _P, _P]),
    "psvo_elbo_bsim_mean_backward": (ctypes.c_int, [_DESC, _P, _P, _P, _P]),
}

_lib = None


def load():
    """Load the library once; raise loudly when it has not been built."""
    global _lib
    if _lib is None:
        if not os.path.exists(LIB_PATH):
            raise PsvoHipError(
                "libpsvo_hip.so not found at %s -- build it with `python -m psvo_amd.build` "
                "(there is no CPU or PyTorch fallback for the PSVO hot path)" % LIB_PATH)
        # torch owns the device memory and the streams handed to the library, so both must share
        # ONE HIP runtime: import torch first so that its libamdhip64 is the one the loader binds.
        import torch  # noqa: F401
        lib = ctypes.CDLL(LIB_PATH)
        for name, (res, args) in SIGNATURES.items():
            fn = getattr(lib, name)
            fn.restype = res
            fn.argtypes = args
        v = os.environ.get("PSVO_BSIM_BWD_VARIANT")      # A/B measurements (tools/, bench.py --bsim-bwd-variant)
        if v is not None and lib.psvo_set_tuning(PSVO_TUNE_BSIM_BWD, int(v)) != PSVO_OK:
            raise PsvoHipError("PSVO_BSIM_BWD_VARIANT=%s is not a valid psvo_set_tuning value" % v)
        v = os.environ.get("PSVO_SKEW")                  # A/B: phase offset of co-resident workgroups, per cent (0 = off)
        if v is not None and lib.psvo_set_tuning(PSVO_TUNE_SKEW, int(v)) != PSVO_OK:
            raise PsvoHipError("PSVO_SKEW=%s is not a valid psvo_set_tuning value" % v)
        v = os.environ.get("PSVO_FILTER_BWD_SCAN")       # A/B: 0 = the persistent reverse filter kernel instead of the affine scan
        if v is not None and lib.psvo_set_tuning(PSVO_TUNE_FILTER_BWD, int(v)) != PSVO_OK:
            raise PsvoHipError("PSVO_FILTER_BWD_SCAN=%s is not a valid psvo_set_tuning value" % v)
        v = os.environ.get("PSVO_WGRAD2")                # A/B: two-layer weight gradients on the bf16 matrix instructions (2 / 3 pieces)
        if v is not None and lib.psvo_set_tuning(PSVO_TUNE_WGRAD2, int(v)) != PSVO_OK:
            raise PsvoHipError("PSVO_WGRAD2=%s is not a valid psvo_set_tuning value" % v)
        v = os.environ.get("PSVO_L2_SPLIT")              # A/B: two-layer backward-simulation kernels with the half-split chains
        if v is not None and lib.psvo_set_tuning(PSVO_TUNE_L2_SPLIT, int(v)) != PSVO_OK:
            raise PsvoHipError("PSVO_L2_SPLIT=%s is not a valid psvo_set_tuning value" % v)
        _lib = lib
    return _lib


def check(status, what):
    if status != PSVO_OK:
        msg = load().psvo_status_string(status).decode()
        if status == PSVO_ERR_HIP:
            msg += ": " + load().psvo_last_hip_error().decode()
        exc = ValueError if status in (PSVO_ERR_INVALID, PSVO_ERR_UNSUPPORTED) else PsvoHipError
        raise exc("%s failed: %s (status %d)" % (what, msg, status))
