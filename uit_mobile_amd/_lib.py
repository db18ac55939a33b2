"""ctypes binding of libuitk.so -- one function per entry point of include/uitk.h, nothing else."""
import ctypes
from ctypes import POINTER, Structure, c_char_p, c_float, c_int, c_int32, c_int64, c_size_t, c_void_p
from pathlib import Path

import os

HERE = Path(__file__).resolve().parent
# UITK_LIB_PATH: development only (A/B timing of two builds of this same library on one GPU box)
LIB_PATH = Path(os.environ["UITK_LIB_PATH"]) if os.environ.get("UITK_LIB_PATH") else HERE / "libuitk.so"

UITK_OK = 0
ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_NO_DEVICE, ERR_HIP, ERR_TOO_SHORT, ERR_WORKSPACE = -1, -2, -3, -4, -5, -6
IN_F32, IN_I16 = 0, 1
PREC_F16, PREC_BF16, PREC_FP8, PREC_F16X2, PREC_F16L6, PREC_FP8W, PREC_F16A, PREC_F16AX, PREC_F32 = 0, 1, 2, 3, 4, 5, 6, 7, 8
AVG_MEAN, AVG_MAX = 0, 1
AMAX_SLOTS = 64

EXPORTS = ["uitk_create", "uitk_destroy", "uitk_pack_weights", "uitk_num_frames", "uitk_num_crops",
           "uitk_reserve", "uitk_forward", "uitk_frontend", "uitk_amax_ptr", "uitk_encode", "uitk_mel_db",
           "uitk_set_debug_tokens", "uitk_set_timing_buffers", "uitk_set_option", "uitk_get_option", "uitk_topk", "uitk_kws_transform", "uitk_last_error",
           "uitk_version", "uitk_calibrate_weights", "uitk_calib_stats_doubles", "uitk_get_calib_stats", "uitk_pack_weights_calibrated",
           "uitk_tune_frontend", "uitk_get_calib_check", "uitk_self_check", "uitk_deviation_stats",
           "uitk_general_create", "uitk_general_destroy", "uitk_general_num_frames", "uitk_general_num_crops", "uitk_general_reserve",
           "uitk_general_frontend", "uitk_general_amax_ptr", "uitk_general_encode", "uitk_general_forward", "uitk_general_mel_db"]


class Config(Structure):
    _fields_ = [(n, c_int32) for n in ("depth", "embed_dim", "num_heads", "mlp_hidden", "outputdim",
                                       "target_length", "n_mels", "n_fft", "hop_size", "win_size", "patch_size",
                                       "center", "eval_avg", "precision")] + [("top_db", c_float), ("attention_full", c_int32),
                                                                                 ("act_gelu", c_int32), ("pooling", c_int32), ("fe_autotune", c_int32)]


_BLOCK_FIELDS = ["norm1_w", "norm1_b", "qkv_w", "qkv_b", "proj_w", "proj_b", "norm2_w", "norm2_b",
                 "fc1_w", "fc1_b", "fc2_w", "fc2_b"]
_GLOBAL_FIELDS = ["window", "mel_fb", "bn_weight", "bn_bias", "bn_mean", "bn_var", "pe_w", "pe_b", "time_pos",
                  "freq_pos", "norm_w", "norm_b", "head_ln_w", "head_ln_b", "head_w", "head_b"]


class BlockWeights(Structure):
    _fields_ = [(n, c_void_p) for n in _BLOCK_FIELDS]


class Weights(Structure):
    _fields_ = [(n, c_void_p) for n in _GLOBAL_FIELDS] + [("blocks", POINTER(BlockWeights)), ("cls_token", c_void_p),
                                                          ("token_pos", c_void_p)]


_lib = None


def load():
    """Load libuitk.so.  Fails loudly if it has not been built -- there is no fallback implementation."""
    global _lib
    if _lib is not None:
        return _lib
    if not LIB_PATH.exists():
        raise RuntimeError(f"{LIB_PATH} is missing: build it with `python -m uit_mobile_amd.build` "
                           "(hipcc, gfx950). uit_mobile_amd has no CPU or PyTorch fallback path.")
    L = ctypes.CDLL(str(LIB_PATH))
    L.uitk_create.argtypes = [POINTER(Config), POINTER(Weights), POINTER(c_void_p)]
    L.uitk_destroy.argtypes = [c_void_p]
    L.uitk_pack_weights.argtypes = [POINTER(Config), POINTER(Weights), c_void_p, c_size_t, POINTER(c_size_t)]
    L.uitk_num_frames.argtypes = [c_void_p, c_int64]
    L.uitk_num_crops.argtypes = [c_void_p, c_int64]
    L.uitk_reserve.argtypes = [c_void_p, c_int64, c_int64]
    L.uitk_forward.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
    L.uitk_frontend.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]
    L.uitk_amax_ptr.argtypes = [c_void_p]
    L.uitk_amax_ptr.restype = c_void_p
    L.uitk_encode.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
    L.uitk_mel_db.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]
    L.uitk_set_debug_tokens.argtypes = [c_void_p, c_void_p]
    L.uitk_set_timing_buffers.argtypes = [c_void_p, c_void_p, c_void_p]
    L.uitk_set_option.argtypes = [c_void_p, c_char_p, c_int]
    L.uitk_get_option.argtypes = [c_void_p, c_char_p, c_void_p]
    L.uitk_topk.argtypes = [c_void_p, c_int64, c_int32, c_int32, c_void_p, c_void_p, c_void_p]
    L.uitk_kws_transform.argtypes = [c_void_p, c_void_p, c_int64, c_int32, c_int32, c_float, c_void_p, c_void_p]
    L.uitk_last_error.restype = c_char_p
    L.uitk_version.restype = c_char_p
    L.uitk_calibrate_weights.argtypes = [c_void_p, POINTER(Weights), c_void_p, c_int, c_int64, c_int64, c_void_p]
    L.uitk_calib_stats_doubles.argtypes = [POINTER(Config)]
    L.uitk_calib_stats_doubles.restype = c_size_t
    L.uitk_get_calib_stats.argtypes = [c_void_p, c_void_p, c_size_t]
    L.uitk_get_calib_check.argtypes = [c_void_p, c_void_p]
    L.uitk_pack_weights_calibrated.argtypes = [POINTER(Config), POINTER(Weights), c_void_p, c_void_p, c_size_t, POINTER(c_size_t)]
    L.uitk_self_check.argtypes = [c_void_p, POINTER(Weights), c_void_p, c_void_p]
    L.uitk_tune_frontend.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]
    L.uitk_deviation_stats.argtypes = [c_void_p, c_void_p, c_size_t, c_void_p]
    L.uitk_general_create.argtypes = [POINTER(Config), POINTER(Weights), POINTER(c_void_p)]
    L.uitk_general_destroy.argtypes = [c_void_p]
    L.uitk_general_num_frames.argtypes = [c_void_p, c_int64]
    L.uitk_general_num_crops.argtypes = [c_void_p, c_int64]
    L.uitk_general_reserve.argtypes = [c_void_p, c_int64, c_int64]
    L.uitk_general_frontend.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p]
    L.uitk_general_amax_ptr.argtypes = [c_void_p]
    L.uitk_general_amax_ptr.restype = c_void_p
    L.uitk_general_encode.argtypes = [c_void_p, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
    L.uitk_general_forward.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p, c_void_p]
    L.uitk_general_mel_db.argtypes = [c_void_p, c_void_p, c_int, c_int64, c_int64, c_void_p, c_void_p]
    for name in EXPORTS:                      # every entry point include/uitk.h declares must be exported
        getattr(L, name)
    _lib = L
    return L


def last_error():
    return load().uitk_last_error().decode()


def check(rc, what):
    """Map C return codes onto the exception types the reference's callers see (Errors row)."""
    if rc == UITK_OK:
        return
    msg = f"{what}: {last_error()} (uitk rc={rc})"
    if rc in (ERR_INVALID_ARG, ERR_UNSUPPORTED, ERR_TOO_SHORT):
        raise ValueError(msg)
    raise RuntimeError(msg)
