"""ctypes binding of libddx_hip.so (the C ABI declared in include/ddx_hip.h).

The product path has no CPU or PyTorch fallback: if the HIP library is missing or a call fails this module
raises.  PyTorch is used only for device memory, streams and (later) torch.distributed.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

_HERE = os.path.dirname(os.path.abspath(__file__))
LIB_PATH = os.path.join(_HERE, "lib", "libddx_hip.so")

DDX_F32, DDX_BF16 = 0, 1
RESAMPLE_KEEP, RESAMPLE_UP, RESAMPLE_DOWN = 0, 1, 2
RESAMPLE_UP_BWD, RESAMPLE_DOWN_BWD = 3, 4
PRO_NONE, PRO_SILU, PRO_SCALE, PRO_SCALE_SILU = 0, 1, 2, 3
PAD_ZERO, PAD_REFLECT_W, PAD_SWAP_SRC1, PAD_SWAP_PAIRED = 0, 1, 2, 4
LAYOUT_SRC0_C16, LAYOUT_SRC1_C16, LAYOUT_OUT_C16, LAYOUT_OUT2_C16 = 1, 2, 4, 8
EPI_STORE, EPI_MPSUM, EPI_PIXELNORM = 0, 1, 3


class DDXError(RuntimeError):
    pass


# Parameters are also written through `.data` and raw device pointers (fused AdamW, forced weight normalisation of the weight
# bank), which torch's per-tensor `_version` counters do not see.  Every such writer bumps this epoch; the prepared-weight caches
# of the inference engines key on it (together with the `_version`s).
_weights_epoch = 0


def bump_weights_epoch() -> None:
    global _weights_epoch
    _weights_epoch += 1


def weights_epoch() -> int:
    return _weights_epoch


class WPrepDesc(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p), ("gain_ptr", C.c_void_p), ("gain", C.c_float),
                ("w_dtype", C.c_int32), ("wp_dtype", C.c_int32), ("Cout", C.c_int32), ("Cg", C.c_int32),
                ("ksize", C.c_int32), ("groups", C.c_int32), ("CK", C.c_int32), ("normalize", C.c_int32),
                ("qk_head_dim", C.c_int32), ("in_split", C.c_int32), ("in_scale0", C.c_float), ("in_scale1", C.c_float),
                ("transpose", C.c_int32), ("row_scale", C.c_void_p), ("row_offset", C.c_int32), ("rows_total", C.c_int32)]


class ConvDesc(C.Structure):
    _fields_ = [("src0", C.c_void_p), ("src1", C.c_void_p), ("chan_scale", C.c_void_p), ("wp", C.c_void_p),
                ("residual", C.c_void_p), ("out", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32),
                ("C0", C.c_int32), ("C1", C.c_int32), ("Cout", C.c_int32), ("groups", C.c_int32), ("ksize", C.c_int32),
                ("CK", C.c_int32), ("resample", C.c_int32), ("prologue", C.c_int32), ("epilogue", C.c_int32),
                ("scale0", C.c_float), ("scale1", C.c_float), ("res_t", C.c_float), ("clip", C.c_float),
                ("dtype", C.c_int32), ("force_direct", C.c_int32),
                ("out_scale", C.c_void_p), ("out2", C.c_void_p), ("out_act", C.c_int32), ("out2_scale", C.c_float),
                ("pad_mode", C.c_int32), ("prologue_rows", C.c_int32),
                ("out2_linear", C.c_int32), ("layout", C.c_int32), ("out2_chan_scale", C.c_void_p), ("src0_alt", C.c_void_p),
                ("residual_up", C.c_int32), ("out_head_norm", C.c_int32), ("out_head_eps", C.c_float)]


class DgradActDesc(C.Structure):
    _fields_ = [("conv", ConvDesc), ("y0", C.c_void_p), ("y1", C.c_void_p), ("out1", C.c_void_p), ("add", C.c_void_p),
                ("chan_scale", C.c_void_p), ("dchan_scale", C.c_void_p), ("workspace", C.c_void_p),
                ("split", C.c_int32), ("act", C.c_int32), ("scale0", C.c_float), ("scale1", C.c_float)]


class ConvPairDesc(C.Structure):
    _fields_ = [("src", C.c_void_p), ("wp0", C.c_void_p), ("wp1", C.c_void_p), ("chan_scale", C.c_void_p), ("residual", C.c_void_p),
                ("out", C.c_void_p), ("out2", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C", C.c_int32), ("hidden", C.c_int32), ("groups", C.c_int32),
                ("CK0", C.c_int32), ("CK1", C.c_int32), ("dtype", C.c_int32),
                ("res_t", C.c_float), ("clip", C.c_float), ("out2_scale", C.c_float)]


class MelStftDesc(C.Structure):
    _fields_ = [("audio", C.c_void_p), ("window", C.c_void_p), ("twiddle", C.c_void_p), ("band_start", C.c_void_p),
                ("band_len", C.c_void_p), ("band_w", C.c_void_p), ("out", C.c_void_p),
                ("B", C.c_int32), ("C", C.c_int32), ("L", C.c_int32), ("T", C.c_int32), ("n_fft", C.c_int32), ("hop", C.c_int32),
                ("n_mel", C.c_int32), ("band_stride", C.c_int32), ("exponent", C.c_float), ("mean", C.c_float), ("scale", C.c_float)]


class MsMelDesc(C.Structure):
    _fields_ = [("audio", C.c_void_p), ("window_low", C.c_void_p), ("window_high", C.c_void_p), ("twiddle", C.c_void_p),
                ("bin_scale_low", C.c_void_p), ("bin_scale_high", C.c_void_p), ("band_start", C.c_void_p), ("band_len", C.c_void_p),
                ("band_w", C.c_void_p), ("out", C.c_void_p),
                ("B", C.c_int32), ("C", C.c_int32), ("L", C.c_int32), ("T", C.c_int32), ("n_fft", C.c_int32), ("hop", C.c_int32),
                ("n_mel", C.c_int32), ("band_stride", C.c_int32), ("exponent", C.c_float), ("scale", C.c_float), ("offset", C.c_float)]


class WgradDesc(C.Structure):
    _fields_ = [("dy", C.c_void_p), ("x0", C.c_void_p), ("x1", C.c_void_p), ("dw", C.c_void_p), ("workspace", C.c_void_p),
                ("B", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("C0", C.c_int32), ("C1", C.c_int32), ("Cout", C.c_int32),
                ("groups", C.c_int32), ("ksize", C.c_int32), ("resample", C.c_int32), ("dtype", C.c_int32), ("accumulate", C.c_int32)]


class BgemmDesc(C.Structure):
    _fields_ = [("A", C.c_void_p), ("B", C.c_void_p), ("C", C.c_void_p), ("lda", C.c_int64), ("ldb", C.c_int64), ("ldc", C.c_int64),
                ("sA0", C.c_int64), ("sA1", C.c_int64), ("sB0", C.c_int64), ("sB1", C.c_int64), ("sC0", C.c_int64), ("sC1", C.c_int64),
                ("M", C.c_int32), ("N", C.c_int32), ("K", C.c_int32), ("nb0", C.c_int32), ("nb1", C.c_int32),
                ("a_kmajor", C.c_int32), ("b_kmajor", C.c_int32), ("c_fp32", C.c_int32), ("alpha", C.c_float)]


class OptimJob(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p), ("n", C.c_int64)]


MAX_EMAS = 4


class OptimJobEx(C.Structure):
    _fields_ = [("p", C.c_void_p), ("g", C.c_void_p), ("m", C.c_void_p), ("v", C.c_void_p), ("ema", C.c_void_p * MAX_EMAS),
                ("n", C.c_int64), ("rows", C.c_int64), ("normalize", C.c_int32), ("reserved", C.c_int32)]


class LinearBwdJob(C.Structure):
    _fields_ = [("dc", C.c_void_p), ("w", C.c_void_p), ("row_scale", C.c_void_p), ("dwp", C.c_void_p), ("O", C.c_int32), ("groups", C.c_int32)]


class WPathJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("wp", C.c_void_p), ("wp_t", C.c_void_p), ("row_scale", C.c_void_p), ("gain_ptr", C.c_void_p),
                ("dwp", C.c_void_p), ("dw", C.c_void_p), ("dgain", C.c_void_p), ("gain", C.c_float),
                ("Cout", C.c_int32), ("Cg", C.c_int32), ("ksize", C.c_int32), ("groups", C.c_int32), ("CK", C.c_int32), ("CK_t", C.c_int32),
                ("normalize", C.c_int32), ("qk_head_dim", C.c_int32), ("in_split", C.c_int32), ("in_scale0", C.c_float), ("in_scale1", C.c_float),
                ("dwp_parts", C.c_int32), ("reserved", C.c_int32)]


WPATH_NORMALIZE, WPATH_PREP, WPATH_ROWSCALE, WPATH_TRANSPOSED, WPATH_BWD = range(5)


class MssDesc(C.Structure):
    _fields_ = [("sample", C.c_void_p), ("target", C.c_void_p), ("window", C.c_void_p), ("weight", C.c_void_p),
                ("twiddle", C.c_void_p), ("loss", C.c_void_p), ("grad", C.c_void_p),
                ("B", C.c_int32), ("C", C.c_int32), ("H", C.c_int32), ("W", C.c_int32), ("block_width", C.c_int32),
                ("step", C.c_int32), ("midside", C.c_int32), ("use_mse", C.c_int32), ("loss_scale", C.c_float),
                ("phase_scale", C.c_float), ("weight_ld", C.c_int32), ("reserved", C.c_int32), ("stats", C.c_void_p)]


class LinearJob(C.Structure):
    _fields_ = [("w", C.c_void_p), ("gain_ptr", C.c_void_p), ("out", C.c_void_p), ("gain", C.c_float),
                ("add_const", C.c_float), ("O", C.c_int32), ("K", C.c_int32), ("groups", C.c_int32),
                ("normalize", C.c_int32)]


_lib: Optional[C.CDLL] = None

# name -> (restype, argtypes); kept in one table so tests can check every header symbol is exported
# ctypes mirrors in the order of ddx_abi_sizeof() (include/ddx_hip.h); tests/test_abi.py compares sizes and tail offsets with the library
ABI_MIRRORS = [WPrepDesc, ConvDesc, DgradActDesc, WgradDesc, LinearBwdJob, WPathJob, LinearJob, MelStftDesc, MsMelDesc, BgemmDesc, MssDesc,
               OptimJob, OptimJobEx, ConvPairDesc]

PROTOTYPES = {
    "ddx_version": (C.c_char_p, []),
    "ddx_last_error": (C.c_char_p, []),
    "ddx_abi_sizeof": (C.c_int64, [C.c_int32]),
    "ddx_abi_offsetof_tail": (C.c_int64, [C.c_int32]),
    "ddx_wprep_bytes": (C.c_size_t, [C.c_int32] * 6),
    "ddx_mpconv_wprep": (C.c_int, [C.POINTER(WPrepDesc), C.c_void_p]),
    "ddx_normalize_weights": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_int64, C.c_void_p]),
    "ddx_mpconv2d_fwd": (C.c_int, [C.POINTER(ConvDesc), C.c_void_p]),
    "ddx_mpconv2d_path": (C.c_int, [C.POINTER(ConvDesc)]),
    "ddx_mpconv2d_pick_ck": (C.c_int32, [C.c_int32, C.c_int32, C.c_int32, C.c_int64]),
    "ddx_pixelnorm_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_pixelnorm_act_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_attn_act_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                   C.c_int32, C.c_void_p]),
    "ddx_attn_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                               C.c_int32, C.c_void_p]),
    "ddx_linear_small_batched": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                           C.c_void_p]),
    "ddx_mpfourier": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_mpsum_rows": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_void_p]),
    "ddx_unet_input_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_unet_output_combine": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32,
                                          C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_mel_stft": (C.c_int, [C.POINTER(MelStftDesc), C.c_void_p]),
    "ddx_ms_mel_spec": (C.c_int, [C.POINTER(MsMelDesc), C.c_void_p]),
    "ddx_mel_to_amplitude": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_float, C.c_void_p]),
    "ddx_fgla_synth": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                 C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_fgla_ola": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_fgla_analysis": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_float, C.c_void_p]),
    "ddx_fgla_iter": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32,
                                C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_wgrad_workspace_bytes": (C.c_size_t, [C.POINTER(WgradDesc)]),
    "ddx_mpconv2d_wgrad": (C.c_int, [C.POINTER(WgradDesc), C.c_void_p]),
    "ddx_wgrad_parts": (C.c_int32, [C.POINTER(WgradDesc)]),
    "ddx_wgrad_parts_max": (C.c_int32, [C.c_int32] * 4),
    "ddx_silu_scale_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_int32,
                                     C.c_int32, C.c_void_p]),
    "ddx_silu_scale_bwd_ex": (C.c_int, [C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p,
                                        C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_silu_scale_fwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_int32, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_add3": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_void_p]),
    "ddx_mpsum_clip_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_float, C.c_int64, C.c_int32, C.c_void_p]),
    "ddx_pixelnorm_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_mpconv_wprep_bwd": (C.c_int, [C.POINTER(WPrepDesc), C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p]),
    "ddx_wprep_rowscale": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_float, C.c_int64, C.c_int64, C.c_int32, C.c_void_p]),
    "ddx_linear_small_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                       C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_bgemm_bf16": (C.c_int, [C.POINTER(BgemmDesc), C.c_void_p]),
    "ddx_bgemm_f32": (C.c_int, [C.POINTER(BgemmDesc), C.c_void_p]),
    "ddx_softmax_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    "ddx_softmax_bwd_rows_f32": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    "ddx_softmax_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    "ddx_softmax_bwd_rows": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int64, C.c_float, C.c_void_p]),
    "ddx_edm2_loss": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32,
                                C.c_int64, C.c_void_p]),
    "ddx_edm2_loss_v": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p,
                                  C.c_int32, C.c_int64, C.c_void_p]),
    "ddx_mp_dropout": (C.c_int, [C.c_void_p, C.c_int64, C.c_float, C.c_uint64, C.c_uint32, C.c_int32, C.c_void_p]),
    "ddx_unet_xref_mix_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_multi_grad_norm": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_float, C.c_void_p, C.c_void_p]),
    "ddx_clip_coef": (C.c_int, [C.c_void_p, C.c_float, C.c_float, C.c_void_p]),
    "ddx_multi_adamw": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                  C.c_int32, C.c_float, C.c_void_p]),
    "ddx_multi_adamw_ema_wn": (C.c_int, [C.c_void_p, C.c_int32, C.c_int64, C.c_void_p, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float, C.c_float,
                                         C.c_int32, C.c_int32, C.POINTER(C.c_float), C.POINTER(C.c_float), C.c_float, C.c_void_p]),
    "ddx_ddec_input_prep": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_float, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_cat2_act": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_cat2_swap": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_void_p, C.c_int64, C.c_int64, C.c_int32, C.c_int32,
                                C.c_int32, C.c_void_p]),
    "ddx_ddec_output_combine": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int64, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_unet_output_combine_bwd": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_float,
                                              C.c_int32, C.c_void_p]),
    "ddx_mss_loss_scale": (C.c_int, [C.POINTER(MssDesc), C.c_void_p]),
    "ddx_resample2d": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_lincomb3": (C.c_int, [C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_float, C.c_void_p, C.c_int64, C.c_void_p]),
    "ddx_nchw_to_nhwc": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_nhwc_to_nchw": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_linear_small_bwd_batched": (C.c_int, [C.c_void_p, C.c_int32, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_mpconv2d_dgrad_act_workspace_bytes": (C.c_size_t, [C.c_void_p]),
    "ddx_mpconv2d_dgrad_act": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddx_mpconv_pair_supported": (C.c_int, [C.c_int32] * 5),
    "ddx_mpconv_pair_fwd": (C.c_int, [C.POINTER(ConvPairDesc), C.c_void_p]),
    "ddx_attn_act_fwd_ld": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                      C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_attn_fold_fwd": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32,
                                    C.c_int32, C.c_int32, C.c_float, C.c_int32, C.c_void_p]),
    "ddx_wpath_multi": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_stereo_to_images": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_images_to_stereo": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_nhwc_to_nchw_ld": (C.c_int, [C.c_void_p, C.c_int32, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_int32, C.c_void_p]),
    "ddx_plan_begin": (C.c_void_p, []),
    "ddx_plan_end": (C.c_int, [C.c_void_p]),
    "ddx_plan_fork": (C.c_int, []),
    "ddx_plan_main": (C.c_int, []),
    "ddx_plan_join": (C.c_int, []),
    "ddx_plan_num_ops": (C.c_int, [C.c_void_p]),
    "ddx_plan_run": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddx_plan_graph_build": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddx_plan_graph_launch": (C.c_int, [C.c_void_p, C.c_void_p]),
    "ddx_plan_op_info": (C.c_int, [C.c_void_p, C.c_int, C.POINTER(C.c_char_p), C.POINTER(C.c_double), C.POINTER(C.c_double)]),
    "ddx_plan_profile": (C.c_int, [C.c_void_p, C.c_void_p, C.c_int, C.POINTER(C.c_float)]),
    "ddx_plan_destroy": (None, [C.c_void_p]),
    "ddx_plan_include": (C.c_int, [C.c_void_p]),
    "ddx_sampler_load": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int64, C.c_void_p]),
    "ddx_lincomb3_dev": (C.c_int, [C.c_void_p, C.c_void_p, C.c_void_p, C.c_void_p, C.c_int64, C.c_void_p, C.c_void_p, C.c_int32, C.c_int32, C.c_int32, C.c_int32,
                                   C.c_int64, C.c_void_p]),
    "ddx_step_advance": (C.c_int, [C.c_void_p, C.c_void_p]),
}


def lib() -> C.CDLL:
    """Load libddx_hip.so (once).  Raises DDXError when the library has not been built."""
    global _lib
    if _lib is None:
        if not os.path.isfile(LIB_PATH):
            raise DDXError(f"{LIB_PATH} not found: build it with `make` (or __graft_entry__.build()); "
                           "dualdiffusion_amd has no CPU fallback")
        handle = C.CDLL(LIB_PATH)
        for name, (res, args) in PROTOTYPES.items():
            fn = getattr(handle, name)  # AttributeError if the library does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = handle
    return _lib


def check(rc: int, what: str = "") -> None:
    if rc != 0:
        msg = lib().ddx_last_error().decode()
        raise DDXError(f"libddx_hip call failed ({what}): rc={rc} {msg}")


def dtype_code(dt: torch.dtype) -> int:
    if dt == torch.float32:
        return DDX_F32
    if dt == torch.bfloat16:
        return DDX_BF16
    raise DDXError(f"unsupported dtype {dt}: the HIP path computes in float32 or bfloat16")


def ptr(t: Optional[torch.Tensor]) -> Optional[int]:
    if t is None:
        return None
    if not t.is_cuda:
        raise DDXError("tensor is not on a ROCm device: dualdiffusion_amd has no CPU path")
    return t.data_ptr()


def current_stream() -> int:
    return torch.cuda.current_stream().cuda_stream


class Plan:
    """Recorded launch sequence (ddx_plan_*): record with `with plan.record():`, then run() / graph_launch()."""

    def __init__(self) -> None:
        self._h = None
        self._graph = False
        self.keepalive: list = []  # descriptors / tensors referenced by the recorded closures

    def record(self):
        plan = self

        class _Ctx:
            def __enter__(self_inner):
                h = lib().ddx_plan_begin()
                if not h:
                    raise DDXError("ddx_plan_begin failed: " + lib().ddx_last_error().decode())
                plan._h = h
                plan._recording = True
                return plan

            def __exit__(self_inner, et, ev, tb):
                plan._recording = False
                check(lib().ddx_plan_end(plan._h), "plan_end")
                return False

        return _Ctx()

    @property
    def num_ops(self) -> int:
        return lib().ddx_plan_num_ops(self._h) if self._h else 0

    def run(self, stream: Optional[int] = None) -> None:
        check(lib().ddx_plan_run(self._h, stream if stream is not None else current_stream()), "plan_run")

    def graph_build(self, stream: Optional[int] = None) -> None:
        check(lib().ddx_plan_graph_build(self._h, stream if stream is not None else current_stream()), "plan_graph_build")
        self._graph = True

    def graph_launch(self, stream: Optional[int] = None) -> None:
        check(lib().ddx_plan_graph_launch(self._h, stream if stream is not None else current_stream()), "plan_graph_launch")

    def include(self, other: "Plan") -> None:
        """While THIS plan is being recorded: append the launches of the finished plan `other`."""
        if not getattr(self, "_recording", False):     # (ddx_plan_include appends to whichever plan is recording: it must be this one)
            raise DDXError("Plan.include: this plan is not the one being recorded")
        check(lib().ddx_plan_include(other._h), "plan_include")
        self.keepalive.append(other)

    def profile(self, reps: int = 3, stream: Optional[int] = None) -> list:
        """Eager replay with a hipEvent pair around every op: [(tag, flops, bytes, mean_ms)] per op."""
        n = self.num_ops
        ms = (C.c_float * n)()
        check(lib().ddx_plan_profile(self._h, stream if stream is not None else current_stream(), reps, ms), "plan_profile")
        out = []
        for i in range(n):
            tag, fl, by = C.c_char_p(), C.c_double(), C.c_double()
            check(lib().ddx_plan_op_info(self._h, i, C.byref(tag), C.byref(fl), C.byref(by)), "plan_op_info")
            out.append((tag.value.decode(), fl.value, by.value, float(ms[i])))
        return out

    @property
    def has_graph(self) -> bool:
        return self._graph

    def __del__(self):
        try:
            if self._h and _lib is not None:
                _lib.ddx_plan_destroy(self._h)
        except Exception:
            pass
