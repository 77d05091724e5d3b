"""Backward of the block's self-attention on the HIP kernels (training building block).

Forward being differentiated (reference src/modules/unets/unet_edm2_b4.py:137-148): q, k, v normalised over the head dim per
token, `scaled_dot_product_attention`.  With a few hundred tokens per image the T x T matrices are small, so the backward is
five batched MFMA GEMMs over materialised P / dS (csrc/bgemm.hip) between row-softmax kernels and the pixel-norm kernels
applied to rows of head_dim elements:
    Qn, Kn, Vn = normalize rows          S = Qn Kn^T            P = softmax(S / sqrt(d))
    dP = dO Vn^T                         dS = P o (dP - rowsum(P o dP)) / sqrt(d)
    dVn = P^T dO      dKn = dS^T Qn      dQn = dS Kn            dq, dk, dv = normalize backward
Layouts as in the forward kernel: qk `[B, T, 2C]` channels (head, {q,k}, d), v / dO `[B, T, C]` channels (head, d); bf16.
"""
from __future__ import annotations

import ctypes as C
import math

import torch

from .. import ops
from .. import _lib as L
from .._lib import check, current_stream, lib, ptr


def _gemm(A, a_off, lda, sA0, sA1, a_kmajor, Bm, b_off, ldb, sB0, sB1, b_kmajor, Cm, c_off, ldc, sC0, sC1, M, N, K, nb0, nb1, alpha=1.0):
    es_a, es_c = A.element_size(), Cm.element_size()
    d = L.BgemmDesc(A=A.data_ptr() + a_off * es_a, B=Bm.data_ptr() + b_off * Bm.element_size(), C=Cm.data_ptr() + c_off * es_c,
                    lda=lda, ldb=ldb, ldc=ldc, sA0=sA0, sA1=sA1, sB0=sB0, sB1=sB1, sC0=sC0, sC1=sC1, M=M, N=N, K=K, nb0=nb0, nb1=nb1,
                    a_kmajor=int(a_kmajor), b_kmajor=int(b_kmajor), c_fp32=int(Cm.dtype == torch.float32), alpha=float(alpha))
    fn = lib().ddx_bgemm_f32 if A.dtype == torch.float32 else lib().ddx_bgemm_bf16
    check(fn(C.byref(d), current_stream()), "bgemm")


def attention_backward(qk: torch.Tensor, v: torch.Tensor, do: torch.Tensor, heads: int, eps: float = 1e-4):
    """qk [B, H, W, 2C] / v, do [B, H, W, C] (NHWC bf16; do = gradient w.r.t. the attention output) -> (dqk, dv)."""
    dt = qk.dtype
    assert dt in (torch.bfloat16, torch.float32) and v.dtype == dt and do.dtype == dt      # fp32: scalar parity kernels (ddx_bgemm_f32)
    softmax, softmax_bwd = (lib().ddx_softmax_rows_f32, lib().ddx_softmax_bwd_rows_f32) if dt == torch.float32 else \
        (lib().ddx_softmax_rows, lib().ddx_softmax_bwd_rows)
    B, Cn = v.shape[0], v.shape[-1]
    T = v.numel() // (B * Cn)
    d = Cn // heads
    assert d % 8 == 0
    Tp = (T + 7) // 8 * 8
    dev = v.device
    scale = 1.0 / math.sqrt(d)
    qk_n = ops.pixelnorm(qk.reshape(-1, d), eps=eps).reshape(B, T, 2 * Cn)
    v_n = ops.pixelnorm(v.reshape(-1, d), eps=eps).reshape(B, T, Cn)
    do2 = do.reshape(B, T, Cn)
    S = torch.empty(B, heads, T, Tp, dtype=torch.float32, device=dev)
    P = torch.zeros(B, heads, T, Tp, dtype=dt, device=dev)       # pad columns stay zero
    dS = torch.zeros(B, heads, T, Tp, dtype=dt, device=dev)
    sq0, sq1 = T * 2 * Cn, 2 * d
    sv0, sv1 = T * Cn, d
    sp0, sp1 = heads * T * Tp, T * Tp
    # S = Qn Kn^T
    _gemm(qk_n, 0, 2 * Cn, sq0, sq1, False, qk_n, d, 2 * Cn, sq0, sq1, False, S, 0, Tp, sp0, sp1, T, T, d, B, heads)
    check(softmax(ptr(S), ptr(P), B * heads * T, T, Tp, scale, current_stream()), "softmax_rows")
    # dP = dO Vn^T  (into S's buffer)
    _gemm(do2, 0, Cn, sv0, sv1, False, v_n, 0, Cn, sv0, sv1, False, S, 0, Tp, sp0, sp1, T, T, d, B, heads)
    check(softmax_bwd(ptr(P), ptr(S), ptr(dS), B * heads * T, T, Tp, scale, current_stream()), "softmax_bwd_rows")
    dqk_n = torch.empty_like(qk_n)
    dv_n = torch.empty_like(v_n)
    # dVn = P^T dO ; dKn = dS^T Qn ; dQn = dS Kn
    _gemm(P, 0, Tp, sp0, sp1, True, do2, 0, Cn, sv0, sv1, True, dv_n, 0, Cn, sv0, sv1, T, d, T, B, heads)
    _gemm(dS, 0, Tp, sp0, sp1, True, qk_n, 0, 2 * Cn, sq0, sq1, True, dqk_n, d, 2 * Cn, sq0, sq1, T, d, T, B, heads)
    _gemm(dS, 0, Tp, sp0, sp1, False, qk_n, d, 2 * Cn, sq0, sq1, True, dqk_n, 0, 2 * Cn, sq0, sq1, T, d, T, B, heads)
    dqk = ops.pixelnorm_bwd(dqk_n.reshape(-1, d), qk.reshape(-1, d), eps).reshape(qk.shape)
    dv = ops.pixelnorm_bwd(dv_n.reshape(-1, d), v.reshape(-1, d), eps).reshape(v.shape)
    return dqk, dv
