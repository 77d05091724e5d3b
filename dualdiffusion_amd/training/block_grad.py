"""Forward-with-tape and backward of one EDM2 decoder-style block on the HIP kernels (training building block).

Mirrors the non-attention path of reference src/modules/unets/unet_edm2_b4.py:110-135,153-158 under autograd:
    x   = mp_cat(src0, src1)                      (scales s0, s1; src1 optional)
    y0  = conv_res0(mp_silu(x))
    y1  = conv_res1(mp_silu(y0 * c))              c = emb_linear(emb) * gain + 1, given here as a [B, Cmid] tensor
    out = clip(mp_sum(conv_skip(x) | x, y1, t))
with forced weight normalisation inside the forward (module.training, mp_tools.py:360-361).  The training forward keeps the
RAW tensors (x sources, y0, out); the backward recomputes the activated conv operands (HBM-bound) and runs
    mp_sum/clip backward -> conv_res1 wgrad + dgrad -> mp_silu(y0*c) backward (dy0, dc) -> conv_res0 wgrad + dgrad ->
    conv_skip wgrad + dgrad -> mp_silu(x) backward per source (+ skip gradient) -> weight-path backward.
This is host orchestration over `dualdiffusion_amd.ops`; it is the unit the whole-UNet backward plan will be made of.
"""
from __future__ import annotations

from dataclasses import dataclass
from typing import Optional

import torch

from .. import ops
from .._lib import PRO_SCALE_SILU, PRO_SILU


@dataclass
class BlockTape:
    src0: torch.Tensor
    src1: Optional[torch.Tensor]
    s0: float
    s1: float
    c: torch.Tensor
    y0: torch.Tensor
    out: torch.Tensor
    groups: int
    res_t: float
    clip: float
    pw_res0: ops.PreparedWeight
    pw_res1: ops.PreparedWeight
    pw_skip: Optional[ops.PreparedWeight]
    w_res0: torch.Tensor
    w_res1: torch.Tensor
    w_skip: Optional[torch.Tensor]


def block_forward_train(src0: torch.Tensor, src1: Optional[torch.Tensor], s0: float, s1: float, c: torch.Tensor, w_res0: torch.Tensor,
                        w_res1: torch.Tensor, w_skip: Optional[torch.Tensor], groups: int, res_t: float = 0.3, clip: float = 256.0):
    """NHWC activations (bf16), fp32 master weights.  Returns (out, tape)."""
    dt = src0.dtype
    C0 = src0.shape[-1]
    pw0 = ops.wprep(w_res0, groups, dt, normalize=True)
    pw1 = ops.wprep(w_res1, groups, dt, normalize=True)
    y0 = ops.conv2d(src0, pw0, src1=src1, scale0=s0, scale1=s1, prologue=PRO_SILU)
    if w_skip is not None:
        pws = ops.wprep(w_skip, 1, dt, normalize=True, in_split=C0 if src1 is not None else 0, in_scale0=s0, in_scale1=s1)
        sk = ops.conv2d(src0, pws, src1=src1)
    else:
        assert src1 is None and s0 == 1.0
        pws, sk = None, src0
    out = ops.conv2d(y0, pw1, prologue=PRO_SCALE_SILU, chan_scale=c, residual=sk, res_t=res_t, clip=clip)
    return out, BlockTape(src0, src1, s0, s1, c, y0, out, groups, res_t, clip, pw0, pw1, pws, w_res0, w_res1, w_skip)


def block_backward(tape: BlockTape, dout: torch.Tensor):
    """Gradients of one block: returns dict(dsrc0, dsrc1, dc, dw_res0, dw_res1, dw_skip) (weights fp32, activations NHWC)."""
    t = tape
    dt = t.src0.dtype
    C0 = t.src0.shape[-1]
    C1 = t.src1.shape[-1] if t.src1 is not None else 0
    G = t.groups
    # out = clip(mp_sum(sk, y1, t))
    dsk, dy1 = ops.mpsum_clip_bwd(dout, t.out, t.res_t, t.clip)
    # y1 = conv_res1(a1), a1 = mp_silu(y0 * c)
    a1 = ops.silu_scale_fwd(t.y0, t.c)
    dwp1 = ops.conv2d_wgrad(dy1, a1, G, t.pw_res1.ksize)
    da1 = ops.conv2d(dy1, ops.wprep(t.w_res1, G, dt, normalize=True, transpose=True))
    dc = torch.zeros_like(t.c)
    dy0 = ops.silu_scale_bwd(da1, t.y0, t.c, 1.0, dc)
    # y0 = conv_res0(a0), a0 = mp_silu(cat(s0 * src0, s1 * src1))
    a00 = ops.silu_scale_fwd(t.src0, None, t.s0)
    a01 = ops.silu_scale_fwd(t.src1, None, t.s1) if t.src1 is not None else None
    dwp0 = ops.conv2d_wgrad(dy0, a00, G, t.pw_res0.ksize, x1=a01)
    da0 = ops.conv2d(dy0, ops.wprep(t.w_res0, G, dt, normalize=True, transpose=True))
    # skip path: sk = conv_skip(cat) with the cat scales folded into the weights, or the input itself
    if t.w_skip is not None:
        dwps = ops.conv2d_wgrad(dsk, t.src0, 1, t.pw_skip.ksize, x1=t.src1)
        dxs = ops.conv2d(dsk, ops.wprep(t.w_skip, 1, dt, normalize=True, transpose=True, in_split=C0 if C1 else 0, in_scale0=t.s0, in_scale1=t.s1))
        dw_skip = ops.wprep_bwd(t.pw_skip, dwps)
    else:
        dxs, dw_skip = dsk, None
    dsrc0 = ops.silu_scale_bwd(da0[..., :C0], t.src0, None, t.s0, add=dxs[..., :C0])
    dsrc1 = ops.silu_scale_bwd(da0[..., C0:], t.src1, None, t.s1, add=dxs[..., C0:]) if C1 else None
    return dict(dsrc0=dsrc0, dsrc1=dsrc1, dc=dc, dw_res0=ops.wprep_bwd(t.pw_res0, dwp0), dw_res1=ops.wprep_bwd(t.pw_res1, dwp1), dw_skip=dw_skip)
