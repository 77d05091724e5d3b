"""Forward-with-tape and backward of one EDM2 block (without attention) on the HIP kernels — the unit the whole-UNet
backward plan is made of.

Mirrors reference src/modules/unets/unet_edm2_b4.py:110-135,153-158 under autograd, forced weight normalisation inside the
forward (module.training, mp_tools.py:360-361):
    x   = resample(mp_cat(src0, src1))                              (scales s0, s1; src1 optional)
    enc: x = normalize(conv_skip(x) | x, dim=channels)
    y0  = conv_res0(mp_silu(x));   c = emb_linear(emb) * emb_gain + 1
    y1  = conv_res1(mp_silu(y0 * c))
    dec: x = conv_skip(x) | x
    out = clip(mp_sum(x, y1, t))
The training forward keeps the RAW tensors (block input, pre-norm skip output, y0, out) AND the activated conv operands (bf16
twins written by the producers' epilogues, see block_forward_train); the backward chains
    mp_sum/clip backward -> conv_res1 wgrad + dgrad -> mp_silu(y0*c) backward (dy0, dc) -> emb_linear backward ->
    conv_res0 wgrad + dgrad -> mp_silu backward per source (+ residual / skip gradient) -> [pixel-norm backward ->]
    conv_skip wgrad + dgrad -> resample adjoint -> weight-path backward (weight norm, gains, folded mp_cat scales).
Host orchestration over `dualdiffusion_amd.ops` only; activations NHWC bf16, master weights fp32.
"""
from __future__ import annotations

from dataclasses import dataclass, field
from typing import Optional

import torch

from .. import engine, ops
from .._lib import PRO_SCALE, PRO_SCALE_SILU, PRO_SILU, RESAMPLE_DOWN, RESAMPLE_DOWN_BWD, RESAMPLE_UP, RESAMPLE_UP_BWD
from .attention_grad import attention_backward


@dataclass
class BlockWeightsT:
    """fp32 master weights of one block (reference state-dict entries `<block>.<name>.weight`, `<block>.emb_gain`)."""
    conv_res0: torch.Tensor
    conv_res1: torch.Tensor
    emb_linear: torch.Tensor
    emb_gain: torch.Tensor            # 0-d / [1] parameter
    conv_skip: Optional[torch.Tensor] = None
    groups: int = 8
    # self-attention part (blocks of the attention levels; unet_edm2_b4.py:137-151)
    attn_qk: Optional[torch.Tensor] = None
    attn_v: Optional[torch.Tensor] = None
    attn_proj: Optional[torch.Tensor] = None
    emb_linear_qk: Optional[torch.Tensor] = None
    emb_gain_qk: Optional[torch.Tensor] = None
    emb_linear_v: Optional[torch.Tensor] = None
    emb_gain_v: Optional[torch.Tensor] = None
    heads: int = 0
    # training.weight_bank: prepared weights come from / weight gradients go to the bank's persistent buffers under the
    # state-dict keys `<prefix>.<layer>.weight`; the weight-path backward then runs once for all layers (bank.backward())
    bank: Optional[object] = None
    prefix: str = ""
    # persistent [B, O] fp32 buffers of the emb_linear* outputs and their gradients (keys c, dc, c_qk, dc_qk, c_v, dc_v) when the
    # caller evaluates every emb_linear* of the network in one batched launch (forward) / back-propagates them in one (backward)
    cvec: Optional[dict] = None


@dataclass
class BlockTape:
    w: BlockWeightsT
    flavor: str
    resample: str
    in0: torch.Tensor                 # block inputs as given (before resampling)
    in1: Optional[torch.Tensor]
    src0: torch.Tensor                # resampled inputs
    src1: Optional[torch.Tensor]
    s0: float
    s1: float
    emb: torch.Tensor
    c: torch.Tensor
    xs: Optional[torch.Tensor]        # enc: conv_skip output / resampled input before the pixel norm
    x1: Optional[torch.Tensor]        # enc: normalised x
    y0: torch.Tensor
    out: torch.Tensor
    res_t: float
    clip: float
    pw: dict = field(default_factory=dict)
    attn: Optional[dict] = None       # c_qk, c_v, qk, v, ao, ap, xs_qk, xa, attn_t
    # activated twins kept for the weight gradients: operands of conv_res0 (per source) and of conv_res1; the block output's
    a00: Optional[torch.Tensor] = None
    a01: Optional[torch.Tensor] = None
    a1: Optional[torch.Tensor] = None
    out_twin: Optional[torch.Tensor] = None
    up_skip: bool = False             # the skip branch ran at the source size (up block)
    dropout: Optional[tuple] = None   # (p, seed, stream id) of the magnitude-preserving dropout applied to a1 (unet_edm2_b4.py:124-125)


def _prep(w: BlockWeightsT, key: str, weight: torch.Tensor, groups: int, dt, **kw):
    """Forward prepared weight of layer `key`: the bank's (filled by bank.prepare()) or prepared here."""
    if w.bank is not None:
        return w.bank.pw[f"{w.prefix}.{key}.weight"]
    return ops.wprep(weight, groups, dt, normalize=True, **kw)


def _prep_t(w: BlockWeightsT, key: str, weight: torch.Tensor, groups: int, dt, **kw):
    """Data-gradient prepared weight of layer `key`."""
    if w.bank is not None:
        return w.bank.pwt[f"{w.prefix}.{key}.weight"]
    return ops.wprep(weight, groups, dt, normalize=True, transpose=True, **kw)


def _wgrad(w: BlockWeightsT, key: str, pw, dy: torch.Tensor, x0: torch.Tensor, groups: int, ksize: int, x1: Optional[torch.Tensor] = None):
    """Master-weight gradient of conv layer `key` (with a bank: the GEMM writes the bank's dwp slot and the returned view is
    filled by bank.backward())."""
    if w.bank is not None:
        name = f"{w.prefix}.{key}.weight"
        w.bank.wgrad(name, dy, x0, groups, ksize, x1=x1)
        return w.bank.dw[name]
    return ops.wprep_bwd(pw, ops.conv2d_wgrad(dy, x0, groups, ksize, x1=x1))


def _linear_bwd(w: BlockWeightsT, key: str, dc: torch.Tensor, emb: torch.Tensor, weight: torch.Tensor, groups: int, gain: torch.Tensor,
                demb: Optional[torch.Tensor]):
    """(dw, dgain) of an emb_linear* layer; accumulates the embedding gradient into demb."""
    if w.bank is not None:
        name = f"{w.prefix}.{key}.weight"
        if w.cvec is None:         # (with cvec the caller back-propagates every emb_linear* at once from the persistent dc buffers)
            ops.linear_small_bwd(dc, emb, weight, groups, gain, True, demb, row_scale=w.bank.rs[name], dwp=w.bank.dwp[name])
        return w.bank.dw[name], w.bank.dgain[name]
    return ops.linear_small_bwd(dc, emb, weight, groups, gain, True, demb)


def _resample(x: torch.Tensor, mode: str) -> torch.Tensor:
    if mode == "keep":
        return x
    B, H, W, Cn = x.shape
    out = torch.empty((B, H * 2, W * 2, Cn) if mode == "up" else (B, H // 2, W // 2, Cn), dtype=x.dtype, device=x.device)
    return ops.resample2d(x, out, RESAMPLE_UP if mode == "up" else RESAMPLE_DOWN)


def _resample_bwd(dx: torch.Tensor, mode: str) -> torch.Tensor:
    if mode == "keep":
        return dx
    B, H, W, Cn = dx.shape
    out = torch.empty((B, H // 2, W // 2, Cn) if mode == "up" else (B, H * 2, W * 2, Cn), dtype=dx.dtype, device=dx.device)
    return ops.resample2d(dx, out, RESAMPLE_UP_BWD if mode == "up" else RESAMPLE_DOWN_BWD)


def block_forward_train(in0: torch.Tensor, in1: Optional[torch.Tensor], s0: float, s1: float, emb: torch.Tensor, w: BlockWeightsT, *,
                        flavor: str, resample: str = "keep", res_t: float = 0.3, clip: float = 256.0, attn_t: float = 0.3,
                        act0: Optional[torch.Tensor] = None, act1: Optional[torch.Tensor] = None, twin_scale: Optional[float] = None,
                        dropout: Optional[tuple] = None):
    """in0 (| in1): NHWC bf16 block input(s) (mp_cat scales s0, s1); emb [B, Cemb] fp32.  Returns (out, tape).

    Producer-side activation, as in the inference plan: every conv operand that the reference activates on the fly is stored
    once as a bf16 twin next to the raw tensor -- conv_res0's epilogue writes y0 AND mp_silu(y0 * c), the block's last conv
    writes its output AND (twin_scale given) mp_silu(twin_scale * out) for the consumer -- so all convs stage their operands
    untouched (LDS-DMA kernels) and the backward finds the weight-gradient operands on the tape instead of recomputing them.
    act0 / act1: mp_silu(s0 * in0) / mp_silu(s1 * in1) from the producers (decoder blocks; computed here when absent).
    tape.out_twin: mp_silu(twin_scale * out) or None.
    dropout = (p, seed, stream id): y = F.dropout(mp_silu(y0 * c), p) * (1 - p)^0.5 (unet_edm2_b4.py:124-125) as an in-place pass over the
    activated twin, the keep mask a function of (seed, stream id, element) that the backward regenerates."""
    dt, G = in0.dtype, w.groups
    src0 = _resample(in0, resample)
    src1 = _resample(in1, resample) if in1 is not None else None
    C0 = src0.shape[-1]
    B = src0.shape[0]
    gain_ptr = w.emb_gain.reshape(1)
    Cmid = w.emb_linear.shape[0]
    if w.cvec is not None:
        c = w.cvec["c"]
    else:
        c = torch.empty(B, Cmid, dtype=torch.float32, device=in0.device)
        table = ops.make_linear_jobs([(w.emb_linear, gain_ptr, c, 1.0, 1.0, G, True)], in0.device)
        ops.linear_small(table, 1, Cmid, emb, B, w.emb_linear.dtype)
    pw = {"res0": _prep(w, "conv_res0", w.conv_res0, G, dt), "res1": _prep(w, "conv_res1", w.conv_res1, G, dt)}
    xs = x1 = a01 = None
    up_skip = False
    if flavor == "enc":
        assert src1 is None and s0 == 1.0
        if w.conv_skip is not None:
            pw["skip"] = _prep(w, "conv_skip", w.conv_skip, 1, dt)
            xs = ops.conv2d(src0, pw["skip"])
        else:
            xs = src0
        a00 = torch.empty_like(xs)
        x1 = ops.pixelnorm(xs, out_act=a00)                 # x1 and mp_silu(x1)
        sk = x1
    else:
        # nearest upsampling commutes with the activation: the producer's twin is resampled like the raw input
        a00 = _resample(act0, resample) if act0 is not None else ops.silu_scale_fwd(src0, None, s0)
        if src1 is not None:
            a01 = _resample(act1, resample) if act1 is not None else ops.silu_scale_fwd(src1, None, s1)
        # up block: the 1x1 skip conv commutes with the nearest resample, so it runs at the source size and conv_res1 gathers the
        # half-size residual (engine.RES_UP; ddx_conv_desc::residual_up) -- forward, data and weight gradient at a quarter of the pixels
        up_skip = engine.RES_UP and resample == "up" and src1 is None
        if w.conv_skip is not None:
            pw["skip"] = _prep(w, "conv_skip", w.conv_skip, 1, dt, in_split=C0 if src1 is not None else 0, in_scale0=s0, in_scale1=s1)
            sk = ops.conv2d(in0, pw["skip"]) if up_skip else ops.conv2d(src0, pw["skip"], src1=src1)
        else:
            assert src1 is None and s0 == 1.0
            sk = in0 if up_skip else src0
    y0 = torch.empty(src0.shape[:3] + (Cmid,), dtype=dt, device=in0.device)
    a1 = torch.empty_like(y0)
    ops.conv2d(a00, pw["res0"], src1=a01, out=y0, out_scale=c, out2=a1)      # y0 and mp_silu(y0 * c)
    if dropout is not None and dropout[0] > 0:
        ops.mp_dropout_(a1, *dropout)
    has_attn = w.attn_qk is not None
    out_twin = None
    tw = {}
    if twin_scale is not None:
        out_twin = torch.empty(src0.shape[:3] + (w.conv_res1.shape[0],), dtype=dt, device=in0.device)
        tw = dict(out2=out_twin, out2_scale=twin_scale)
    out = ops.conv2d(a1, pw["res1"], residual=sk, res_t=res_t, clip=0.0 if has_attn else clip, residual_up=up_skip, **({} if has_attn else tw))
    tape = BlockTape(w, flavor, resample, in0, in1, src0, src1, s0, s1, emb, c, xs, x1, y0, out, res_t, 0.0 if has_attn else clip, pw)
    tape.a00, tape.a01, tape.a1, tape.out_twin, tape.up_skip = a00, a01, a1, out_twin, up_skip
    tape.dropout = dropout if (dropout is not None and dropout[0] > 0) else None
    if not has_attn:
        return out, tape
    # ---- self-attention: qk = attn_qk(x * c_qk), v = attn_v(x), y = attn_proj(mp_silu(attention * c_v)), x = mp_sum(x, y, t)
    Cout = out.shape[-1]
    if w.cvec is not None:
        c_qk, c_v = w.cvec["c_qk"], w.cvec["c_v"]
    else:
        c_qk = torch.empty(B, Cout, dtype=torch.float32, device=in0.device)
        c_v = torch.empty(B, Cout, dtype=torch.float32, device=in0.device)
        table = ops.make_linear_jobs([(w.emb_linear_qk, w.emb_gain_qk.reshape(1), c_qk, 1.0, 1.0, 1, True),
                                      (w.emb_linear_v, w.emb_gain_v.reshape(1), c_v, 1.0, 1.0, 1, True)], in0.device)
        ops.linear_small(table, 2, Cout, emb, B, w.emb_linear_qk.dtype)
    pw["qk"] = _prep(w, "attn_qk", w.attn_qk, 1, dt, qk_head_dim=Cout // w.heads)
    pw["v"] = _prep(w, "attn_v", w.attn_v, 1, dt)
    pw["proj"] = _prep(w, "attn_proj", w.attn_proj, 1, dt)
    xs_qk = ops.silu_scale_fwd(out, c_qk, 1.0, act=False)                 # out * c_qk: operand of attn_qk and of its weight gradient
    qk = ops.conv2d(xs_qk, pw["qk"])
    vv = ops.conv2d(out, pw["v"])
    ao = ops.attention(qk, vv, w.heads)
    ap = ops.silu_scale_fwd(ao, c_v)
    xa = ops.conv2d(ap, pw["proj"], residual=out, res_t=attn_t, clip=clip, **tw)
    tape.attn = dict(c_qk=c_qk, c_v=c_v, qk=qk, v=vv, ao=ao, ap=ap, xs_qk=xs_qk, xa=xa, attn_t=attn_t, clip=clip)
    return xa, tape


def block_backward(t: BlockTape, dout: torch.Tensor, demb: Optional[torch.Tensor] = None) -> dict:
    """Gradients of one block.  Returns din0, din1, dw_{conv_res0,conv_res1,conv_skip,emb_linear}, demb_gain; accumulates the
    embedding gradient into `demb` [B, Cemb] fp32 when given."""
    w, dt, G = t.w, t.src0.dtype, t.w.groups
    C0 = t.src0.shape[-1]
    C1 = t.src1.shape[-1] if t.src1 is not None else 0
    g: dict = {}
    if t.attn is not None:
        a = t.attn
        one = w.emb_gain_qk.reshape(1)
        # xa = clip(mp_sum(out, attn_proj(ap), t)), ap = mp_silu(ao * c_v)
        dout_res, dyp = ops.mpsum_clip_bwd(dout, a["xa"], a["attn_t"], a["clip"])
        ap = a["ap"]
        g["dw_attn_proj"] = _wgrad(w, "attn_proj", t.pw["proj"], dyp, ap, 1, 1)
        dc_v = w.cvec["dc_v"] if w.cvec is not None else torch.zeros_like(a["c_v"])
        dao, _ = ops.conv2d_dgrad_act(dyp, _prep_t(w, "attn_proj", w.attn_proj, 1, dt), a["ao"], chan_scale=a["c_v"], dchan_scale=dc_v)
        g["dw_emb_linear_v"], g["demb_gain_v"] = _linear_bwd(w, "emb_linear_v", dc_v, t.emb, w.emb_linear_v, 1, w.emb_gain_v.reshape(1), demb)
        dqk, dv = attention_backward(a["qk"], a["v"], dao, w.heads)
        # v = attn_v(out);  qk = attn_qk(out * c_qk) with the (head, {q,k}, d) row order of the forward preparation
        g["dw_attn_v"] = _wgrad(w, "attn_v", t.pw["v"], dv, t.out, 1, 1)
        dout_v = ops.conv2d(dv, _prep_t(w, "attn_v", w.attn_v, 1, dt))
        xs_qk = a["xs_qk"]
        g["dw_attn_qk"] = _wgrad(w, "attn_qk", t.pw["qk"], dqk, xs_qk, 1, 1)
        dc_qk = w.cvec["dc_qk"] if w.cvec is not None else torch.zeros_like(a["c_qk"])
        dout_qk, _ = ops.conv2d_dgrad_act(dqk, _prep_t(w, "attn_qk", w.attn_qk, 1, dt, qk_head_dim=t.out.shape[-1] // w.heads), t.out,
                                          chan_scale=a["c_qk"], dchan_scale=dc_qk, add=dout_v, act=False)
        g["dw_emb_linear_qk"], g["demb_gain_qk"] = _linear_bwd(w, "emb_linear_qk", dc_qk, t.emb, w.emb_linear_qk, 1, one, demb)
        dout = ops.add3(dout_res, dout_qk)
    # out = clip(mp_sum(sk, y1, t))
    dsk, dy1 = ops.mpsum_clip_bwd(dout, t.out, t.res_t, t.clip)
    # y1 = conv_res1(a1), a1 = mp_silu(y0 * c)
    a1 = t.a1
    g["dw_conv_res1"] = _wgrad(w, "conv_res1", t.pw["res1"], dy1, a1, G, 3)
    # data gradient of conv_res1 through a1 = mp_silu(y0 * c): one launch, the activation backward runs in the conv's epilogue
    dc = w.cvec["dc"] if w.cvec is not None else torch.zeros_like(t.c)
    if t.dropout is not None:
        # a1 = dropout(mp_silu(y0 * c)): the gradient of the dropped operand gets the same keep / sqrt(1 - p) factor before the activation's
        # backward (three launches instead of the fused one: the mask sits between the conv and the activation)
        dA = ops.conv2d(dy1, _prep_t(w, "conv_res1", w.conv_res1, G, dt))
        ops.mp_dropout_(dA, *t.dropout)
        dy0 = ops.silu_scale_bwd(dA, t.y0, t.c, 1.0, dc=dc)
    else:
        dy0, _ = ops.conv2d_dgrad_act(dy1, _prep_t(w, "conv_res1", w.conv_res1, G, dt), t.y0, chan_scale=t.c, dchan_scale=dc)
    # c = emb_linear(emb) * emb_gain + 1
    g["dw_emb_linear"], g["demb_gain"] = _linear_bwd(w, "emb_linear", dc, t.emb, w.emb_linear, G, w.emb_gain.reshape(1), demb)
    g["dc"] = dc
    if t.flavor == "enc":
        a0 = t.a00
        g["dw_conv_res0"] = _wgrad(w, "conv_res0", t.pw["res0"], dy0, a0, G, 3)
        dx1, _ = ops.conv2d_dgrad_act(dy0, _prep_t(w, "conv_res0", w.conv_res0, G, dt), t.x1, add=dsk)
        dxs = ops.pixelnorm_bwd(dx1, t.xs)
        if w.conv_skip is not None:
            g["dw_conv_skip"] = _wgrad(w, "conv_skip", t.pw["skip"], dxs, t.src0, 1, 1)
            dsrc0 = ops.conv2d(dxs, _prep_t(w, "conv_skip", w.conv_skip, 1, dt))
        else:
            dsrc0 = dxs
        dsrc1 = None
    else:
        a00, a01 = t.a00, t.a01
        g["dw_conv_res0"] = _wgrad(w, "conv_res0", t.pw["res0"], dy0, a00, G, 3, x1=a01)
        if t.up_skip:
            # the skip branch ran before the resample: its gradient is pooled (2x2 sums) first, then everything is half-size
            dsk = _resample_bwd(dsk, "up")
            if w.conv_skip is not None:
                g["dw_conv_skip"] = _wgrad(w, "conv_skip", t.pw["skip"], dsk, t.in0, 1, 1)
                dxs = ops.conv2d(dsk, _prep_t(w, "conv_skip", w.conv_skip, 1, dt))
            else:
                dxs = dsk
            dsrc0, _ = ops.conv2d_dgrad_act(dy0, _prep_t(w, "conv_res0", w.conv_res0, G, dt), t.src0, scale0=t.s0)
            g["din0"], g["din1"] = ops.add3(_resample_bwd(dsrc0, "up"), dxs), None
            return g
        if w.conv_skip is not None:
            g["dw_conv_skip"] = _wgrad(w, "conv_skip", t.pw["skip"], dsk, t.src0, 1, 1, x1=t.src1)
            dxs = ops.conv2d(dsk, _prep_t(w, "conv_skip", w.conv_skip, 1, dt, in_split=C0 if C1 else 0, in_scale0=t.s0, in_scale1=t.s1))
        else:
            dxs = dsk
        # data gradient of conv_res0 through mp_silu(s0 * src0) | mp_silu(s1 * src1), plus the skip-path gradient
        dsrc0, dsrc1 = ops.conv2d_dgrad_act(dy0, _prep_t(w, "conv_res0", w.conv_res0, G, dt), t.src0, y1=t.src1 if C1 else None, scale0=t.s0,
                                            scale1=t.s1, add=dxs)
    g["din0"] = _resample_bwd(dsrc0, t.resample)
    g["din1"] = _resample_bwd(dsrc1, t.resample) if dsrc1 is not None else None
    return g
