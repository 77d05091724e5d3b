"""Launch-plan builder shared by the HIP-backed modules (UNet, VAE).

A module forward for a fixed (batch, size, dtype, training) is a static sequence of libddx_hip launches over static
buffers.  `PlanBuilder` collects, while the module walks its topology once:
  * the prepared-weight buffers and their weight-preparation launches (`wplan`, re-run when weights change),
  * the per-block modulation vectors `c = emb_linear(emb)*gain + 1` as jobs of ONE batched small-M kernel,
  * the activation buffers and the forward launches (`fplan`, replayed eagerly or as a hipGraph).
It also knows how to queue one EDM2 block (reference src/modules/unets/unet_edm2_b4.py:110-158 and its VAE twin
src/modules/old/vaes/vae_edm2.py:96-156), which is the same launch sequence for both model families.
"""
from __future__ import annotations

import math
import os
from typing import Callable, Optional

import torch

from . import ops
from ._lib import PRO_NONE, PRO_SCALE, PRO_SCALE_SILU, PRO_SILU, RESAMPLE_DOWN, RESAMPLE_KEEP, RESAMPLE_UP, Plan, weights_epoch

_RESAMPLE = {"keep": RESAMPLE_KEEP, "up": RESAMPLE_UP, "down": RESAMPLE_DOWN}


def mp_cat_weights(na: int, nb: int, t: float) -> tuple:
    """Scalars of the magnitude-preserving concat (reference mp_tools.py:294-301)."""
    c = math.sqrt((na + nb) / ((1 - t) ** 2 + t ** 2))
    return c / math.sqrt(na) * (1 - t), c / math.sqrt(nb) * t


# (Measured and removed: attn_qk / attn_v as two convs -- the merged conv saves 15 small-M launches, 5.23 -> 5.07 ms per step in round 1;
# the batched emb_linear launch on a side lane of the plan -- 5.14 vs 5.055 ms, the 130 MB weight stream slows the HBM-bound level-0 convs it
# runs beside by more than the 55 us it hides.)
# plan-time kernel selection by measurement (ops.tuning): every conv of an inference plan times its kernel candidates once.
# Measured (default UNet, hipGraph): B=1 2.77 -> 2.66 ms, B=4 4.99 -> 4.98 ms, B=8 8.44 -> 8.40 ms -- the built-in heuristics were
# tuned at B=4/8 and leave little there, so it is opt-in (adds ~1-2 s to the first call, run-to-run choices may differ).
AUTOTUNE = os.environ.get("DDX_AUTOTUNE", "0") != "0"
# 1x1 layers of an inference plan with at most this many output pixels (B*H*W) run on the small-M weight-streaming kernel
# (conv_sm.hip): their weights are prepared with 16-channel chunks, the layout that kernel streams straight into the MFMA operand
# registers.  Measured on MI355X (tools/conv_bench.py --cases small, graph-chained launches, B=4): level-4 1x1 layers (344 pixels)
# 7.0-12.9 us against 9.1-17.2 us on the register-staged split-K kernel; level-3 1x1 layers (1376 pixels) 13-31 us against 12-26 us,
# (round 4, eight-wave one-burst kernels: level-4 1x1 layers 7.2-15.6 us; a 3x3 variant of the same structure measured 8.3-23.7 us against
# 9.8-19.7 us on the register-staged kernel and 4.42 vs 4.30 ms on the whole step -- removed).  DDX_SM_MAX_PIXELS=0 switches the path off.
# Channel-blocked [B, C/16, H, W, 16] storage of the tensors that only 3x3 LDS-DMA convs read (conv_res0's output, the activated
# twins written by conv_res1): a K-stage of the consumer is then one contiguous plane slice instead of 32-byte granules a pixel
# stride apart.  Measured on MI355X (tools/conv_bench.py --epi real --path dma+dma16, B=4): conv_res1 at level 0 63.2 -> 49.3 us
# (512 -> 256 channels) and 166.6 -> 151.2 us (1024 -> 512), level 1 42.9 -> 38.7 us; conv_res0 layers 3-8 %; bit-identical results.
C16 = os.environ.get("DDX_C16", "1") != "0"
SM_MAX_PIXELS = int(os.environ.get("DDX_SM_MAX_PIXELS", "512"))
# The residual branch of an up block is conv_skip(upsample(x)): a 1x1 conv commutes with the nearest resample exactly, so the
# skip conv runs at the SOURCE size (a quarter of the matrix work and output bytes) and conv_res1 gathers the half-size
# residual in its epilogue (ddx_conv_desc::residual_up).  Inference plans (4.735 -> 4.63-4.66 ms per step in round 3; tests flip the constant).
RES_UP = True
# encoder blocks: normalize(conv_skip(x)) and its activated twin from the skip conv's epilogue where one unit holds all channels of a pixel
# (five pixelnorm launches less: 4.65 -> 4.60, 4.53 -> 4.49-4.51 ms in round 3)
FUSE_PIXELNORM = True
# attention blocks outside the small-M regime: x * c_qk as a materialised twin (written by conv_res1 where it runs on the register-staged
# kernel) + the merged qkv conv on the 1x1 GEMM kernel / the wide 1x1 units of the LDS-DMA kernel (0 = never)
QKV_TWIN_MIN_PIXELS = 1024
# attention blocks: normalize(q | k | v) in the epilogue of the merged qkv conv instead of inside every attention workgroup (tests flip it)
HEAD_NORM = True
# level-0 encoder blocks: conv_res0 -> conv_res1 as one launch with the hidden tensor in LDS (csrc/conv_pair.hip); DDX_CONV_PAIR=0 for the A/B
CONV_PAIR = os.environ.get("DDX_CONV_PAIR", "1") != "0"
PIXELNORM_EPS = 1e-4     # eps of normalize() (mp_tools.py:42-49), the default of ops.pixelnorm


class PlanBuilder:

    def __init__(self, device: torch.device, dtype: torch.dtype, batch: int, training: bool) -> None:
        self.dev, self.dt, self.B, self.training = device, dtype, batch, training
        self.keep: list = []        # every tensor the recorded launches point into
        self.gains: list = []       # 0-d gain parameters, mirrored into one fp32 vector read by the kernels
        self.convs: list = []       # weight-preparation specs
        self.padded: list = []      # (conv, zero-row-padded weight copy) for convs prepared with cout_pad
        self.lin_jobs: list = []    # (weight holder, gain slot, out tensor, groups, add_const)
        self.steps: list = []       # closures executed while recording the forward plan
        self.wplan, self.fplan = Plan(), Plan()
        self.gain_f32: Optional[torch.Tensor] = None
        self._weights_key = None

    # ------------------------------------------------------------------------------------------ resources
    def f32(self, *shape) -> torch.Tensor:
        t = torch.empty(*shape, device=self.dev, dtype=torch.float32)
        self.keep.append(t)
        return t

    def act(self, h: int, w: int, c: int) -> torch.Tensor:
        t = torch.empty(self.B, h, w, c, device=self.dev, dtype=self.dt)
        self.keep.append(t)
        return t

    def pick_ck(self, Cg: int, ks: int, npix: int) -> int:
        # (1x1 layers: a wave of the small-M kernel owns a quarter of the channels in whole 64-channel lines, four steps in flight)
        if (not self.training and self.dt == torch.bfloat16 and 0 < npix <= SM_MAX_PIXELS and Cg >= 32 and
                ks == 1 and Cg % 256 == 0):
            return 16
        return ops.pick_ck(Cg, ks, self.dt, npix)

    def c16_ok(self) -> bool:
        return C16 and not self.training and self.dt == torch.bfloat16

    def gain_slot(self, p) -> int:
        self.gains.append(p)
        return len(self.gains) - 1

    def prep(self, conv, gain_param=None, qk_head_dim: int = 0, cg_pad: Optional[int] = None, npix: int = 0,
             in_split: int = 0, in_scale0: float = 1.0, in_scale1: float = 1.0, cout_pad: int = 0, allow_sm: bool = True):
        """Declare a conv's prepared weights; the buffer is filled whenever `wplan` runs.
        in_split / in_scale*: mp_cat scales of a linear consumer folded into the weights.
        cout_pad: prepare from a zero-row-padded copy of the weight (refreshed with the weights), so that a conv with
        fewer than 4 output channels still runs on the matrix-core kernels; the consumer reads the first channels."""
        w = conv.weight
        wsrc = None
        if cout_pad > w.shape[0]:
            wsrc = torch.zeros((cout_pad,) + tuple(w.shape[1:]), dtype=w.dtype, device=self.dev)
            self.padded.append((conv, wsrc))
            w = wsrc
        Cg, ks = w.shape[1], (w.shape[2] if w.ndim == 4 else 1)
        CK = self.pick_ck(Cg, ks, npix) if (cg_pad is None and allow_sm) else ops.pick_ck(cg_pad or Cg, ks, self.dt, npix)
        nbytes = ops.lib().ddx_wprep_bytes(w.shape[0], Cg, ks, conv.groups, CK, ops.dtype_code(self.dt))
        buf = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self.keep.append(buf)
        self.convs.append(dict(conv=conv, wsrc=wsrc, buf=buf, CK=CK, qk=qk_head_dim, cg_pad=cg_pad, in_split=in_split, in_scale0=in_scale0,
                               in_scale1=in_scale1, gain_slot=self.gain_slot(gain_param) if gain_param is not None else None))
        return ops.PreparedWeight(buf, w.shape[0], Cg, ks, conv.groups, CK, self.dt, None)

    def prep_merged(self, parts: list, npix: int = 0, allow_sm: bool = True):
        """Several 1x1 convs on the same input as ONE prepared matrix (rows concatenated): parts = [(conv, qk_head_dim), ...].
        Each part is prepared into its row range whenever `wplan` runs."""
        w0 = parts[0][0].weight
        Cg = w0.shape[1]
        total = sum(c.weight.shape[0] for c, _ in parts)
        assert all(c.weight.shape[1] == Cg and c.groups == 1 and c.weight.ndim in (2, 4) and (c.weight.ndim == 2 or c.weight.shape[2] == 1)
                   for c, _ in parts)
        CK = self.pick_ck(Cg, 1, npix) if allow_sm else ops.pick_ck(Cg, 1, self.dt, npix)
        nbytes = ops.lib().ddx_wprep_bytes(total, Cg, 1, 1, CK, ops.dtype_code(self.dt))
        buf = torch.zeros(nbytes, dtype=torch.uint8, device=self.dev)       # padding rows stay zero
        self.keep.append(buf)
        off = 0
        for conv, qk in parts:
            self.convs.append(dict(conv=conv, wsrc=None, buf=buf, CK=CK, qk=qk, cg_pad=None, in_split=0, in_scale0=1.0, in_scale1=1.0,
                                   gain_slot=None, row_offset=off, rows_total=total))
            off += conv.weight.shape[0]
        return ops.PreparedWeight(buf, total, Cg, 1, 1, CK, self.dt, None)

    def cvec(self, lin, gain_param, add_const: float = 1.0) -> torch.Tensor:
        """Per-(batch, channel) vector lin(emb)*gain + add_const, produced by the batched small-M kernel."""
        out = self.f32(self.B, lin.out_channels)
        self.lin_jobs.append((lin, self.gain_slot(gain_param) if gain_param is not None else None, out, lin.groups, add_const))
        return out

    def step(self, fn: Callable[[], None]) -> None:
        self.steps.append(fn)

    # ------------------------------------------------------------------------------------------ one EDM2 block
    def block(self, blk, src0: torch.Tensor, src1: Optional[torch.Tensor], s0: float, s1: float, h: int, w: int,
              mlp_multiplier: int, res_balance: float, attn_balance: float, clip: float = 256.0,
              act0: Optional[torch.Tensor] = None, act1: Optional[torch.Tensor] = None, twin_scale: Optional[float] = None):
        """Queue one block.  `src0/src1`: raw NHWC inputs at the PRE-resample size (mp_cat scales s0/s1);
        `act0/act1`: their activated twins mp_silu(s * x) written by the producers (None: conv_res0 falls back to the fused
        prologue).  `twin_scale`: write mp_silu(twin_scale * out) beside the output for the next block's conv_res0.
        Returns (out, out_twin | None).

        Producer-side activation: conv_res0's epilogue stores mp_silu(y * c) and the attention kernel stores mp_silu(o * c_v),
        so conv_res1 / attn_proj / (with twins) conv_res0 stage their operands untouched -- the per-chunk prologue VALU
        work leaves the MFMA loop and is done once per element on fp32 accumulators instead."""
        cout, mm = blk.out_channels, mlp_multiplier
        rs = _RESAMPLE[blk.resample_mode]
        npix = self.B * h * w
        c_emb = self.cvec(blk.emb_linear, blk.emb_gain)
        pw_res0, pw_res1 = self.prep(blk.conv_res0, npix=npix), self.prep(blk.conv_res1, npix=npix)
        y0, xo = self.act(h, w, cout * mm), self.act(h, w, cout)
        attn = blk.use_attention
        last_clip = 0.0 if attn else clip
        twin = self.act(h, w, cout) if twin_scale is not None else None
        tw_res1 = dict(out2=twin, out2_scale=twin_scale) if (twin is not None and not attn) else {}
        if attn:
            c_qk, c_v = self.cvec(blk.emb_linear_qk, blk.emb_gain_qk), self.cvec(blk.emb_linear_v, blk.emb_gain_v)
        S = self.step
        if blk.flavor == "enc":
            pw_skip = self.prep(blk.conv_skip, npix=npix) if blk.conv_skip is not None else None
            x1, x1a = self.act(h, w, cout), self.act(h, w, cout)
            if pw_skip is not None and rs == RESAMPLE_DOWN:
                # avg-pool once (memory-bound) and run the skip conv on the pooled tensor: the in-conv 2x2 gather costs
                # four strided loads per staged vector and serialises the small-M layers
                xd = self.act(h, w, src0.shape[3])
                S(lambda: ops.resample2d(src0, xd, rs))
                if self._fused_pixelnorm(xd, pw_skip, x1, x1a):
                    S(lambda: ops.conv2d(xd, pw_skip, out=x1, out2=x1a, out2_scale=1.0, pixelnorm_eps=PIXELNORM_EPS))
                else:
                    S(lambda: ops.conv2d(xd, pw_skip, out=x1))
                    S(lambda: ops.pixelnorm(x1, out=x1, out_act=x1a))
            elif pw_skip is not None and rs == RESAMPLE_KEEP and self._fused_pixelnorm(src0, pw_skip, x1, x1a):
                # wide 1x1 skip conv with the pixel norm and the activated twin in its epilogue (one 256-channel unit sees a whole pixel)
                S(lambda: ops.conv2d(src0, pw_skip, out=x1, out2=x1a, out2_scale=1.0, pixelnorm_eps=PIXELNORM_EPS))
            elif pw_skip is not None:
                S(lambda: ops.conv2d(src0, pw_skip, out_hw=(h, w), resample=rs, out=x1))
                S(lambda: ops.pixelnorm(x1, out=x1, out_act=x1a))
            elif rs != RESAMPLE_KEEP:
                S(lambda: ops.resample2d(src0, x1, rs))
                S(lambda: ops.pixelnorm(x1, out=x1, out_act=x1a))
            else:
                S(lambda: ops.pixelnorm(src0, out=x1, out_act=x1a))
            kw0 = dict(out_act=True, out_scale=c_emb, out=y0)
            kw1 = dict(residual=x1, res_t=res_balance, clip=last_clip, out=xo, **tw_res1)
            if self._pair_ok(blk, cout, mm, pw_res0, pw_res1, attn):
                # level 0 (32 -> 64 -> 32 channels per group): conv_res0 -> mp_silu(y * c) -> conv_res1 -> mp_sum as ONE launch, the hidden
                # tensor stays in LDS (csrc/conv_pair.hip; 115 -> 88 us per block at B = 4, 950 -> 700 us at B = 32)
                self.keep = [t for t in self.keep if t is not y0]      # (the hidden tensor is never materialised)
                S(lambda: ops.conv_pair(x1a, pw_res0, pw_res1, c_emb, x1, res_balance, clip=last_clip, out=xo,
                                        out2=tw_res1.get("out2"), out2_scale=tw_res1.get("out2_scale", 1.0)))
            else:
                self._block_layouts(None, None, x1a, pw_res0, kw0, y0, pw_res1, kw1, twin if not attn else None)
                S(lambda: ops.conv2d(x1a, pw_res0, **kw0))
                S(lambda: ops.conv2d(y0, pw_res1, **kw1))
        else:
            kw0 = None
            if act0 is not None and (src1 is None or act1 is not None):
                kw0 = dict(out_hw=(h, w), src1=act1, resample=rs, out_act=True, out_scale=c_emb, out=y0)
                S(lambda: ops.conv2d(act0, pw_res0, **kw0))
            else:   # no twins available: fused prologue on the raw inputs
                S(lambda: ops.conv2d(src0, pw_res0, out_hw=(h, w), src1=src1, scale0=s0, scale1=s1, resample=rs, prologue=PRO_SILU,
                                     out_act=True, out_scale=c_emb, out=y0))
            res_up = False
            if RES_UP and not self.training and rs == RESAMPLE_UP and src1 is None:
                # up block: residual at the source size, gathered by conv_res1 (see RES_UP above)
                res_up = True
                if blk.conv_skip is not None:
                    sh, sw = src0.shape[1], src0.shape[2]
                    pw_skip = self.prep(blk.conv_skip, npix=self.B * sh * sw)
                    sk = self.act(sh, sw, cout)
                    self.steps.insert(len(self.steps) - 1, lambda: ops.conv2d(src0, pw_skip, out=sk))
                else:
                    sk = src0
            elif blk.conv_skip is not None:
                # (the skip conv only depends on the block input: queued ahead of conv_res0.  Running the two on parallel branches of the
                # graph measured 2.5 % slower -- every fork / join pair costs more cross-queue synchronisation than the overlapped 15-20 us
                # kernels save; round 4: two half-batch chains on two branches 525 vs 457 us for the full batch on one, tools/lane_probe.py)
                pw_skip = self.prep(blk.conv_skip, npix=npix, in_split=src0.shape[3] if src1 is not None else 0, in_scale0=s0, in_scale1=s1)
                sk = self.act(h, w, cout)
                self.steps.insert(len(self.steps) - 1, lambda: ops.conv2d(src0, pw_skip, out_hw=(h, w), src1=src1, resample=rs, out=sk))
            elif rs != RESAMPLE_KEEP:
                sk = self.act(h, w, cout)      # no skip conv: the residual is the (resampled) block input itself
                S(lambda: ops.resample2d(src0, sk, rs))
            else:
                assert src1 is None, "a concatenated input always changes the channel count, i.e. has a skip conv"
                sk = src0
            kw1 = dict(residual=sk, res_t=res_balance, clip=last_clip, out=xo, residual_up=res_up, **tw_res1)
            self._block_layouts(act0, act1, act0 if kw0 is not None else None, pw_res0, kw0, y0, pw_res1, kw1, twin if not attn else None)
            S(lambda: ops.conv2d(y0, pw_res1, **kw1))
        if not attn:
            return xo, twin
        heads = blk.num_heads
        # attn_qk and attn_v read the same tensor: ONE conv over the row-concatenated weights writes [q|k (2C) | v (C)]; the
        # channel-scale prologue (x * c_qk) only applies to the q|k output tiles (one 15 us small-M launch less per block)
        pw_proj = self.prep(blk.attn_proj, npix=npix)
        ao, xa = self.act(h, w, cout), self.act(h, w, cout)
        tw_proj = dict(out2=twin, out2_scale=twin_scale) if twin is not None else {}
        pw_qkv = self.prep_merged([(blk.attn_qk, cout // heads), (blk.attn_v, 0)], npix=npix, allow_sm=False)
        qkv = self.act(h, w, 3 * cout)
        qk, vv = qkv[..., :2 * cout], qkv[..., 2 * cout:]
        if self._qkv_twin_ok(xo, pw_qkv, cout, qkv, npix):
            # from QKV_TWIN_MIN_PIXELS pixels: x * c_qk is a materialised twin and the q|k output tiles of the merged conv read it through
            # src0_alt, so the conv takes raw operands and runs on the 1x1 GEMM kernel (level 3 at B=4: 35 -> 21-24 us) or the wide 1x1
            # units of the LDS-DMA kernel (B=32: 430 -> 700+ TFLOP/s)
            xs2 = self.act(h, w, cout)
            try:    # conv_res1 on the register-staged kernel writes the twin itself (kw1 is read when the queued step runs)
                twin_in_conv = ops.conv2d(y0, pw_res1, query=True, **dict(kw1, out2=xs2, out2_chan_scale=c_qk)) == 2
            except Exception:
                twin_in_conv = False
            if twin_in_conv:
                kw1.update(out2=xs2, out2_chan_scale=c_qk)
            else:
                S(lambda: ops.silu_scale_fwd(xo, c_qk, 1.0, act=False, out=xs2))
            kwq = dict(src0_alt=xs2, prologue_rows=2 * cout, out=qkv)
        else:
            kwq = dict(prologue=PRO_SCALE, chan_scale=c_qk, prologue_rows=2 * cout, out=qkv)
        # q, k, v normalised per head in the merged conv's epilogue (fp32, on the accumulators) where its kernel serves it: the attention
        # kernel then stages its operands untouched (isolated 18.1 -> 14.6 us at 344 tokens, 9.0 -> 7.7 us at 86)
        pre = False
        if HEAD_NORM and not self.training and self.dt == torch.bfloat16 and cout // heads == 64:
            try:
                pre = ops.conv2d(xo, pw_qkv, query=True, head_norm=64, **kwq) in (2, 6)
            except Exception:
                pre = False
        if pre:
            kwq.update(head_norm=64, head_eps=PIXELNORM_EPS)
        S(lambda: ops.conv2d(xo, pw_qkv, **kwq))
        S(lambda: ops.attention(qk, vv, heads, out=ao, out_scale=c_v, prenorm=pre))
        S(lambda: ops.conv2d(ao, pw_proj, residual=xo, res_t=attn_balance, clip=clip, out=xa, **tw_proj))
        return xa, twin

    def _pair_ok(self, blk, cout, mm, pw_res0, pw_res1, attn) -> bool:
        """Fused conv_res0 -> conv_res1 launch for this encoder block (bf16 inference, 32 -> 64 -> 32 channels per group, no attention)?"""
        if not CONV_PAIR or self.training or self.dt != torch.bfloat16 or attn or pw_res0.CK != 32 or pw_res1.CK != 32:
            return False
        return ops.conv_pair_supported(self.B, cout, blk.conv_res0.groups, cout * mm, self.dt)

    def _qkv_twin_ok(self, xo, pw_qkv, cout, qkv, npix) -> bool:
        """Merged attn_qk | attn_v conv on raw operands with a materialised x * c_qk twin (src0_alt)?  From QKV_TWIN_MIN_PIXELS pixels
        (level 3 at B=4; below that -- 344 pixels at level 4 -- neither the GEMM kernel nor the LDS-DMA units have enough tiles)."""
        if self.training or self.dt != torch.bfloat16 or QKV_TWIN_MIN_PIXELS <= 0 or npix < QKV_TWIN_MIN_PIXELS:
            return False
        try:
            return ops.conv2d(xo, pw_qkv, src0_alt=xo, prologue_rows=2 * cout, out=qkv, query=True) in (3, 6)
        except Exception:
            return False

    def _fused_pixelnorm(self, src0, pw_skip, x1, x1a) -> bool:
        """Does the library run this encoder skip conv with DDX_EPI_PIXELNORM (bf16 inference plans)?"""
        if not FUSE_PIXELNORM or self.training or self.dt != torch.bfloat16:
            return False
        try:
            return ops.conv2d(src0, pw_skip, out=x1, out2=x1a, out2_scale=1.0, pixelnorm_eps=PIXELNORM_EPS, query=True) == 3
        except Exception:
            return False

    def _block_layouts(self, act0, act1, in0, pw_res0, kw0, y0, pw_res1, kw1, twin) -> None:
        """Decide which tensors of a block are channel-blocked (ops.mark_c16): the library is asked which kernel each of the two 3x3
        convs will run on (`query`); only the LDS-DMA kernel reads / writes the blocked layout.
          * y0 (conv_res0 -> conv_res1): blocked when both run there;
          * the twin conv_res1 writes for the NEXT block: blocked when conv_res1 runs there -- the consumer takes the mark back when
            its own conv_res0 does not (the marks are read when the plan is recorded, after every block was declared);
          * act0 / act1 (twins this block consumes): un-marked when this block's conv_res0 is not an LDS-DMA launch."""
        q0 = ops.conv2d(in0, pw_res0, query=True, **kw0) if (in0 is not None and kw0 is not None) else 0
        if q0 != 3:
            for t in (act0, act1):
                if t is not None:
                    ops.mark_c16(t, False)
        if not self.c16_ok():
            return
        q1 = ops.conv2d(y0, pw_res1, query=True, **kw1)
        if q0 == 3 and q1 == 3 and y0.shape[3] % 16 == 0:
            ops.mark_c16(y0)
        if twin is not None and q1 == 3 and twin.shape[3] % 16 == 0 and kw1.get("out2") is twin:
            ops.mark_c16(twin)

    # ------------------------------------------------------------------------------------------ finalize / run
    def gain_ptr(self, slot: Optional[int]):
        return self.gain_f32[slot:slot + 1] if slot is not None else None

    def finalize(self, emb: Optional[torch.Tensor], emb_stride: int, pre_steps: Optional[Callable[[], None]] = None) -> None:
        """Build the job table of the modulation vectors (input `emb` [B, emb_stride] fp32) and record both plans.
        `pre_steps` runs first inside the forward plan (front end that produces `emb`)."""
        self.gain_f32 = torch.zeros(max(len(self.gains), 1), device=self.dev, dtype=torch.float32)
        if self.lin_jobs:
            self.emb_table = ops.make_linear_jobs(
                [(lin.weight, self.gain_ptr(slot), out, 1.0, addc, groups, self.training and not getattr(lin, "disable_weight_norm", False))
                 for (lin, slot, out, groups, addc) in self.lin_jobs], self.dev)
            n_jobs, max_o = len(self.lin_jobs), max(lin.out_channels for (lin, *_r) in self.lin_jobs)
            wdt = self.lin_jobs[0][0].weight.dtype
        with self.wplan.record():
            for sp in self.convs:
                conv = sp["conv"]
                ops.wprep(conv.weight if sp["wsrc"] is None else sp["wsrc"], conv.groups, self.dt, gain_ptr=self.gain_ptr(sp["gain_slot"]),
                          normalize=self.training and not conv.disable_weight_norm, qk_head_dim=sp["qk"], CK=sp["CK"],
                          cg_pad=sp["cg_pad"], out=sp["buf"], in_split=sp["in_split"], in_scale0=sp["in_scale0"],
                          in_scale1=sp["in_scale1"], row_offset=sp.get("row_offset", 0), rows_total=sp.get("rows_total", 0))
        if AUTOTUNE and not self.training and self.dt == torch.bfloat16:
            # one eager pass over the steps in which every conv times its kernel candidates on the plan's own buffers
            # (ops.tuning); the recording below then asks for the winners.  Weights must be prepared for it.
            self.wplan.run()
            with ops.tuning():
                if pre_steps is not None:
                    pre_steps()
                if self.lin_jobs:
                    ops.linear_small(self.emb_table, n_jobs, max_o, emb, self.B, wdt, x_stride=emb_stride)
                for st in self.steps:
                    st()
            torch.cuda.current_stream().synchronize()
        with self.fplan.record():
            if pre_steps is not None:
                pre_steps()
            # the batched emb_linear launch streams every block's modulation weights (~130 MB for the default UNet, ~55 us)
            if self.lin_jobs:
                ops.linear_small(self.emb_table, n_jobs, max_o, emb, self.B, wdt, x_stride=emb_stride)
            for st in self.steps:
                st()

    def refresh_weights(self, params) -> None:
        """Re-run weight preparation when training (forced weight norm every forward) or when any parameter changed."""
        key = None if self.training else (weights_epoch(),) + tuple(p._version for p in params)
        if self.training or key != self._weights_key:
            if self.gains:
                self.gain_f32.copy_(torch.stack([g.detach().float().reshape(()) for g in self.gains]))
            for conv, wsrc in self.padded:
                wsrc[:conv.weight.shape[0]].copy_(conv.weight.detach())
            self.wplan.run()
            self._weights_key = key

    def launch(self, use_graph: bool) -> None:
        if use_graph:
            if not self.fplan.has_graph:
                self.fplan.run()                            # warm-up outside capture (function attributes, module load)
                torch.cuda.current_stream().synchronize()
                cap = torch.cuda.Stream(device=self.dev)    # the legacy default stream cannot be captured
                self.fplan.graph_build(cap.cuda_stream)
                cap.synchronize()
            self.fplan.graph_launch()
        else:
            self.fplan.run()
