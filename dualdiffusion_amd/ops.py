"""Tensor-level wrappers over the C ABI (libddx_hip.so).

These are the host-side counterparts of the reference's op library (src/modules/mp_tools.py) for the hot
path: every function enqueues hand-written HIP kernels on the current torch stream; none has a PyTorch
fallback.  Activations are NHWC (`[B, H, W, C]` contiguous) float32 or bfloat16 device tensors.
"""
from __future__ import annotations

import ctypes as C
import os
from typing import Optional

import torch

from . import _lib as L
from ._lib import check, current_stream, dtype_code, lib, ptr


class PreparedWeight:
    """Output of weight preparation (mp_tools.py:359-364) in the implicit-GEMM layout."""

    __slots__ = ("wp", "Cout", "Cg", "ksize", "groups", "CK", "dtype", "desc", "workspace", "refs")

    def __init__(self, wp, Cout, Cg, ksize, groups, CK, dtype, desc):
        self.wp, self.Cout, self.Cg, self.ksize, self.groups, self.CK, self.dtype, self.desc = wp, Cout, Cg, ksize, groups, CK, dtype, desc
        self.workspace, self.refs = None, None   # tensors the descriptor points at (master weight, gain) are kept alive here


def pick_ck(Cg: int, ksize: int, dtype: torch.dtype, npix: int = 0) -> int:
    return int(lib().ddx_mpconv2d_pick_ck(Cg, ksize, dtype_code(dtype), npix))


def wprep(weight: torch.Tensor, groups: int, dtype: torch.dtype, *, gain: float = 1.0, gain_ptr: Optional[torch.Tensor] = None,
          normalize: bool = False, qk_head_dim: int = 0, CK: Optional[int] = None, cg_pad: Optional[int] = None,
          out: Optional[torch.Tensor] = None, npix: int = 0, in_split: int = 0, in_scale0: float = 1.0,
          in_scale1: float = 1.0, transpose: bool = False, row_offset: int = 0, rows_total: int = 0) -> PreparedWeight:
    """Prepare MPConv weights `[Cout, Cg, k, k]` for ddx_mpconv2d_fwd.  `cg_pad`: channel count of the activation
    tensor per group when it is zero-padded beyond the weight's Cg (conv_in: 6 -> 8).
    transpose: prepare the data-gradient conv instead (dX = conv2d(dY, wprep(w, transpose=True)))."""
    assert weight.is_contiguous()
    Cout, Cg = weight.shape[0], weight.shape[1]
    ksize = weight.shape[2] if weight.ndim == 4 else 1
    if transpose:
        return _wprep_transposed(weight, groups, dtype, gain, gain_ptr, normalize, qk_head_dim, CK, out, npix, in_split, in_scale0, in_scale1)
    if CK is None:
        CK = pick_ck(cg_pad or Cg, ksize, dtype, npix)
    nbytes = lib().ddx_wprep_bytes(rows_total or Cout, Cg, ksize, groups, CK, dtype_code(dtype))
    if cg_pad is not None:
        assert (cg_pad + CK - 1) // CK == (Cg + CK - 1) // CK, "padded channels must stay inside the last K chunk"
    if out is None:
        assert rows_total == 0, "a merged prepared matrix is allocated (zero-filled) by the caller"
        out = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    assert out.numel() >= nbytes
    d = L.WPrepDesc(w=ptr(weight), wp=ptr(out), gain_ptr=ptr(gain_ptr), gain=float(gain), w_dtype=dtype_code(weight.dtype),
                    wp_dtype=dtype_code(dtype), Cout=Cout, Cg=Cg, ksize=ksize, groups=groups, CK=CK,
                    normalize=int(normalize), qk_head_dim=qk_head_dim, in_split=in_split, in_scale0=in_scale0, in_scale1=in_scale1,
                    row_offset=row_offset, rows_total=rows_total)
    check(lib().ddx_mpconv_wprep(C.byref(d), current_stream()), "mpconv_wprep")
    pw = PreparedWeight(out, rows_total or Cout, Cg, ksize, groups, CK, dtype, d)
    pw.refs = (weight, gain_ptr)
    return pw


def _wprep_transposed(weight, groups, dtype, gain, gain_ptr, normalize, qk_head_dim, CK, out, npix, in_split, in_scale0, in_scale1):
    Cout, Cg = weight.shape[0], weight.shape[1]
    ksize = weight.shape[2] if weight.ndim == 4 else 1
    Ng, Cin = Cout // groups, Cg * groups
    if CK is None:
        CK = pick_ck(Ng, ksize, dtype, npix)
    nbytes = lib().ddx_wprep_bytes(Cin, Ng, ksize, groups, CK, dtype_code(dtype))
    if out is None:
        out = torch.empty(nbytes, dtype=torch.uint8, device=weight.device)
    assert out.numel() >= nbytes
    ws = torch.empty(Cout, dtype=torch.float32, device=weight.device)
    d = L.WPrepDesc(w=ptr(weight), wp=ptr(out), gain_ptr=ptr(gain_ptr), gain=float(gain), w_dtype=dtype_code(weight.dtype),
                    wp_dtype=dtype_code(dtype), Cout=Cout, Cg=Cg, ksize=ksize, groups=groups, CK=CK, normalize=int(normalize),
                    qk_head_dim=qk_head_dim, in_split=in_split, in_scale0=in_scale0, in_scale1=in_scale1, transpose=1, row_scale=ptr(ws))
    check(lib().ddx_mpconv_wprep(C.byref(d), current_stream()), "mpconv_wprep(transpose)")
    pw = PreparedWeight(out, Cin, Ng, ksize, groups, CK, dtype, d)
    pw.workspace = ws      # keeps the row-scale workspace alive as long as the prepared weight
    pw.refs = (weight, gain_ptr)
    return pw


def normalize_weights_(weight: torch.Tensor) -> None:
    """In-place forced weight normalisation (mp_tools.py:375-378)."""
    rows = weight.shape[0]
    check(lib().ddx_normalize_weights(ptr(weight), dtype_code(weight.dtype), rows, weight.numel() // rows, current_stream()),
          "normalize_weights")


# ---- plan-time kernel selection by measurement.  Inside `with tuning():` every conv2d call times the kernel candidates on
# its real operands (heuristic choice, LDS-DMA kernel, every built tile / split-K configuration of the register-staged kernel)
# and remembers the fastest per layer signature; later calls (the plan recording) ask for that kernel.  The heuristic stays
# unless a candidate is at least 4 % faster.
_PATH_CODE = {"auto": 0, "direct": 1, "mfma": 2, "dma": 3, "sm": 4, "few": 5, "gemm": 6}
_conv_choice: dict = {}
_tuning = False


class tuning:
    def __enter__(self):
        global _tuning
        self.prev, _tuning = _tuning, True
        return self

    def __exit__(self, *exc):
        global _tuning
        _tuning = self.prev
        return False


def _conv_signature(d: "L.ConvDesc") -> tuple:
    return (d.B, d.H, d.W, d.C0, d.C1, d.Cout, d.groups, d.ksize, d.CK, d.resample, d.prologue, d.epilogue, d.dtype, d.out_act,
            bool(d.chan_scale), bool(d.out_scale), bool(d.out2), d.pad_mode, d.prologue_rows, d.scale0 == 1.0, d.scale1 == 1.0, d.clip > 0,
            bool(d.src0_alt), d.out2_linear, d.residual_up, d.out_head_norm)


def _tune_conv(d: "L.ConvDesc") -> int:
    st = current_stream()

    def run(code: int, n: int) -> bool:
        d.force_direct = code
        for _ in range(n):
            if lib().ddx_mpconv2d_fwd(C.byref(d), st) != 0:
                return False
        return True

    def timed(code: int) -> float:
        if not run(code, 2):
            return float("inf")
        best = float("inf")
        for _ in range(2):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            run(code, 8)
            e1.record()
            e1.synchronize()
            best = min(best, e0.elapsed_time(e1) / 8)
        return best

    t_auto = timed(0)
    best_code, best_t = 0, t_auto
    for code in [3, 6] + [16 + i for i in range(12)]:
        t = timed(code)
        if t < best_t:
            best_code, best_t = code, t
    return best_code if best_t < 0.96 * t_auto else 0


def mark_c16(t: torch.Tensor, on: bool = True) -> torch.Tensor:
    """Declare that the bytes of NHWC-shaped `t` [B, H, W, C] are (to be) laid out channel-blocked as [B, C/16, H, W, 16]."""
    t._ddx_c16 = bool(on)
    return t


def is_c16(t: Optional[torch.Tensor]) -> bool:
    return t is not None and getattr(t, "_ddx_c16", False)


def to_c16(x: torch.Tensor) -> torch.Tensor:
    """NHWC values -> a marked channel-blocked tensor of the same shape attribute (host-side helper for tests and tools)."""
    B, H, W, Cn = x.shape
    return mark_c16(x.reshape(B, H, W, Cn // 16, 16).permute(0, 3, 1, 2, 4).contiguous().reshape(B, H, W, Cn))


def from_c16(x: torch.Tensor) -> torch.Tensor:
    B, H, W, Cn = x.shape
    return x.reshape(B, Cn // 16, H, W, 16).permute(0, 2, 3, 1, 4).contiguous().reshape(B, H, W, Cn)


def conv2d(src0: torch.Tensor, pw: PreparedWeight, *, out_hw: Optional[tuple] = None, src1: Optional[torch.Tensor] = None,
           scale0: float = 1.0, scale1: float = 1.0, resample: int = L.RESAMPLE_KEEP, prologue: int = L.PRO_NONE,
           chan_scale: Optional[torch.Tensor] = None, residual: Optional[torch.Tensor] = None, res_t: float = 0.0,
           clip: float = 0.0, out: Optional[torch.Tensor] = None, force_direct: bool = False, out_act: bool = False,
           out_scale: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None, out2_scale: float = 1.0,
           path: str = "auto", reflect_w: bool = False, prologue_rows: int = 0, swap_src1: bool = False, swap_paired: bool = False, pixelnorm_eps: float = 0.0,
           out2_chan_scale: Optional[torch.Tensor] = None, src0_alt: Optional[torch.Tensor] = None, query: bool = False,
           residual_up: bool = False, head_norm: int = 0, head_eps: float = 1e-4):
    """Magnitude-preserving conv2d forward with fused prologue / epilogue (see include/ddx_hip.h).
    path: "auto" | "direct" (scalar kernel) | "mfma" (register-staged) | "dma" (LDS-DMA staged; raw bf16 operands only)
    | "sm" (small-M weight-streaming kernel; weights prepared with CK = 16, which also selects it automatically)
    | "few" (3x3 over 8 zero-padded input channels: the input convs; the automatic choice where it applies)
    | "gemm" (mid-size raw 1x1 layers: 128 x 128 GEMM tiles on a deep LDS-DMA ring; the automatic choice inside its size window)
    | an integer force_direct code (16 + 3 * tile + k: one tile / split-K configuration of the register-staged kernel).
    out_act: store mp_silu(y * out_scale[b, cout]) instead of y; out2: also store mp_silu(out2_scale * y_final)
    (with out_act=False and out_scale given: out = y and out2 = mp_silu(out2_scale * out_scale[b, cout] * y)).
    swap_src1: src1 is read from the pair-swapped image (b ^ 1): second depth tap of a (2,k,k) conv on a folded stereo pair.
    swap_paired: the input is [src0 | src1 | src0' | src1'] (' = image b ^ 1): both depth taps over a two-source (mp_cat) operand.
    pixelnorm_eps > 0: the stored output is normalize(y, dim=channels) (DDX_EPI_PIXELNORM; LDS-DMA kernel, one group, Cout <= 64).
    Small-M kernel only: out2_chan_scale [B, Cout] makes out2 the LINEAR twin y_final * out2_chan_scale (operand of attn_qk);
    src0_alt (with prologue_rows > 0): output channels below prologue_rows read src0_alt instead of src0.
    residual_up: `residual` is [B, H/2, W/2, Cout] and enters mp_sum nearest-upsampled (skip conv of an up block run at the source size).
    head_norm = 64: every 64 consecutive output channels of a pixel are RMS-normalised in the epilogue (normalize() of the q | k | v vectors of the
    merged attn_qk | attn_v conv; the attention op then takes prenorm=True).  1x1 GEMM kernel / register-staged kernel with 64-multiple tiles.
    Tensors marked with `mark_c16` are channel-blocked [B, C/16, H, W, 16] (same shape attribute, same bytes; 3x3 LDS-DMA kernel only).
    query=True: no launch, returns the kernel code the library would choose (2 register-staged, 3 LDS-DMA, 4 small-M, 1 scalar)."""
    B, sH, sW, C0 = src0.shape
    if out_hw is None:
        out_hw = {L.RESAMPLE_KEEP: (sH, sW), L.RESAMPLE_UP: (sH * 2, sW * 2), L.RESAMPLE_DOWN: (sH // 2, sW // 2)}[resample]
    H, W = out_hw
    C1 = src1.shape[3] if src1 is not None else 0
    if out is None:
        out = torch.empty(B, H, W, pw.Cout, dtype=src0.dtype, device=src0.device)
    if residual is not None:
        want = (B, H // 2, W // 2, pw.Cout) if residual_up else (B, H, W, pw.Cout)
        if tuple(residual.shape) != want:
            raise L.DDXError(f"conv2d: residual has shape {tuple(residual.shape)}, expected {want} (residual_up={residual_up})")
    d = L.ConvDesc(src0=ptr(src0), src1=ptr(src1), chan_scale=ptr(chan_scale), wp=ptr(pw.wp), residual=ptr(residual), out=ptr(out),
                   B=B, H=H, W=W, C0=C0, C1=C1, Cout=pw.Cout, groups=pw.groups, ksize=pw.ksize, CK=pw.CK, resample=resample,
                   prologue=prologue, epilogue=L.EPI_PIXELNORM if pixelnorm_eps > 0 else (L.EPI_MPSUM if residual is not None else L.EPI_STORE),
                   scale0=scale0, scale1=scale1, res_t=pixelnorm_eps if pixelnorm_eps > 0 else res_t, clip=clip, dtype=dtype_code(src0.dtype),
                   force_direct=1 if force_direct else (path if isinstance(path, int) else _PATH_CODE[path]),
                   out_scale=ptr(out_scale), out2=ptr(out2), out_act=int(out_act), out2_scale=float(out2_scale),
                   pad_mode=(L.PAD_REFLECT_W if reflect_w else L.PAD_ZERO) | (L.PAD_SWAP_SRC1 if swap_src1 else 0) | (L.PAD_SWAP_PAIRED if swap_paired else 0),
                   prologue_rows=prologue_rows, out2_linear=int(out2_chan_scale is not None), out2_chan_scale=ptr(out2_chan_scale), src0_alt=ptr(src0_alt),
                   residual_up=int(residual_up), out_head_norm=int(head_norm), out_head_eps=float(head_eps if head_norm else 0.0))
    if query:
        return int(lib().ddx_mpconv2d_path(C.byref(d)))
    d.layout = ((L.LAYOUT_SRC0_C16 if is_c16(src0) else 0) | (L.LAYOUT_SRC1_C16 if is_c16(src1) else 0) |
                (L.LAYOUT_OUT_C16 if is_c16(out) else 0) | (L.LAYOUT_OUT2_C16 if is_c16(out2) else 0))
    if d.force_direct == 0 and d.CK != 16 and not d.layout and (_tuning or _conv_choice):
        sig = _conv_signature(d)
        if _tuning and sig not in _conv_choice:
            _conv_choice[sig] = _tune_conv(d)
        d.force_direct = _conv_choice.get(sig, 0)
    check(lib().ddx_mpconv2d_fwd(C.byref(d), current_stream()), "mpconv2d_fwd")
    return out


def conv_pair_supported(B: int, C: int, groups: int, hidden: int, dtype: torch.dtype) -> bool:
    """Does the fused conv_res0 -> conv_res1 launch (ddx_mpconv_pair_fwd, csrc/conv_pair.hip) serve this block shape?"""
    return bool(lib().ddx_mpconv_pair_supported(B, C, groups, hidden, dtype_code(dtype)))


def conv_pair(src: torch.Tensor, pw0: PreparedWeight, pw1: PreparedWeight, chan_scale: torch.Tensor, residual: torch.Tensor, res_t: float, *,
              clip: float = 0.0, out: Optional[torch.Tensor] = None, out2: Optional[torch.Tensor] = None, out2_scale: float = 1.0) -> torch.Tensor:
    """out = clip(mp_sum(residual, conv_res1(mp_silu(conv_res0(src) * chan_scale)), res_t)) (+ out2 = mp_silu(out2_scale * out)) in ONE launch:
    the hidden tensor stays in LDS (include/ddx_hip.h: ddx_conv_pair_desc; reference unet_edm2_b4.py:121-135).  `src` is the activated input."""
    B, H, W, Cn = src.shape
    assert pw0.ksize == 3 and pw1.ksize == 3 and pw0.groups == pw1.groups and pw1.Cout == Cn and pw0.Cout == pw1.Cg * pw1.groups
    if residual.shape != src.shape or chan_scale.shape != (B, pw0.Cout):
        raise L.DDXError("conv_pair: residual / chan_scale shape")
    if out is None:
        out = torch.empty_like(src)
    d = L.ConvPairDesc(src=ptr(src), wp0=ptr(pw0.wp), wp1=ptr(pw1.wp), chan_scale=ptr(chan_scale), residual=ptr(residual), out=ptr(out), out2=ptr(out2),
                       B=B, H=H, W=W, C=Cn, hidden=pw0.Cout, groups=pw0.groups, CK0=pw0.CK, CK1=pw1.CK, dtype=dtype_code(src.dtype),
                       res_t=float(res_t), clip=float(clip), out2_scale=float(out2_scale))
    check(lib().ddx_mpconv_pair_fwd(C.byref(d), current_stream()), "mpconv_pair_fwd")
    return out


def conv2d_dgrad_act(dy: torch.Tensor, pw_t: PreparedWeight, y0: torch.Tensor, *, y1: Optional[torch.Tensor] = None, scale0: float = 1.0,
                     scale1: float = 1.0, chan_scale: Optional[torch.Tensor] = None, dchan_scale: Optional[torch.Tensor] = None,
                     add: Optional[torch.Tensor] = None, act: bool = True):
    """Data gradient of a conv whose operand was a = mp_silu(y * chan_scale * scale) (act) / y * chan_scale * scale, through the
    activation:  returns (dy0, dy1 | None) = silu_scale_bwd(conv2d(dy, pw_t), y, ...) per channel part (y0 | y1 are the two
    sources of an mp_cat operand), `add` [.., C0 + C1] is added, dchan_scale [B, C] accumulates.  One launch with the activation
    backward in the conv's epilogue when the layer qualifies (LDS-DMA or register-staged kernel; include/ddx_hip.h:
    ddx_mpconv2d_dgrad_act), otherwise the conv followed by ddx_silu_scale_bwd per part."""
    B, H, W, C0 = dy.shape
    Cs = y0.shape[-1]
    split = Cs if y1 is not None else 0
    assert pw_t.Cout == Cs + (y1.shape[-1] if y1 is not None else 0)
    out0 = torch.empty_like(y0)
    out1 = torch.empty_like(y1) if y1 is not None else None
    conv = L.ConvDesc(src0=ptr(dy), src1=None, chan_scale=None, wp=ptr(pw_t.wp), residual=None, out=ptr(out0), B=B, H=H, W=W, C0=C0, C1=0,
                      Cout=pw_t.Cout, groups=pw_t.groups, ksize=pw_t.ksize, CK=pw_t.CK, resample=L.RESAMPLE_KEEP, prologue=L.PRO_NONE,
                      epilogue=L.EPI_STORE, scale0=1.0, scale1=1.0, res_t=0.0, clip=0.0, dtype=dtype_code(dy.dtype), force_direct=0,
                      out_scale=None, out2=None, out_act=0, out2_scale=1.0, pad_mode=L.PAD_ZERO)
    d = L.DgradActDesc(conv=conv, y0=ptr(y0), y1=ptr(y1), out1=ptr(out1), add=ptr(add), chan_scale=ptr(chan_scale), dchan_scale=ptr(dchan_scale),
                       workspace=None, split=split, act=int(act), scale0=float(scale0), scale1=float(scale1))
    nbytes = lib().ddx_mpconv2d_dgrad_act_workspace_bytes(C.byref(d)) if (dy.dtype == torch.bfloat16 and _FUSE_DGRAD_ACT) else 0
    if nbytes:
        global _dgrad_act_fused_calls
        _dgrad_act_fused_calls += 1
        check(lib().ddx_mpconv2d_dgrad_act(C.byref(d), current_stream()), "mpconv2d_dgrad_act")
        return out0, out1
    da = conv2d(dy, pw_t)
    if y1 is None:
        return silu_scale_bwd(da, y0, chan_scale, scale0, dchan_scale, add=add, act=act), None
    assert chan_scale is None and dchan_scale is None
    return (silu_scale_bwd(da[..., :Cs], y0, None, scale0, add=add[..., :Cs] if add is not None else None, act=act),
            silu_scale_bwd(da[..., Cs:], y1, None, scale1, add=add[..., Cs:] if add is not None else None, act=act))


_FUSE_DGRAD_ACT = True       # (tests flip it to compare against conv + silu_scale_bwd)
_dgrad_act_fused_calls = 0      # how often the fused launch was taken (tests check that a qualifying layer really used it)


def conv2d_wgrad(dy: torch.Tensor, x0: torch.Tensor, groups: int, ksize: int, *, x1: Optional[torch.Tensor] = None,
                 resample: int = L.RESAMPLE_KEEP, out: Optional[torch.Tensor] = None, accumulate: bool = False) -> torch.Tensor:
    """Weight gradient of the conv w.r.t. its prepared weight: dy `[B,H,W,Cout]`, operand x0 (| x1) NHWC bf16 ->
    fp32 `[Cout, Cin/groups, k, k]` (see include/ddx_hip.h)."""
    B, H, W, Cout = dy.shape
    C0 = x0.shape[3]
    C1 = x1.shape[3] if x1 is not None else 0
    if out is None:
        out = torch.empty(Cout, (C0 + C1) // groups, ksize, ksize, dtype=torch.float32, device=dy.device)
    d = L.WgradDesc(dy=ptr(dy), x0=ptr(x0), x1=ptr(x1), dw=ptr(out), workspace=None, B=B, H=H, W=W, C0=C0, C1=C1, Cout=Cout,
                    groups=groups, ksize=ksize, resample=resample, dtype=dtype_code(dy.dtype), accumulate=int(accumulate))
    nbytes = lib().ddx_wgrad_workspace_bytes(C.byref(d))
    if nbytes == 0:
        check(-1, "mpconv2d_wgrad (workspace query)")
    ws = torch.empty(nbytes, dtype=torch.uint8, device=dy.device)
    d.workspace = ptr(ws)
    check(lib().ddx_mpconv2d_wgrad(C.byref(d), current_stream()), "mpconv2d_wgrad")
    return out


def conv2d_wgrad_parts(dy: torch.Tensor, x0: torch.Tensor, groups: int, ksize: int, parts: torch.Tensor, *, x1: Optional[torch.Tensor] = None,
                       resample: int = L.RESAMPLE_KEEP) -> int:
    """Weight gradient as split-K partial sums: `parts` [P, Cout, Cin/groups, k, k] fp32 receives the launch's slices in its first
    ddx_wgrad_parts() entries (returned), no reduction launch -- the consumer adds them (training.weight_bank: the weight-path backward
    reads `dwp_parts` slices per row).  Entries beyond the returned count are NOT touched: the caller keeps them zero."""
    B, H, W, Cout = dy.shape
    C0 = x0.shape[3]
    C1 = x1.shape[3] if x1 is not None else 0
    assert parts.dtype == torch.float32 and parts.is_contiguous() and parts.shape[1:] == (Cout, (C0 + C1) // groups, ksize, ksize)
    d = L.WgradDesc(dy=ptr(dy), x0=ptr(x0), x1=ptr(x1), dw=ptr(parts), workspace=ptr(parts), B=B, H=H, W=W, C0=C0, C1=C1, Cout=Cout,
                    groups=groups, ksize=ksize, resample=resample, dtype=dtype_code(dy.dtype), accumulate=2)
    n = int(lib().ddx_wgrad_parts(C.byref(d)))
    if n <= 0 or n > parts.shape[0]:
        raise L.DDXError(f"conv2d_wgrad_parts: the launch writes {n} slices, the buffer holds {parts.shape[0]}")
    check(lib().ddx_mpconv2d_wgrad(C.byref(d), current_stream()), "mpconv2d_wgrad(parts)")
    return n


def _chan_view(t: torch.Tensor, Cn: int):
    """(data pointer, row stride) of an NHWC tensor or of a channel slice `t[..., c0:c0+Cn]` of one."""
    assert t.shape[-1] == Cn and t.stride(-1) == 1
    ld = t.stride(-2)
    assert all(t.stride(i) == t.stride(i + 1) * t.shape[i + 1] for i in range(t.dim() - 2)), "rows must be evenly strided"
    return t.data_ptr(), ld


def silu_scale_bwd(da: torch.Tensor, y: torch.Tensor, chan_scale: Optional[torch.Tensor] = None, scale: float = 1.0,
                   dc: Optional[torch.Tensor] = None, add: Optional[torch.Tensor] = None, act: bool = True) -> torch.Tensor:
    """Backward of a = mp_silu(y * chan_scale[b, c] * scale) on NHWC tensors: returns dy (+ add); accumulates into dc [B, C]
    fp32.  `da` and `add` may be channel slices of wider NHWC tensors (one source of an mp_cat)."""
    B, Cn = y.shape[0], y.shape[-1]
    dy = torch.empty_like(y)
    da_p, da_ld = _chan_view(da, Cn)
    add_p, add_ld = _chan_view(add, Cn) if add is not None else (None, 0)
    check(lib().ddx_silu_scale_bwd_ex(da_p, da_ld, ptr(y), ptr(chan_scale), float(scale), add_p, add_ld, ptr(dy), ptr(dc), B,
                                      y.numel() // (B * Cn), Cn, int(act), dtype_code(y.dtype), current_stream()), "silu_scale_bwd")
    return dy


def add3(a: torch.Tensor, b: torch.Tensor, c: Optional[torch.Tensor] = None) -> torch.Tensor:
    out = torch.empty_like(a)
    check(lib().ddx_add3(ptr(a), ptr(b), ptr(c), ptr(out), a.numel(), dtype_code(a.dtype), current_stream()), "add3")
    return out


def silu_scale_fwd(x: torch.Tensor, chan_scale: Optional[torch.Tensor] = None, scale: float = 1.0, act: bool = True,
                   out: Optional[torch.Tensor] = None) -> torch.Tensor:
    """mp_silu(x * chan_scale[b, c] * scale) on an NHWC tensor (recomputed conv operand of the backward pass)."""
    B, Cn = x.shape[0], x.shape[-1]
    if out is None:
        out = torch.empty_like(x)
    check(lib().ddx_silu_scale_fwd(ptr(x), ptr(chan_scale), float(scale), ptr(out), B, x.numel() // (B * Cn), Cn, int(act), dtype_code(x.dtype),
                                   current_stream()), "silu_scale_fwd")
    return out


def mpsum_clip_bwd(dout: torch.Tensor, out: Optional[torch.Tensor], t: float, clip: float = 0.0, want_dres: bool = True):
    """Backward of out = clip(mp_sum(res, y, t)): returns (dres | None, dy)."""
    dres = torch.empty_like(dout) if want_dres else None
    dy = torch.empty_like(dout)
    check(lib().ddx_mpsum_clip_bwd(ptr(dout), ptr(out), ptr(dres), ptr(dy), float(t), float(clip), dout.numel(), dtype_code(dout.dtype),
                                   current_stream()), "mpsum_clip_bwd")
    return dres, dy


def pixelnorm_bwd(dy: torch.Tensor, x: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    Cn = x.shape[-1]
    dx = torch.empty_like(x)
    check(lib().ddx_pixelnorm_bwd(ptr(dy), ptr(x), ptr(dx), x.numel() // Cn, Cn, eps, dtype_code(x.dtype), current_stream()), "pixelnorm_bwd")
    return dx


def wprep_bwd(pw: PreparedWeight, dwp: torch.Tensor, dw: Optional[torch.Tensor] = None, dgain: Optional[torch.Tensor] = None,
              accumulate: bool = False) -> torch.Tensor:
    """Gradient w.r.t. the master weight (fp32, the weight's shape) from the gradient w.r.t. the prepared weight `dwp`
    (fp32 natural layout, e.g. from conv2d_wgrad); `pw` is the forward preparation (ops.wprep) of that weight."""
    assert dwp.dtype == torch.float32 and dwp.is_contiguous()
    if dw is None:
        dw = torch.empty_like(dwp)
    check(lib().ddx_mpconv_wprep_bwd(C.byref(pw.desc), ptr(dwp), ptr(dw), ptr(dgain), int(accumulate), current_stream()), "wprep_bwd")
    return dw


def linear_small_bwd(dc: torch.Tensor, x: torch.Tensor, weight: torch.Tensor, groups: int = 1, gain_ptr: Optional[torch.Tensor] = None,
                     normalize: bool = False, dx: Optional[torch.Tensor] = None, row_scale: Optional[torch.Tensor] = None,
                     dwp: Optional[torch.Tensor] = None):
    """Backward of c = const + x @ w'^T (w' = weight path of `weight` [O, K/groups]): returns (dw, dgain) for the master weight
    and the gain parameter; accumulates the input gradient into dx [M, K] (fp32) when given.
    With `row_scale` and `dwp` given (training.weight_bank: the row scales were computed for every layer at once and the
    weight-path backward runs later for every layer at once) only the gradient w.r.t. the prepared weight is written, into
    `dwp`, and (None, None) is returned."""
    M, O = dc.shape
    K = x.shape[1]
    w2 = weight.reshape(O, -1)
    deferred = row_scale is not None and dwp is not None
    if deferred:
        rs = row_scale
        dwp = dwp.view(O, w2.shape[1])
    else:
        rs = torch.empty(O, dtype=torch.float32, device=dc.device)
        check(lib().ddx_wprep_rowscale(ptr(w2), dtype_code(w2.dtype), ptr(rs), ptr(gain_ptr), 1.0, O, w2.shape[1], int(normalize), current_stream()),
              "wprep_rowscale")
        dwp = torch.empty(O, w2.shape[1], dtype=torch.float32, device=dc.device)
    check(lib().ddx_linear_small_bwd(ptr(dc), ptr(x), x.stride(0), ptr(w2), dtype_code(w2.dtype), ptr(rs), ptr(dwp), ptr(dx), M, O, K, groups,
                                     current_stream()), "linear_small_bwd")
    if deferred:
        return None, None
    d = L.WPrepDesc(w=ptr(w2), wp=None, gain_ptr=ptr(gain_ptr), gain=1.0, w_dtype=dtype_code(w2.dtype), wp_dtype=L.DDX_F32, Cout=O, Cg=w2.shape[1],
                    ksize=1, groups=groups, CK=32, normalize=int(normalize), qk_head_dim=0, in_split=0, in_scale0=1.0, in_scale1=1.0, transpose=0,
                    row_scale=None)
    dw = torch.empty_like(dwp)
    dgain = torch.zeros(1, dtype=torch.float32, device=dc.device) if gain_ptr is not None else None
    check(lib().ddx_mpconv_wprep_bwd(C.byref(d), ptr(dwp), ptr(dw), ptr(dgain), 0, current_stream()), "wprep_bwd(linear)")
    return dw.reshape(weight.shape), dgain


def edm2_loss(denoised: torch.Tensor, target: torch.Tensor, sigma: torch.Tensor, logvar: Optional[torch.Tensor], sigma_data: float,
              want_grad: bool = True, sigma_data_vec: Optional[torch.Tensor] = None):
    """Per-sample EDM2 loss [B] (+ d mean(loss)/d denoised, d mean(loss)/d logvar) -- unet_trainer.py:271-282; `sigma_data_vec` [B]: the
    per-sample sigma_data of use_dynamic_sigma_data (:263-269)."""
    B = denoised.shape[0]
    n = denoised.numel() // B
    dev = denoised.device
    loss = torch.empty(B, dtype=torch.float32, device=dev)
    dd = torch.empty_like(denoised) if want_grad else None
    dlv = torch.empty(B, dtype=torch.float32, device=dev) if (want_grad and logvar is not None) else None
    ws = torch.empty(B, dtype=torch.float32, device=dev)
    check(lib().ddx_edm2_loss_v(ptr(denoised), ptr(target), ptr(sigma), ptr(logvar), float(sigma_data), ptr(sigma_data_vec), ptr(loss), ptr(dd), ptr(dlv),
                                ptr(ws), B, n, current_stream()), "edm2_loss")
    return loss, dd, dlv


def mp_dropout_(x: torch.Tensor, p: float, seed: int, stream_id: int) -> torch.Tensor:
    """In place: x <- keep ? x / sqrt(1 - p) : 0 with the keep mask Philox(seed, stream_id, element index) -- the block's magnitude-preserving
    dropout (unet_edm2_b4.py:124-125) on the activation, and, called again with the same (seed, stream_id), its backward on the gradient."""
    check(lib().ddx_mp_dropout(ptr(x), x.numel(), float(p), int(seed) & 0xFFFFFFFFFFFFFFFF, int(stream_id) & 0xFFFFFFFF, dtype_code(x.dtype), current_stream()),
          "mp_dropout")
    return x


def unet_xref_mix_bwd(d_out: torch.Tensor, d0: torch.Tensor, x_ref: torch.Tensor, want_dxref: bool = True):
    """Backward of D = mp_sum(x_ref[:, :-1], D0, t = x_ref[:, -1:]) (NCHW fp32): returns (dD0, d x_ref | None)."""
    B, Cn, H, W = d0.shape
    dd0 = torch.empty_like(d0)
    dxr = torch.empty_like(x_ref) if want_dxref else None
    check(lib().ddx_unet_xref_mix_bwd(ptr(d_out), ptr(d0), ptr(x_ref), ptr(dd0), ptr(dxr), B, Cn, H, W, current_stream()), "xref_mix_bwd")
    return dd0, dxr


def cat2_act(a: torch.Tensor, scale_a: float, b: torch.Tensor, scale_b: float):
    """(cat, mp_silu(cat)) with cat = [scale_a * a | scale_b * b] on the channel axis (mp_cat), NHWC, one pass."""
    C0, C1 = a.shape[-1], b.shape[-1]
    out = torch.empty(*a.shape[:-1], C0 + C1, dtype=a.dtype, device=a.device)
    out_act = torch.empty_like(out)
    check(lib().ddx_cat2_act(ptr(a), float(scale_a), ptr(b), float(scale_b), ptr(out), ptr(out_act), a.numel() // C0, C0, C1,
                             dtype_code(a.dtype), current_stream()), "cat2_act")
    return out, out_act


def cat2_swap(a: torch.Tensor, scale_a: float = 1.0, b: Optional[torch.Tensor] = None, scale_b: float = 1.0, want_cat: bool = True):
    """NHWC images ordered n = 2*b + z: returns ([scale_a*a | scale_b*b] or None, the same rows with the stereo pair swapped)."""
    N, C0 = a.shape[0], a.shape[-1]
    C1 = b.shape[-1] if b is not None else 0
    shape = tuple(a.shape[:-1]) + (C0 + C1,)
    out = torch.empty(shape, dtype=a.dtype, device=a.device) if (want_cat and (b is not None or scale_a != 1.0)) else None
    out_sw = torch.empty(shape, dtype=a.dtype, device=a.device)
    check(lib().ddx_cat2_swap(ptr(a), float(scale_a), ptr(b), float(scale_b), ptr(out), ptr(out_sw), N, a.numel() // (N * C0), C0, C1,
                              dtype_code(a.dtype), current_stream()), "cat2_swap")
    return (out if out is not None else a), out_sw


def pixelnorm(x: torch.Tensor, out: Optional[torch.Tensor] = None, eps: float = 1e-4, out_act: Optional[torch.Tensor] = None) -> torch.Tensor:
    """RMS normalisation over the last (channel) axis of contiguous rows; `out_act` also receives mp_silu(result)."""
    Cn = x.shape[-1]
    rows = x.numel() // Cn
    if out is None:
        out = torch.empty_like(x)
    check(lib().ddx_pixelnorm_act_fwd(ptr(x), ptr(out), ptr(out_act), rows, Cn, eps, dtype_code(x.dtype), current_stream()), "pixelnorm_fwd")
    return out


def attention(qk: torch.Tensor, v: torch.Tensor, heads: int, out: Optional[torch.Tensor] = None, eps: float = 1e-4,
              out_scale: Optional[torch.Tensor] = None, prenorm: bool = False) -> torch.Tensor:
    """qk `[B, H, W, 2C]` (head, {q,k}, d), v `[B, H, W, C]` (head, d) -> `[B, H, W, C]`;
    with `out_scale` [B, C] fp32 the stored result is mp_silu(o * out_scale) (operand of attn_proj).
    qk / v may be channel slices of one wider NHWC tensor (a merged attn_qk | attn_v conv output).
    prenorm (bf16): q, k, v were normalised per head by the conv that made them (conv2d(head_norm=...)); the kernel stages them untouched."""
    if prenorm:
        if v.dtype != torch.bfloat16:
            raise L.DDXError("attention: prenorm is a bf16 path")
        eps = -1.0
    B, H, W, Cn = v.shape
    if out is None:
        out = torch.empty(B, H, W, Cn, dtype=v.dtype, device=v.device)
    qk_p, qk_ld = _chan_view(qk, 2 * Cn)
    v_p, v_ld = _chan_view(v, Cn)
    check(lib().ddx_attn_act_fwd_ld(qk_p, qk_ld, v_p, v_ld, ptr(out), ptr(out_scale), B, H * W, heads, Cn // heads, eps, dtype_code(v.dtype),
                                    current_stream()), "attn_fwd")
    return out


def attention_fold(qk: torch.Tensor, v: torch.Tensor, heads: int, out: Optional[torch.Tensor] = None, eps: float = 1e-4,
                   out_scale: Optional[torch.Tensor] = None) -> torch.Tensor:
    """Axis-folded attention on NHWC maps `[N, H, W, .]`: for every (image, column) the H positions attend to each other
    (reference DAE_G1 block, dae_edm2_g1.py:209-228).  qk `[N, H, W, 2C]` (head, {q,k}, d), v `[N, H, W, C]` (head, d) -> `[N, H, W, C]`;
    with `out_scale` [N, C] fp32 the stored result is mp_silu(o * out_scale)."""
    N, H, W, Cn = v.shape
    if out is None:
        out = torch.empty(N, H, W, Cn, dtype=v.dtype, device=v.device)
    qk_p, qk_ld = _chan_view(qk, 2 * Cn)
    v_p, v_ld = _chan_view(v, Cn)
    check(lib().ddx_attn_fold_fwd(qk_p, qk_ld, v_p, v_ld, ptr(out), ptr(out_scale), N, H, W, heads, Cn // heads, eps, dtype_code(v.dtype),
                                  current_stream()), "attn_fold_fwd")
    return out


def make_linear_jobs(jobs: list, device) -> torch.Tensor:
    """Pack [(weight[O,K], gain_ptr|None, out[M,O] fp32, gain, add_const, groups, normalize)] into a device job table."""
    arr = (L.LinearJob * len(jobs))()
    for i, (w, gptr, out, gain, addc, groups, normalize) in enumerate(jobs):
        O = w.shape[0]
        K = w.numel() // O
        arr[i] = L.LinearJob(w=ptr(w), gain_ptr=ptr(gptr), out=ptr(out), gain=float(gain), add_const=float(addc), O=O, K=K,
                             groups=groups, normalize=int(normalize))
    raw = bytes(arr)
    return torch.frombuffer(bytearray(raw), dtype=torch.uint8).to(device)


def linear_small(job_table: torch.Tensor, njobs: int, max_O: int, x: torch.Tensor, M: int, w_dtype: torch.dtype,
                 x_stride: Optional[int] = None) -> None:
    check(lib().ddx_linear_small_batched(ptr(job_table), njobs, max_O, ptr(x), x_stride if x_stride is not None else x.shape[-1], M,
                                         dtype_code(w_dtype), current_stream()), "linear_small_batched")


def mpfourier(x: torch.Tensor, freqs: torch.Tensor, phases: torch.Tensor, out: torch.Tensor, log_sigma_quarter: bool) -> None:
    check(lib().ddx_mpfourier(ptr(x), ptr(freqs), ptr(phases), ptr(out), x.numel(), freqs.numel(), int(log_sigma_quarter),
                              current_stream()), "mpfourier")


def mpsum_rows(a: torch.Tensor, b: torch.Tensor, out: torch.Tensor, *, t: float = 0.5, t_rows: Optional[torch.Tensor] = None,
               silu: bool = False) -> None:
    M, Cn = b.shape
    check(lib().ddx_mpsum_rows(ptr(a), a.shape[0] if a.ndim == 2 else 1, ptr(b), ptr(t_rows), float(t), ptr(out), M, Cn, int(silu),
                               current_stream()), "mpsum_rows")


def unet_input_prep(x_nchw: torch.Tensor, sigma: torch.Tensor, ln_freq_h: torch.Tensor, out_nhwc: torch.Tensor, sigma_data: float) -> None:
    B, Cn, H, W = x_nchw.shape
    check(lib().ddx_unet_input_prep(ptr(x_nchw), ptr(sigma), ptr(ln_freq_h), ptr(out_nhwc), B, Cn, H, W, out_nhwc.shape[3],
                                    sigma_data, dtype_code(out_nhwc.dtype), current_stream()), "unet_input_prep")


def unet_output_combine(y_nhwc: torch.Tensor, x_in: torch.Tensor, sigma: torch.Tensor, x_ref: Optional[torch.Tensor],
                        out: torch.Tensor, sigma_data: float) -> None:
    B, Cn, H, W = x_in.shape
    check(lib().ddx_unet_output_combine(ptr(y_nhwc), ptr(x_in), ptr(sigma), ptr(x_ref), ptr(out), B, Cn, H, W, sigma_data,
                                        dtype_code(y_nhwc.dtype), current_stream()), "unet_output_combine")


def resample2d(x: torch.Tensor, out: torch.Tensor, mode: int) -> torch.Tensor:
    """2x nearest upsample (mode UP) / 2x2 average pool (mode DOWN) of an NHWC tensor into `out`."""
    B, H, W, Cn = out.shape
    check(lib().ddx_resample2d(ptr(x), ptr(out), B, H, W, Cn, mode, dtype_code(x.dtype), current_stream()), "resample2d")
    return out


def lincomb3(out: torch.Tensor, x: torch.Tensor, a: float, y: Optional[torch.Tensor] = None, b: float = 0.0,
             z: Optional[torch.Tensor] = None, c: float = 0.0) -> torch.Tensor:
    """out = a*x + b*y + c*z (fp32, contiguous, same numel)."""
    assert out.dtype == torch.float32 and x.dtype == torch.float32 and out.is_contiguous() and x.is_contiguous()
    check(lib().ddx_lincomb3(ptr(x), float(a), ptr(y), float(b), ptr(z), float(c), ptr(out), out.numel(), current_stream()), "lincomb3")
    return out


def sampler_load(sample: torch.Tensor, x_in: torch.Tensor, x_pre: Optional[torch.Tensor], sigma_out: torch.Tensor, sig_table: torch.Tensor,
                 step: torch.Tensor, which: int) -> None:
    """x_in[c] = x_pre[c] = sample for the nb / B batch copies and sigma_out = sig_table[step][which] (device step counter)."""
    B, nb = sample.shape[0], x_in.shape[0]
    assert sample.is_contiguous() and x_in.is_contiguous() and sig_table.shape[1:] == (2, nb) and step.dtype == torch.int32
    check(lib().ddx_sampler_load(ptr(sample), ptr(x_in), ptr(x_pre), ptr(sigma_out), ptr(sig_table), ptr(step), which, B, nb, sample.numel(),
                                 current_stream()), "sampler_load")


def lincomb3_dev(out: torch.Tensor, coef: torch.Tensor, step: torch.Tensor, x: torch.Tensor, ia: int, y: Optional[torch.Tensor] = None, ib: int = -1,
                 z: Optional[torch.Tensor] = None, ic: int = -1, z_step_stride: int = 0) -> torch.Tensor:
    """out = a*x + b*y + c*z with (a, b, c) = coef[step, (ia, ib, ic)] read on the device; z advanced by step * z_step_stride elements."""
    assert out.dtype == torch.float32 and coef.dtype == torch.float32 and coef.ndim == 2 and step.dtype == torch.int32
    check(lib().ddx_lincomb3_dev(ptr(x), ptr(y), ptr(z), ptr(out), out.numel(), ptr(coef), ptr(step), coef.shape[1], ia, ib, ic, z_step_stride,
                                 current_stream()), "lincomb3_dev")
    return out


def step_advance(step: torch.Tensor) -> None:
    check(lib().ddx_step_advance(ptr(step), current_stream()), "step_advance")


def nchw_to_nhwc(x: torch.Tensor, dtype: torch.dtype, out: Optional[torch.Tensor] = None) -> torch.Tensor:
    B, Cn, H, W = x.shape
    if out is None:
        out = torch.empty(B, H, W, Cn, dtype=dtype, device=x.device)
    check(lib().ddx_nchw_to_nhwc(ptr(x), ptr(out), B, Cn, H, W, dtype_code(dtype), current_stream()), "nchw_to_nhwc")
    return out


def nhwc_to_nchw(x: torch.Tensor, out: Optional[torch.Tensor] = None, channels: Optional[int] = None) -> torch.Tensor:
    """`channels`: take only the first channels of a wider (row-padded) NHWC tensor."""
    B, H, W, ld = x.shape
    Cn = channels or ld
    if out is None:
        out = torch.empty(B, Cn, H, W, dtype=torch.float32, device=x.device)
    check(lib().ddx_nhwc_to_nchw_ld(ptr(x), ld, ptr(out), B, Cn, H, W, dtype_code(x.dtype), current_stream()), "nhwc_to_nchw")
    return out


def stereo_to_images(x: torch.Tensor, channels: int, cpad: int, add_const: bool, dtype: torch.dtype) -> torch.Tensor:
    """NCHW fp32 [B, channels * 2, H, W] (channel = c * 2 + z) -> NHWC images n = 2 b + z, [2B, H, W, cpad] (data | 1 | zeros)."""
    B, CZ, H, W = x.shape
    Z = CZ // channels
    out = torch.empty(B * Z, H, W, cpad, dtype=dtype, device=x.device)
    check(lib().ddx_stereo_to_images(ptr(x), ptr(out), B, channels, Z, H, W, cpad, int(add_const), dtype_code(dtype), current_stream()), "stereo_to_images")
    return out


def images_to_stereo(x: torch.Tensor, channels: int, Z: int = 2) -> torch.Tensor:
    """NHWC images n = Z b + z `[Z B, H, W, ld]` -> NCHW fp32 [B, channels * Z, H, W] (channel = c * Z + z), first `channels` of ld."""
    N, H, W, ld = x.shape
    out = torch.empty(N // Z, channels * Z, H, W, dtype=torch.float32, device=x.device)
    check(lib().ddx_images_to_stereo(ptr(x), ld, ptr(out), N // Z, channels, Z, H, W, dtype_code(x.dtype), current_stream()), "images_to_stereo")
    return out
