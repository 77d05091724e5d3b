"""Weight bank: every MPConv weight of a module prepared / back-propagated / normalised in ONE launch per phase.

The weight branch of reference MPConv.forward (src/modules/mp_tools.py:359-364: forced normalisation in training, the
1/sqrt(fan_in) * gain scale, the cast to the activation dtype) and MPConv.normalize_weights (:375-378) run per layer in the
reference (autograd).  Per optimizer step the default UNet needs them for 151 convs + 69 linear layers -- about 1100 launches
of a few microseconds of payload each.  The bank keeps all device buffers of that path persistent (prepared weights for the
forward and for the data gradient, row scales, the natural-layout gradient w.r.t. the prepared weights) and drives
`ddx_wpath_multi` over one job table:
    prepare()    PREP + ROWSCALE + TRANSPOSED       before the forward
    backward()   BWD                                after all weight-gradient GEMMs wrote `dwp[name]`
    normalize()  NORMALIZE                          after the optimizer step (trainer.py:375-381)
"""
from __future__ import annotations

import ctypes as C
from dataclasses import dataclass
from typing import Optional

import torch

from .. import _lib as L
from .. import ops
from .._lib import check, current_stream, dtype_code, lib, ptr


@dataclass
class BankEntry:
    name: str                                   # state-dict key of the weight
    weight: torch.Tensor                        # fp32 master weight on the device, [Cout, Cg, k, k] or [O, K / groups]
    groups: int = 1
    gain: Optional[torch.Tensor] = None         # learnable gain parameter (0-d / [1] fp32) multiplied into the weight
    gain_name: Optional[str] = None             # its state-dict key (receives dgain)
    qk_head_dim: int = 0
    in_split: int = 0                           # mp_cat scales folded into a linear consumer (decoder conv_skip)
    in_scale0: float = 1.0
    in_scale1: float = 1.0
    npix: int = 0                               # B*H*W the conv runs at (chunk-width choice); 0 = unknown
    prep: bool = True                           # forward prepared buffer   (False: linear layers, used from the master weight)
    transpose: bool = True                      # data-gradient prepared buffer
    grad: bool = True                           # takes part in backward()
    normalize: bool = True                      # forced weight normalisation (False: MPConv.disable_weight_norm)


def _align(n: int, a: int = 256) -> int:
    return (n + a - 1) // a * a


class WeightBank:

    def __init__(self, entries: list, dtype: torch.dtype, grad_views: dict, early: Optional[set] = None) -> None:
        """grad_views: {parameter name: fp32 view} the master-weight gradients (and gain gradients) are written to.
        early: names of the entries whose gradients are complete first (the decoder, back-propagated before the encoder):
        backward("early") / backward("late") run the weight-path backward for the two parts separately, so that the first
        gradient bucket can be all-reduced while the rest of the backward pass runs."""
        self.entries, self.dtype = entries, dtype
        self.split_parts = dtype == torch.bfloat16      # (the fp32 parity GEMM writes finished gradients)
        self.dwp_parts: dict = {}                       # conv layers: [parts, *weight.shape] views of the split-K slices
        self._parts_used: dict = {}                     # slices the last launch of a layer wrote (stale ones are zeroed when it shrinks)
        dev = entries[0].weight.device
        self.dev = dev
        geo = []
        wp_bytes = wpt_bytes = rs_floats = dwp_floats = 0
        for e in entries:
            w = e.weight
            if w.dtype != torch.float32 or not w.is_contiguous() or w.device != dev:
                raise L.DDXError(f"WeightBank: {e.name} must be a contiguous float32 device tensor")
            Cout, Cg = w.shape[0], w.shape[1]
            ks = w.shape[2] if w.ndim == 4 else 1
            Ng, Cin = Cout // e.groups, Cg * e.groups
            CK = ops.pick_ck(Cg, ks, dtype, e.npix) if e.prep else 32
            CKt = ops.pick_ck(Ng, ks, dtype, e.npix) if e.transpose else 32
            nb = _align(lib().ddx_wprep_bytes(Cout, Cg, ks, e.groups, CK, dtype_code(dtype))) if e.prep else 0
            nbt = _align(lib().ddx_wprep_bytes(Cin, Ng, ks, e.groups, CKt, dtype_code(dtype))) if e.transpose else 0
            geo.append(dict(Cout=Cout, Cg=Cg, ks=ks, Ng=Ng, Cin=Cin, CK=CK, CKt=CKt, wp_off=wp_bytes, wpt_off=wpt_bytes, rs_off=rs_floats,
                            dwp_off=dwp_floats, nb=nb, nbt=nbt))
            wp_bytes += nb
            wpt_bytes += nbt
            rs_floats += _align(Cout, 64) if (e.transpose or e.grad) else 0
            # conv layers: the weight-gradient GEMM leaves its split-K slices here and the weight-path backward adds them while it reads
            # the row (no reduction launch per layer: 106 launches / ~1 ms per B=8 step); the bound only depends on the weight's shape
            parts = max(1, int(lib().ddx_wgrad_parts_max(Cout, Cg, e.groups, ks))) if (e.grad and e.prep and self.split_parts) else 1
            geo[-1]["parts"] = parts
            dwp_floats += _align(w.numel() * parts, 64) if e.grad else 0
        # zero-initialised: the padding rows / channels of the prepared layouts are never written afterwards
        self.wp_flat = torch.zeros(max(wp_bytes, 1), dtype=torch.uint8, device=dev)
        self.wpt_flat = torch.zeros(max(wpt_bytes, 1), dtype=torch.uint8, device=dev)
        self.rs_flat = torch.zeros(max(rs_floats, 1), dtype=torch.float32, device=dev)
        self.dwp_flat = torch.zeros(max(dwp_floats, 1), dtype=torch.float32, device=dev)
        self.pw, self.pwt, self.rs, self.dwp, self.dw, self.dgain = {}, {}, {}, {}, {}, {}
        jobs = (L.WPathJob * len(entries))()
        rows = {ph: [0] for ph in range(5)}
        self._keep = []
        for i, (e, g) in enumerate(zip(entries, geo)):
            w = e.weight
            if e.prep:
                buf = self.wp_flat[g["wp_off"]:g["wp_off"] + g["nb"]]
                self.pw[e.name] = ops.PreparedWeight(buf, g["Cout"], g["Cg"], g["ks"], e.groups, g["CK"], dtype, None)
            if e.transpose:
                buft = self.wpt_flat[g["wpt_off"]:g["wpt_off"] + g["nbt"]]
                self.pwt[e.name] = ops.PreparedWeight(buft, g["Cin"], g["Ng"], g["ks"], e.groups, g["CKt"], dtype, None)
            if e.transpose or e.grad:
                self.rs[e.name] = self.rs_flat[g["rs_off"]:g["rs_off"] + g["Cout"]]
            gain_ptr = e.gain.reshape(1) if e.gain is not None else None
            if e.grad:
                self.dwp[e.name] = self.dwp_flat[g["dwp_off"]:g["dwp_off"] + w.numel()].view(w.shape)
                if g["parts"] > 1:
                    self.dwp_parts[e.name] = self.dwp_flat[g["dwp_off"]:g["dwp_off"] + w.numel() * g["parts"]].view((g["parts"],) + tuple(w.shape))
                self.dw[e.name] = grad_views[e.name]
                if self.dw[e.name].numel() != w.numel() or self.dw[e.name].dtype != torch.float32:
                    raise L.DDXError(f"WeightBank: gradient view of {e.name} does not match the weight")
                if e.gain is not None:
                    self.dgain[e.name] = grad_views[e.gain_name].reshape(1)
            self._keep.append((w, gain_ptr))
            jobs[i] = L.WPathJob(w=ptr(w), wp=ptr(self.pw[e.name].wp) if e.prep else None,
                                 wp_t=ptr(self.pwt[e.name].wp) if e.transpose else None,
                                 row_scale=ptr(self.rs[e.name]) if e.name in self.rs else None, gain_ptr=ptr(gain_ptr),
                                 dwp=ptr(self.dwp[e.name]) if e.grad else None, dw=ptr(self.dw[e.name]) if e.grad else None,
                                 dgain=ptr(self.dgain[e.name]) if e.name in self.dgain else None, gain=1.0,
                                 Cout=g["Cout"], Cg=g["Cg"], ksize=g["ks"], groups=e.groups, CK=g["CK"], CK_t=g["CKt"],
                                 normalize=int(e.normalize), qk_head_dim=e.qk_head_dim, in_split=e.in_split, in_scale0=e.in_scale0,
                                 in_scale1=e.in_scale1, dwp_parts=g["parts"], reserved=0)
            part = {L.WPATH_NORMALIZE: g["Cout"] if e.normalize else 0, L.WPATH_PREP: g["Cout"] if e.prep else 0,
                    # (PREP writes the row scales of the entries it prepares: the ROWSCALE phase only serves the others)
                    L.WPATH_ROWSCALE: g["Cout"] if (e.name in self.rs and not e.prep) else 0, L.WPATH_TRANSPOSED: g["Cin"] if e.transpose else 0,
                    L.WPATH_BWD: g["Cout"] if e.grad else 0}
            for ph in range(5):
                rows[ph].append(rows[ph][-1] + part[ph])
        self.njobs = len(entries)
        self.jobs = torch.frombuffer(bytearray(bytes(jobs)), dtype=torch.uint8).to(dev)
        self.prefix = {ph: torch.tensor(rows[ph], dtype=torch.int32).to(dev) for ph in range(5)}
        self.total = {ph: rows[ph][-1] for ph in range(5)}
        # BWD restricted to one part: the same job table with zero rows for the other part's entries
        self.part_prefix, self.part_total = {}, {}
        if early is not None:
            for part, want in (("early", True), ("late", False)):
                r = [0]
                for e, g in zip(entries, geo):
                    r.append(r[-1] + (g["Cout"] if (e.grad and ((e.name in early) == want)) else 0))
                self.part_prefix[part] = torch.tensor(r, dtype=torch.int32).to(dev)
                self.part_total[part] = r[-1]

    def wgrad(self, name: str, dy: torch.Tensor, x0: torch.Tensor, groups: int, ksize: int, x1: Optional[torch.Tensor] = None) -> None:
        """Weight-gradient GEMM of conv layer `name` into the bank: split-K slices for backward() to add, or the finished gradient."""
        parts = self.dwp_parts.get(name)
        if parts is None:
            ops.conv2d_wgrad(dy, x0, groups, ksize, x1=x1, out=self.dwp[name])
            return
        n = ops.conv2d_wgrad_parts(dy, x0, groups, ksize, parts, x1=x1)
        prev = self._parts_used.get(name, 0)
        if prev > n:            # a smaller batch / image than last time: the slices it no longer writes must read as zero
            parts[n:prev].zero_()
        self._parts_used[name] = n

    def _run(self, phase: int) -> None:
        check(lib().ddx_wpath_multi(ptr(self.jobs), ptr(self.prefix[phase]), self.njobs, self.total[phase], phase, dtype_code(self.dtype),
                                    current_stream()), "wpath_multi")

    def prepare(self) -> None:
        """Forward + data-gradient prepared weights and row scales of every entry from the current master weights."""
        self._run(L.WPATH_PREP)
        self._run(L.WPATH_ROWSCALE)
        self._run(L.WPATH_TRANSPOSED)

    def backward(self, part: Optional[str] = None) -> None:
        """dw[name] (and dgain) from dwp[name] for every entry (part None) or for the "early" / "late" entries only; the
        gain-gradient slots must be zero on entry."""
        if part is None:
            self._run(L.WPATH_BWD)
        else:
            check(lib().ddx_wpath_multi(ptr(self.jobs), ptr(self.part_prefix[part]), self.njobs, self.part_total[part], L.WPATH_BWD,
                                        dtype_code(self.dtype), current_stream()), "wpath_multi")

    def normalize(self) -> None:
        self._run(L.WPATH_NORMALIZE)
        L.bump_weights_epoch()       # raw-pointer write to the master weights: eval-mode prepared-weight caches must refresh
