"""Optimizer side of the training step on the HIP kernels: global-norm clipping with the reference's dynamic clip threshold,
fused AdamW + EMA over all parameters in two launches, the `edm2` learning-rate schedule.

Mirrors reference src/training/trainer.py: OptimizerConfig / LRScheduleConfig (:99-126), get_max_grad_norm /
update_grad_norm_stats (:407-431), clip_grad_norm_ + optimizer.step (:1027-1063), torch.optim.AdamW (:461-472), lr_schedule
`edm2` (:653-663).  Gradients arrive as a {name: tensor} dict (UNetTrainer.train_batch); loss_scale is applied here, as the
reference applies it to the loss before backward (:1016).
"""
from __future__ import annotations

import ctypes as C
import math
from dataclasses import dataclass
from typing import Optional

import torch

from .. import _lib as L
from .._lib import check, current_stream, lib, ptr


@dataclass
class LRScheduleConfig:
    lr_schedule: str = "edm2"
    learning_rate: float = 3e-3
    lr_warmup_steps: int = 5000
    lr_reference_steps: int = 70000
    lr_decay_exponent: float = 0.5
    min_learning_rate: float = 1e-4


@dataclass
class OptimizerConfig:
    adam_beta1: float = 0.9
    adam_beta2: float = 0.99
    adam_epsilon: float = 1e-8
    adam_weight_decay: float = 0.
    loss_scale: float = 250.
    max_grad_norm: float = 1.
    grad_norm_std_ema_beta: float = 0.999
    grad_norm_mean_ema_beta: float = 0.99
    dynamic_max_grad_norm_z: Optional[float] = 3


def lr_multiplier(cfg: LRScheduleConfig, step: int, warmup_steps: Optional[int] = None, reference_steps: Optional[int] = None) -> float:
    """trainer.py:653-663 (`edm2`) / :665-669 (`constant`); warmup / reference steps may be passed pre-scaled (:628-634)."""
    wu = cfg.lr_warmup_steps if warmup_steps is None else warmup_steps
    ref = cfg.lr_reference_steps if reference_steps is None else reference_steps
    if cfg.lr_schedule == "constant":
        return step / wu if step < wu else 1.0
    lr = 1.0
    if step < wu:
        lr *= step / wu
    if step > ref:
        lr /= (step / ref) ** cfg.lr_decay_exponent
        lr = max(lr * cfg.learning_rate, cfg.min_learning_rate) / cfg.learning_rate
    return lr


def std_to_exp(std: float) -> float:
    """EDM2 power-function EMA: exponent gamma of a profile with relative standard deviation `std` -- the real root of
    gamma^3 + 7 gamma^2 + (16 - 1/std^2) gamma + (12 - 1/std^2) = 0 (Karras et al. 2024, Algorithm 2; reference ema.py:95-107)."""
    import numpy as np
    tmp = np.float64(std) ** -2
    roots = np.roots([1.0, 7.0, 16.0 - tmp, 12.0 - tmp])
    return float(roots.real.max())


def power_function_beta(std: float, t_next: int, t_delta: int) -> float:
    """reference ema.py:112-114."""
    return float((1 - t_delta / t_next) ** (std_to_exp(std) + 1))


@dataclass
class EMASpec:
    """One EMA of the training weights (reference ema.py EMA_Config :197-206): classic (`beta`) or power-function (`std`), optional
    warm-up of beta and optional feedback of the EMA into the training weights."""
    name: str
    tensors: dict                               # parameter name -> fp32 device tensor (the EMA weights)
    beta: Optional[float] = None
    std: Optional[float] = None
    num_warmup_steps: Optional[int] = None
    feedback_beta: Optional[float] = None

    def __post_init__(self) -> None:
        if (self.beta is None) == (self.std is None):
            raise ValueError(f"EMA {self.name}: give exactly one of beta / std")

    def effective_beta(self, global_step: int, total_samples_processed: int, total_batch_size: int) -> float:
        """ema.py:297-301 (the update runs after the step's samples were counted)."""
        beta = self.beta if self.beta is not None else power_function_beta(self.std, total_samples_processed + total_batch_size, total_batch_size)
        if self.num_warmup_steps:
            beta *= min(global_step / self.num_warmup_steps, 1)
        return beta


class FusedAdamW:
    """AdamW over a fixed set of fp32 device tensors.  `step(grads)` = clip_grad_norm_(max_grad_norm) + optimizer.step() (+ EMA).
    With `emas` (list of EMASpec, at most 4) and / or `wn_rows` ({parameter name: output rows} of the weight-normalised tensors) the
    update runs as ONE launch that also steps every EMA in order (with feedback) and re-normalises the rows afterwards
    (ddx_multi_adamw_ema_wn; reference trainer.py:1063-1108 + ema.py:284-321)."""

    def __init__(self, params: dict, cfg: OptimizerConfig = OptimizerConfig(), ema: Optional[dict] = None, ema_beta: float = 0.0,
                 emas: Optional[list] = None, wn_rows: Optional[dict] = None) -> None:
        for k, p in params.items():
            if p.dtype != torch.float32 or p.device.type != "cuda" or not p.is_contiguous():
                raise L.DDXError(f"FusedAdamW: parameter {k} must be a contiguous float32 tensor on the ROCm device")
        self.params, self.cfg, self.ema, self.ema_beta = params, cfg, ema, ema_beta
        self.emas, self.wn_rows = list(emas or []), wn_rows
        if len(self.emas) > L.MAX_EMAS:
            raise L.DDXError(f"FusedAdamW: at most {L.MAX_EMAS} EMAs per launch")
        if self.emas and ema is not None:
            raise L.DDXError("FusedAdamW: give either `ema` (one fixed-beta EMA) or `emas`")
        for j, e in enumerate(self.emas):      # the kernel gets raw pointers: every EMA must shadow every parameter exactly
            missing = [k for k in params if k not in e.tensors]
            if missing:
                raise L.DDXError(f"FusedAdamW: EMA {j} has no tensor for {missing[:3]} (a missing entry would silently get no update)")
            for k, p_ in params.items():
                t = e.tensors[k]
                if t.dtype != torch.float32 or not t.is_contiguous() or t.shape != p_.shape or t.device != p_.device:
                    raise L.DDXError(f"FusedAdamW: EMA {j} tensor of {k} must be contiguous float32 with the parameter's shape on its device")
        self.fused = bool(self.emas) or wn_rows is not None
        self.m = {k: torch.zeros_like(p) for k, p in params.items()}
        self.v = {k: torch.zeros_like(p) for k, p in params.items()}
        self.steps = 0
        self.grad_norm_logmean = float(math.log(cfg.max_grad_norm))    # trainer.py:227-228
        self.grad_norm_logvar = self.grad_norm_logmean
        dev = next(iter(params.values())).device
        self._ws = torch.zeros(3, dtype=torch.float32, device=dev)
        self._names = list(params)
        self._max_n = max(p.numel() for p in params.values())

    # ---- dynamic clip threshold (trainer.py:407-431)
    def get_max_grad_norm(self) -> float:
        c = self.cfg
        if c.dynamic_max_grad_norm_z is None:
            return c.max_grad_norm
        return math.exp(self.grad_norm_logmean) + math.exp(self.grad_norm_logvar / 2) * c.dynamic_max_grad_norm_z

    def update_grad_norm_stats(self, grad_norm: float, eps: float = 1e-8) -> None:
        c = self.cfg
        grad_norm = max(grad_norm, eps)
        grad_var = max((grad_norm - math.exp(self.grad_norm_logmean)) ** 2, eps)
        self.grad_norm_logmean = self.grad_norm_logmean * c.grad_norm_mean_ema_beta + (1 - c.grad_norm_mean_ema_beta) * math.log(grad_norm)
        self.grad_norm_logvar = self.grad_norm_logvar * c.grad_norm_std_ema_beta + (1 - c.grad_norm_std_ema_beta) * math.log(grad_var)

    def _table_ex(self, grads: dict) -> torch.Tensor:
        key = tuple(grads[k].data_ptr() for k in self._names)
        if getattr(self, "_table_ex_key", None) == key:
            return self._table_ex_dev
        arr = (L.OptimJobEx * len(self._names))()
        self._max_rows = 1
        for i, k in enumerate(self._names):
            g, p = grads[k], self.params[k]
            rows = int((self.wn_rows or {}).get(k, 0))
            norm = int(rows > 0)
            rows = rows if rows > 0 else (p.shape[0] if p.ndim >= 2 else 1)
            self._max_rows = max(self._max_rows, rows)
            emap = (C.c_void_p * L.MAX_EMAS)(*[ptr(e.tensors[k]) if (j < len(self.emas) and k in (e := self.emas[j]).tensors) else None
                                               for j in range(L.MAX_EMAS)])
            arr[i] = L.OptimJobEx(p=ptr(p), g=ptr(g), m=ptr(self.m[k]), v=ptr(self.v[k]), ema=emap, n=g.numel(), rows=rows, normalize=norm, reserved=0)
        self._table_ex_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self._ws.device)
        self._table_ex_key = key
        return self._table_ex_dev

    def _table(self, grads: dict) -> torch.Tensor:
        key = tuple(grads[k].data_ptr() for k in self._names)      # the trainer's gradients live in one persistent bucket:
        if getattr(self, "_table_key", None) == key:               # the table is built once, no host->device copy per step
            return self._table_dev
        arr = (L.OptimJob * len(self._names))()
        for i, k in enumerate(self._names):
            g = grads[k]
            if g.dtype != torch.float32 or not g.is_contiguous() or g.numel() != self.params[k].numel():
                raise L.DDXError(f"FusedAdamW: gradient of {k} must be contiguous float32 with the parameter's size")
            arr[i] = L.OptimJob(p=ptr(self.params[k]), g=ptr(g), m=ptr(self.m[k]), v=ptr(self.v[k]),
                                ema=ptr(self.ema[k]) if self.ema is not None else None, n=g.numel())
        self._table_dev = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(self._ws.device)
        self._table_key = key
        return self._table_dev

    def step(self, grads: dict, lr: float, grad_scale: Optional[float] = None, ema_betas: Optional[list] = None) -> float:
        """grads: d mean(loss) / d parameter (summed over ranks when distributed; pass grad_scale = loss_scale / world_size).
        ema_betas: this step's beta of every EMASpec (EMASpec.effective_beta), in order.
        Returns the (scaled) global gradient norm before clipping, like accelerator.clip_grad_norm_."""
        c = self.cfg
        gs = c.loss_scale if grad_scale is None else grad_scale
        max_norm = self.get_max_grad_norm()
        table = self._table(grads)
        n = len(self._names)
        check(lib().ddx_multi_grad_norm(ptr(table), n, self._max_n, gs, max_norm, ptr(self._ws), current_stream()), "multi_grad_norm")
        self.steps += 1
        if self.fused:
            tex = self._table_ex(grads)
            ne = len(self.emas)
            if ne and (ema_betas is None or len(ema_betas) != ne):
                raise L.DDXError("FusedAdamW.step: ema_betas must give one beta per EMA")
            betas = (C.c_float * max(ne, 1))(*([float(b) for b in ema_betas] if ne else [1.0]))
            fbs = (C.c_float * max(ne, 1))(*([float(e.feedback_beta) if e.feedback_beta is not None else -1.0 for e in self.emas] if ne else [-1.0]))
            check(lib().ddx_multi_adamw_ema_wn(ptr(tex), n, self._max_rows, self._ws.data_ptr() + 4, gs, lr, c.adam_beta1, c.adam_beta2,
                                               c.adam_epsilon, c.adam_weight_decay, self.steps, ne, betas, fbs, 1e-4, current_stream()),
                  "multi_adamw_ema_wn")
        else:
            check(lib().ddx_multi_adamw(ptr(table), n, self._max_n, self._ws.data_ptr() + 4, gs, lr, c.adam_beta1, c.adam_beta2, c.adam_epsilon,
                                        c.adam_weight_decay, self.steps, self.ema_beta, current_stream()), "multi_adamw")
        L.bump_weights_epoch()                   # raw-pointer parameter write
        grad_norm = float(self._ws[2])           # one host sync per step, as the reference's .item()
        if not math.isfinite(grad_norm):
            # the kernel skipped the update (non-finite clip coefficient): parameters, moments and EMA are untouched.
            # The reference warns on inf and aborts on NaN before optimizer.step() (trainer.py:1053-1060).
            self.steps -= 1
            if math.isnan(grad_norm):
                raise FloatingPointError("gradient norm is NaN: optimizer step skipped (reference trainer aborts here, trainer.py:1055-1060)")
            return grad_norm
        self.update_grad_norm_stats(grad_norm)
        return grad_norm
