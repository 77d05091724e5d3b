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


class FusedAdamW:
    """AdamW over a fixed set of fp32 device tensors.  `step(grads)` = clip_grad_norm_(max_grad_norm) + optimizer.step() (+ EMA)."""

    def __init__(self, params: dict, cfg: OptimizerConfig = OptimizerConfig(), ema: Optional[dict] = None, ema_beta: float = 0.0) -> None:
        for k, p in params.items():
            if p.dtype != torch.float32 or p.device.type != "cuda" or not p.is_contiguous():
                raise L.DDXError(f"FusedAdamW: parameter {k} must be a contiguous float32 tensor on the ROCm device")
        self.params, self.cfg, self.ema, self.ema_beta = params, cfg, ema, ema_beta
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

    def step(self, grads: dict, lr: float, grad_scale: Optional[float] = None) -> float:
        """grads: d mean(loss) / d parameter (summed over ranks when distributed; pass grad_scale = loss_scale / world_size).
        Returns the (scaled) global gradient norm before clipping, like accelerator.clip_grad_norm_."""
        c = self.cfg
        gs = c.loss_scale if grad_scale is None else grad_scale
        max_norm = self.get_max_grad_norm()
        table = self._table(grads)
        n = len(self._names)
        check(lib().ddx_multi_grad_norm(ptr(table), n, self._max_n, gs, max_norm, ptr(self._ws), current_stream()), "multi_grad_norm")
        self.steps += 1
        check(lib().ddx_multi_adamw(ptr(table), n, self._max_n, self._ws.data_ptr() + 4, gs, lr, c.adam_beta1, c.adam_beta2, c.adam_epsilon,
                                    c.adam_weight_decay, self.steps, self.ema_beta, current_stream()), "multi_adamw")
        grad_norm = float(self._ws[2])           # one host sync per step, as the reference's .item()
        self.update_grad_norm_stats(grad_norm)
        return grad_norm
