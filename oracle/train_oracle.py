"""CPU restatement of the post-backward parameter pass of the reference trainer -- TEST INFRASTRUCTURE (only tests/, bench.py's
cpu_baseline leg and __graft_entry__.smoke() may import this package; the product path never does).

What it follows:
  * dynamic clip + AdamW: src/training/trainer.py:1027-1063 (accelerator.clip_grad_norm_, torch.optim.AdamW :461-472)
  * EMA manager update: src/training/ema.py:284-321 (EMAs in configuration order; ema <- lerp(ema, p, 1 - beta);
    feedback: p <- lerp(p, ema, 1 - feedback_beta)); beta of a power-function EMA: ema.py:112-114 with std_to_exp (:95-107,
    Karras et al. 2024 "Analyzing and Improving the Training Dynamics of Diffusion Models", Algorithm 2)
  * forced weight normalisation: src/modules/mp_tools.py:375-378 over trainer.py:1105-1108
Pinned by tests/golden/ema_step.safetensors (tools/make_golden.py gen_ema runs the reference's EMA_Manager + torch.optim.AdamW).
"""
from __future__ import annotations

import math
from typing import Optional

import numpy as np
import torch


def std_to_exp(std: float) -> float:
    tmp = np.float64(std) ** -2
    return float(np.roots([1.0, 7.0, 16.0 - tmp, 12.0 - tmp]).real.max())


def power_function_beta(std: float, t_next: int, t_delta: int) -> float:
    return float((1 - t_delta / t_next) ** (std_to_exp(std) + 1))


def normalize_rows(w: torch.Tensor, eps: float = 1e-4) -> torch.Tensor:
    """mp_tools.normalize over all dims but 0."""
    dims = list(range(1, w.ndim))
    n = torch.linalg.vector_norm(w.float(), dim=dims, keepdim=True)
    n = torch.add(eps, n, alpha=math.sqrt(n.numel() / w.numel()))
    return w / n


def clip_coef(grads: dict, grad_scale: float, max_norm: float) -> tuple:
    norm = math.sqrt(sum(float((g.double() ** 2).sum()) for g in grads.values())) * grad_scale
    return min(1.0, max_norm / (norm + 1e-6)), norm


def adamw_ema_wn_step(params: dict, grads: dict, m: dict, v: dict, step: int, lr: float, grad_scale: float, max_norm: float,
                      emas: list, ema_betas: list, wn: set, beta1: float = 0.9, beta2: float = 0.99, eps: float = 1e-8,
                      weight_decay: float = 0.0) -> float:
    """In place on params / m / v / the EMA dicts.  emas: [(tensors dict, feedback_beta | None)], ema_betas: this step's betas;
    wn: names of the weight-normalised tensors.  Returns the (scaled) gradient norm before clipping."""
    coef, norm = clip_coef(grads, grad_scale, max_norm)
    b1c, b2c = 1 - beta1 ** step, 1 - beta2 ** step
    for k, p in params.items():
        g = grads[k].float() * (grad_scale * coef)
        m[k].mul_(beta1).add_(g, alpha=1 - beta1)
        v[k].mul_(beta2).addcmul_(g, g, value=1 - beta2)
        p.mul_(1 - lr * weight_decay)
        p.addcdiv_(m[k], (v[k].sqrt() / math.sqrt(b2c)).add_(eps), value=-lr / b1c)
    for (tensors, fb), beta in zip(emas, ema_betas):
        for k, p in params.items():
            tensors[k].lerp_(p, 1 - beta)
            if fb is not None:
                p.lerp_(tensors[k], 1 - fb)
    for k in wn:
        params[k].copy_(normalize_rows(params[k]))
    return norm
