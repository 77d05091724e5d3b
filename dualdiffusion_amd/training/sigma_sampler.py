"""Per-sigma stratified noise-level sampling for training (host side, microseconds).

Counterpart of reference src/training/sigma_sampler.py:61-211: one draw covers the whole GLOBAL batch so that every
mini-batch is stratified across ranks (the draw is broadcast from rank 0 and strided per rank, see
dualdiffusion_amd/distributed.py).  Each distribution is an inverse CDF applied to the stratified quantiles.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional

import torch

DISTRIBUTIONS = ("ln_normal", "ln_sech", "ln_sech^2", "ln_linear", "ln_pdf", "scale_invariant", "linear")


@dataclass
class SigmaSamplerConfig:
    sigma_max: float = 200.
    sigma_min: float = 0.03
    sigma_data: float = 1.
    distribution: str = "ln_sech"
    dist_scale: float = 1.
    dist_offset: float = 0.3
    dist_pdf: Optional[torch.Tensor] = None
    use_stratified_sigma_sampling: bool = True
    use_static_sigma_sampling: bool = False
    sigma_pdf_warmup_steps: int = 5000
    sigma_pdf_resolution: int = 127
    sigma_pdf_sanitization: bool = True
    sigma_pdf_offset: float = 0
    sigma_pdf_min: float = 1e-3

    @property
    def ln_sigma_min(self) -> float:
        return math.log(self.sigma_min)

    @property
    def ln_sigma_max(self) -> float:
        return math.log(self.sigma_max)


def _unimodal(pdf: torch.Tensor) -> torch.Tensor:
    """Force a histogram to rise monotonically to its peak and fall after it (reference _sanitize_pdf, :166-170)."""
    peak = int(torch.argmax(pdf))
    return torch.cat([torch.cummax(pdf[:peak + 1], dim=0).values, torch.cummin(pdf[peak:], dim=0).values[1:]])


class SigmaSampler:

    def __init__(self, config: SigmaSamplerConfig) -> None:
        if config.distribution not in DISTRIBUTIONS:
            raise ValueError(f"Invalid distribution: {config.distribution}")
        self.config = config
        if config.distribution == "ln_pdf":
            pdf = config.dist_pdf if config.dist_pdf is not None else torch.ones(config.sigma_pdf_resolution)
            self.update_pdf(pdf, first=True)

    # ------------------------------------------------------------------ quantiles (reference :94-109)
    def quantiles(self, n: int, jitter: Optional[torch.Tensor] = None) -> Optional[torch.Tensor]:
        c = self.config
        centres = (torch.arange(n) + 0.5) / n
        if c.use_static_sigma_sampling:
            return centres
        if c.use_stratified_sigma_sampling:
            u = jitter if jitter is not None else torch.rand(1)
            return centres + (u - 0.5) / n
        return None

    @torch.no_grad()
    def sample(self, n_samples: int, device=None, jitter: Optional[torch.Tensor] = None) -> torch.Tensor:
        q = self.quantiles(n_samples, jitter)
        if q is None:
            q = torch.rand(n_samples)
        return self.from_quantiles(q).to(device)

    # ------------------------------------------------------------------ inverse CDFs (reference :111-165, :193-211)
    def from_quantiles(self, q: torch.Tensor) -> torch.Tensor:
        c = self.config
        lo, hi = c.sigma_min, c.sigma_max
        d = c.distribution
        if d == "ln_sech":
            th_lo = math.atan(1 / hi * math.exp(c.dist_offset))
            th_hi = math.atan(1 / lo * math.exp(c.dist_offset))
            theta = q * (th_hi - th_lo) + th_lo
            ln_sigma = (1 / theta.tan()).log() * c.dist_scale + c.dist_offset
        elif d == "ln_normal":
            cdf = lambda s: 0.5 * (1 + math.erf((2 ** 0.5 * s - 2 ** 0.5 * c.dist_offset) / (2 * c.dist_scale)))
            q_lo, q_hi = cdf(c.ln_sigma_min), cdf(c.ln_sigma_max)
            qq = q_lo + q * (q_hi - q_lo)
            ln_sigma = c.dist_offset + (c.dist_scale * 2 ** 0.5) * (qq * 2 - 1).erfinv().clip(min=-6, max=6)
        elif d == "ln_sech^2":
            a, b = math.tanh(c.ln_sigma_min), math.tanh(c.ln_sigma_max)
            ln_sigma = (q * (b - a) + a).atanh() * c.dist_scale + c.dist_offset
            span = c.ln_sigma_max - c.ln_sigma_min
            ln_sigma = torch.where(ln_sigma < c.ln_sigma_min, ln_sigma + span, ln_sigma)
            ln_sigma = torch.where(ln_sigma > c.ln_sigma_max, ln_sigma - span, ln_sigma)
        elif d == "ln_linear":
            ln_sigma = q * (c.ln_sigma_max - c.ln_sigma_min) + c.ln_sigma_min
        elif d == "ln_pdf":
            ln_sigma = self._invert_pdf(q) * (c.ln_sigma_max - c.ln_sigma_min) + c.ln_sigma_min
        elif d == "scale_invariant":
            a, b = 1 / hi ** c.dist_scale, 1 / lo ** c.dist_scale
            return 1 / (q * (b - a) + a) ** (1 / c.dist_scale)
        else:  # linear
            a, b = lo ** (1 / c.dist_scale), hi ** (1 / c.dist_scale)
            return (q * (b - a) + a).pow(c.dist_scale).clip(lo, hi)
        return ln_sigma.exp().clip(min=lo, max=hi)

    # ------------------------------------------------------------------ learned pdf (reference :166-203)
    def update_pdf(self, pdf: torch.Tensor, first: bool = False) -> None:
        if self.config.sigma_pdf_sanitization:
            pdf = _unimodal(pdf)
        self.dist_pdf = pdf / pdf.sum()
        self.dist_cdf = torch.cat((torch.zeros(1, device=pdf.device), self.dist_pdf.cumsum(dim=0)))

    def update_pdf_from_logvar(self, unet, global_step: int) -> None:
        c = self.config
        warm = min(global_step / c.sigma_pdf_warmup_steps, 1) if c.sigma_pdf_warmup_steps > 0 else 1
        ln_sigma = torch.linspace(c.ln_sigma_min, c.ln_sigma_max, c.sigma_pdf_resolution)
        err = unet.get_sigma_loss_logvar(ln_sigma.exp().to(unet.device)).float().flatten().detach().cpu()
        self.update_pdf(((-warm * c.dist_scale * err).exp() + c.sigma_pdf_offset).clip(min=c.sigma_pdf_min))

    def _invert_pdf(self, q: torch.Tensor) -> torch.Tensor:
        cdf = self.dist_cdf
        idx = torch.searchsorted(cdf, q.to(cdf.device), out_int32=True).clip(max=cdf.shape[0] - 2)
        left, right = cdf[idx], cdf[idx + 1]
        return (idx + (q - left) / (right - left)) / (cdf.shape[0] - 1)
