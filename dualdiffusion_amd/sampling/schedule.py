"""Sigma schedules of the sampler (counterpart of reference src/sampling/schedule.py:28-76); host-side, a few dozen floats."""
from __future__ import annotations

import math

import torch


class SamplingSchedule:

    @staticmethod
    def get_schedule(name: str, steps: int, t_start: float = 1., device=None, **kwargs) -> torch.Tensor:
        fn = getattr(SamplingSchedule, f"schedule_{name}", None)
        if fn is None:
            raise ValueError(f"Unknown sampling schedule: {name}")
        return fn(torch.linspace(t_start, 0, int(steps) + 1, device=device), **kwargs)

    @staticmethod
    def get_schedules_list() -> list:
        return [a[len("schedule_"):] for a in dir(SamplingSchedule) if a.startswith("schedule_")]

    @staticmethod
    def schedule_edm2(t: torch.Tensor, sigma_max: float, sigma_min: float, rho: float = 7., **_) -> torch.Tensor:
        hi, lo = sigma_max ** (1 / rho), sigma_min ** (1 / rho)
        return (hi + (1 - t) * (lo - hi)) ** rho

    @staticmethod
    def schedule_ln_linear(t: torch.Tensor, sigma_max: float, sigma_min: float, **_) -> torch.Tensor:
        return (math.log(sigma_min) + (math.log(sigma_max) - math.log(sigma_min)) * t).exp()

    @staticmethod
    def schedule_linear(t: torch.Tensor, sigma_max: float, sigma_min: float, rho: float = 1., **_) -> torch.Tensor:
        return ((sigma_max ** (1 / rho) - sigma_min ** (1 / rho)) * t + sigma_min ** (1 / rho)) ** rho

    @staticmethod
    def schedule_cos(t: torch.Tensor, sigma_max: float, sigma_min: float, rho: float = 1., **_) -> torch.Tensor:
        th_max = math.pi / 2 - math.atan(sigma_max / rho)
        th_min = math.pi / 2 - math.atan(sigma_min / rho)
        theta = (1 - t) * (th_min - th_max) + th_max
        return theta.cos() / theta.sin() * rho

    @staticmethod
    def schedule_scale_invariant(t: torch.Tensor, sigma_max: float, sigma_min: float, rho: float = 1., **_) -> torch.Tensor:
        return sigma_min / ((1 - t) ** rho + sigma_min / sigma_max)
