"""Sampling pipeline around the HIP UNet: the EDM sampler loop of the reference
(src/pipelines/dual_diffusion_pipeline.py:589-752 `diffusion_decode`) with the same `SampleParams` fields and call
signature.  Differences that do not change results: the sigma inputs of every step live in one device table built
before the loop, the element-wise step algebra (CFG lerp, Heun average, update + ancestral noise) runs in one HIP kernel
per expression (`ddx_lincomb3`), and there are no per-step `.item()` host syncs (the reference does four per step, :740-744).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Union

import numpy as np
import torch

from .. import ops
from ..sampling.schedule import SamplingSchedule


@dataclass
class SampleParams:
    seed: Optional[int] = None
    num_steps: int = 100
    batch_size: int = 1
    length: Optional[int] = None
    seamless_loop: bool = False
    cfg_scale: float = 1.5
    sigma_max: Optional[float] = None
    sigma_min: Optional[float] = None
    sigma_data: Optional[float] = None
    rho: float = 7.
    schedule: Optional[str] = "edm2"
    prompt: Optional[str] = None
    use_heun: bool = True
    input_perturbation: float = 1.
    input_perturbation_offset: float = 0.
    stereo_fix: float = 0
    img2img_strength: float = 0.5
    input_audio: Optional[Union[str, torch.Tensor]] = None
    input_audio_pre_encoded: bool = False
    inpainting_mask: Optional[torch.Tensor] = None

    def sanitize(self) -> "SampleParams":
        self.seed = int(self.seed) if self.seed is not None else None
        self.length = int(self.length) if self.length is not None else None
        self.num_steps, self.batch_size, self.stereo_fix = int(self.num_steps), int(self.batch_size), float(self.stereo_fix)
        return self


class DualDiffusionPipeline(torch.nn.Module):
    """Holds the modules named in `model_index.json` (reference :205-228) and runs the sampler."""

    def __init__(self, pipeline_modules: dict) -> None:
        super().__init__()
        for name, module in pipeline_modules.items():
            if isinstance(module, torch.nn.Module):
                self.add_module(name, module)
            else:
                setattr(self, name, module)
        self.debug_info: dict = {}

    @torch.no_grad()
    def diffusion_decode(self, params: SampleParams, quiet: bool = False, audio_embedding: Optional[torch.Tensor] = None,
                         sample_shape: Optional[tuple] = None, x_ref: Optional[torch.Tensor] = None, module=None,
                         noises: Optional[list] = None) -> torch.Tensor:
        """EDM sampler with CFG batch doubling, Heun correction and sigma-dependent input perturbation.
        `noises` (optional, for parity tests): [initial noise, ancestral noise of step 0, 1, ...] instead of the generator."""
        unet = module if module is not None else getattr(self, "unet")
        fmt = getattr(self, "format", None)
        p = SampleParams(**params.__dict__).sanitize()
        p.seed = p.seed or int(np.random.randint(100000, 999999))
        p.sigma_max = p.sigma_max or unet.config.sigma_max
        p.sigma_min = p.sigma_min or unet.config.sigma_min
        p.sigma_data = p.sigma_data or unet.config.sigma_data
        if p.seamless_loop:
            raise NotImplementedError("seamless_loop sampling is not available on the HIP path yet")
        dev, B = unet.device, p.batch_size
        gen = torch.Generator(device=dev).manual_seed(p.seed)

        emb = None
        if audio_embedding is not None:
            mask = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool)))
            emb = unet.get_embeddings(audio_embedding, mask)           # (2B, cemb): conditioned rows first
        nb = 2 * B if emb is not None else B
        if x_ref is not None:
            sample_shape = tuple(sample_shape or x_ref.shape)
            ref_in = (x_ref.repeat(2, 1, 1, 1) if emb is not None else x_ref).to(dev, torch.float32).contiguous()
        else:
            if sample_shape is None:
                raise ValueError("sample_shape is required (no format-derived default on this path)")
            ref_in = None
        sample_shape = tuple(sample_shape)

        sched = SamplingSchedule.get_schedule(p.schedule, p.num_steps, 1, device="cpu", sigma_max=p.sigma_max, sigma_min=p.sigma_min, rho=p.rho)
        sig = sched.tolist()
        self.debug_info = {"sigma_schedule": sig, "effective_input_perturbation": []}

        # host pre-pass: every scalar of the loop is known before it starts -> one device table of sigma inputs
        steps = []
        for i, (s_curr, s_next) in enumerate(zip(sig[:-1], sig[1:])):
            old_next = s_next
            ipo = math.log(s_curr) + p.input_perturbation_offset
            eff = (math.tanh(ipo) / 2 + 0.5) * float(p.input_perturbation)
            s_next = s_next * (1 - max(min(eff, 1), 0))
            self.debug_info["effective_input_perturbation"].append(old_next - s_next)
            t_hat = max(old_next, p.sigma_min) / s_curr
            t = s_next / s_curr if (i + 1) < p.num_steps else 0.0
            noise_gain = max(old_next ** 2 - s_next ** 2, 0) ** 0.5 if (i + 1) < p.num_steps else 0.0
            steps.append((s_curr, t_hat, t, noise_gain))
        sig_table = torch.tensor([[s, th * s] for (s, th, _, _) in steps], dtype=torch.float32).repeat_interleave(nb, dim=1) \
            .reshape(len(steps), 2, nb).to(dev)

        f32 = dict(device=dev, dtype=torch.float32)
        draw = (lambda k: noises[k].to(**f32).contiguous()) if noises is not None else \
            (lambda k: torch.randn(sample_shape, device=dev, generator=gen))
        noise = draw(0)
        if p.stereo_fix > 0 and noises is None:
            noise[:, ::2] = noise[:, 1::2]
            fresh = torch.randn(sample_shape, device=dev, generator=gen)
            s = p.stereo_fix
            noise = ops.lincomb3(torch.empty_like(noise), fresh, (1 - s) / math.hypot(1 - s, s), noise, s / math.hypot(1 - s, s))
        sample = ops.lincomb3(torch.empty(sample_shape, **f32), noise, (sig[0] ** 2 + p.sigma_data ** 2) ** 0.5)
        x2 = torch.empty((nb,) + sample_shape[1:], **f32)
        cfg, cfg_hat, x_hat = (torch.empty(sample_shape, **f32) for _ in range(3))

        def guided(x: torch.Tensor, sig_row: torch.Tensor, out: torch.Tensor) -> torch.Tensor:
            x2[:B].copy_(x)
            if emb is not None:
                x2[B:].copy_(x)
            y = unet(x2, sig_row, fmt, emb, ref_in)
            if emb is not None:   # uncond.lerp(cond, cfg_scale)
                return ops.lincomb3(out, y[:B].contiguous(), p.cfg_scale, y[B:].contiguous(), 1.0 - p.cfg_scale)
            return out.copy_(y)

        for i, (s_curr, t_hat, t, noise_gain) in enumerate(steps):
            guided(sample, sig_table[i, 0], cfg)
            if p.use_heun:
                ops.lincomb3(x_hat, cfg, 1.0 - t_hat, sample, t_hat)             # lerp(cfg, sample, t_hat)
                guided(x_hat, sig_table[i, 1], cfg_hat)
                ops.lincomb3(cfg, cfg, 0.5, cfg_hat, 0.5)
            if noise_gain > 0:
                ops.lincomb3(sample, cfg, 1.0 - t, sample, t, draw(1 + i), noise_gain)
            else:
                ops.lincomb3(sample, cfg, 1.0 - t, sample, t)
        return sample
