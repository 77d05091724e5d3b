"""Sampling pipeline around the HIP UNet: the EDM sampler loop of the reference
(src/pipelines/dual_diffusion_pipeline.py:589-752 `diffusion_decode`) with the same `SampleParams` fields and call
signature.  Differences that do not change results: the sigma inputs of every step live in one device table built
before the loop, the element-wise step algebra (CFG lerp, Heun average, update + ancestral noise) runs in one HIP kernel
per expression (`ddx_lincomb3`), and there are no per-step `.item()` host syncs (the reference does four per step, :740-744).
"""
from __future__ import annotations

import importlib
import json
import math
import os
import re
from dataclasses import dataclass, field
from typing import Optional, Union

import numpy as np
import torch

from .. import ops
from ..sampling.schedule import SamplingSchedule


@dataclass
class ModuleInventory:
    """Checkpoints and EMA files found for one module of a model directory (reference dual_diffusion_pipeline.py:113-117)."""
    name: str
    checkpoints: list = field(default_factory=list)
    emas: dict = field(default_factory=dict)


def find_emas_in_dir(module_path: str) -> dict:
    """{ema name: file name} of the `ema_<name>.safetensors` files of a module directory, newest name first (reference ema.py:137-146)."""
    out = {}
    if os.path.isdir(module_path):
        for path in reversed(sorted(os.listdir(module_path))):
            if os.path.isfile(os.path.join(module_path, path)) and path.startswith("ema_") and path.endswith(".safetensors"):
                out[path[len("ema_"):-len(".safetensors")]] = path
    return out


def _import_module_class(package: str, cls: str):
    """model_index.json names the reference's packages (`modules.unets.unet_edm2_b4`); the drop-in classes live under
    `dualdiffusion_amd.` with the same relative path and class name (INTEGRATION.md section 1 shows the json edit; both spellings load)."""
    for name in (package, "dualdiffusion_amd." + package):
        try:
            mod = importlib.import_module(name)
        except ImportError:
            continue
        if hasattr(mod, cls) and mod.__name__.startswith("dualdiffusion_amd"):
            return getattr(mod, cls)
    raise ImportError(f"model_index.json: no dualdiffusion_amd drop-in for {package}.{cls}")


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
        self.model_metadata: dict = {}

    # ------------------------------------------------------------------ placement (reference :131-176)
    def to(self, device=None, dtype=None, memory_format=None, **kwargs) -> "DualDiffusionPipeline":
        for name, module in self.named_children():
            pick = lambda v: v.get(name) if isinstance(v, dict) else v      # noqa: E731  (per-module dicts as in the reference)
            module.to(device=pick(device), dtype=pick(dtype), memory_format=pick(memory_format), **kwargs)
        return self

    def half(self) -> "DualDiffusionPipeline":
        for module in self.children():
            module.to(dtype=torch.bfloat16)
        return self

    def compile(self, compile_options: Optional[dict] = None) -> None:
        """reference :178-187: a dict of per-module dicts {module_name: options} compiles only the listed modules, anything else is
        one option set for every module."""
        opts = compile_options or {}
        per_module = bool(opts) and all(isinstance(v, dict) for v in opts.values()) and "options" not in opts
        for name, module in self.named_children():
            if not hasattr(module, "compile"):
                continue
            if per_module:
                if name in opts:
                    module.compile(**opts[name])
            else:
                module.compile(**opts)

    # ------------------------------------------------------------------ model directory (reference :178-300)
    @staticmethod
    def get_model_module_classes(model_path: str) -> dict:
        with open(os.path.join(model_path, "model_index.json")) as f:
            index = json.load(f)
        return {name: _import_module_class(d["package"], d["class"]) for name, d in index["modules"].items()}

    @staticmethod
    def get_model_module_inventory(model_path: str) -> dict:
        with open(os.path.join(model_path, "model_index.json")) as f:
            index = json.load(f)
        inv = {}
        for name in index["modules"]:
            mi = ModuleInventory(name)
            for path in os.listdir(model_path):
                if os.path.isdir(os.path.join(model_path, path)) and name in path.split("_") and "_checkpoint-" in path:
                    mi.checkpoints.append(path)
            mi.checkpoints.sort(key=lambda x: int(re.search(r"\d+", x.split("-")[1]).group()))
            mi.emas[""] = list(find_emas_in_dir(os.path.join(model_path, name)).values())
            for ck in mi.checkpoints:
                mi.emas[ck] = list(find_emas_in_dir(os.path.join(model_path, ck, name)).values())
            inv[name] = mi
        return inv

    @staticmethod
    @torch.no_grad()
    def from_pretrained(model_path: str, torch_dtype=torch.float32, device=None, memory_format="channels_last",
                        load_checkpoints: Union[dict, bool, None] = False, load_emas: Union[dict, bool, None] = False,
                        compile_options: Optional[dict] = None) -> "DualDiffusionPipeline":
        """model_index.json -> module classes -> `{module}/{module}.json` + `.safetensors` (latest checkpoint / EMA on request),
        as reference dual_diffusion_pipeline.py:230-300."""
        classes = DualDiffusionPipeline.get_model_module_classes(model_path)
        inventory = DualDiffusionPipeline.get_model_module_inventory(model_path)
        if isinstance(load_checkpoints, bool) or load_checkpoints is None:
            load_checkpoints = {n: mi.checkpoints[-1] for n, mi in inventory.items() if mi.checkpoints} if load_checkpoints else {}
        if isinstance(load_emas, bool) or load_emas is None:
            load_emas = ({n: mi.emas[load_checkpoints.get(n, "")][-1] for n, mi in inventory.items() if mi.emas.get(load_checkpoints.get(n, ""))}
                         if load_emas else {})
        modules = {}
        for name, cls in classes.items():
            module_path = os.path.join(model_path, load_checkpoints.get(name, ""), name)
            modules[name] = cls.from_pretrained(module_path, load_config_only=name in load_emas)
            if name in load_emas:
                modules[name].load_ema(os.path.join(module_path, load_emas[name]), os.path.join(model_path, f"{name}_ema_archive"))
        if isinstance(memory_format, str):
            memory_format = getattr(torch, memory_format)
        pipe = DualDiffusionPipeline(modules).to(device=device, dtype=torch_dtype, memory_format=memory_format)
        if compile_options is not None:
            pipe.compile(compile_options)
        pipe.model_metadata = {"model_path": model_path, "model_module_classes": {n: str(c) for n, c in classes.items()},
                               "torch_dtype": torch_dtype, "memory_format": memory_format, "load_checkpoints": load_checkpoints,
                               "load_emas": load_emas, "compile_options": compile_options,
                               "last_global_step": {n: getattr(getattr(m, "config", None), "last_global_step", 0) for n, m in pipe.named_children()}}
        return pipe

    @torch.no_grad()
    def save_pretrained(self, model_path: str, subfolder: Optional[str] = None, save_config_only: bool = False) -> None:
        if subfolder is not None:
            model_path = os.path.join(model_path, subfolder)
        os.makedirs(model_path, exist_ok=True)
        index = {}
        for name, module in self.named_children():
            if hasattr(module, "save_pretrained"):
                index[name] = {"package": type(module).__module__, "class": type(module).__name__}
                module.save_pretrained(model_path, subfolder=name, save_config_only=save_config_only)
        with open(os.path.join(model_path, "model_index.json"), "w") as f:
            json.dump({"modules": index}, f, indent=2)

    # ------------------------------------------------------------------ shapes (reference :326-348)
    def _encoder(self):
        return getattr(self, "dae", None) or getattr(self, "vae", None)

    def get_mel_spec_shape(self, bsz: int = 1, raw_length: Optional[int] = None) -> tuple:
        fmt = self.format
        if hasattr(fmt, "get_mel_spec_shape"):
            shape = fmt.get_mel_spec_shape(bsz=bsz, raw_length=raw_length)
        elif hasattr(fmt, "get_mdct_shape"):
            shape = fmt.get_mdct_shape(bsz=bsz, raw_length=raw_length)
        else:
            shape = fmt.get_sample_shape(bsz=bsz, length=raw_length)           # SpectrogramFormat of the default model
        enc = self._encoder()
        if enc is None:
            return tuple(shape)
        latent = self.get_latent_shape(shape)
        return tuple(enc.get_mel_spec_shape(latent) if hasattr(enc, "get_mel_spec_shape") else enc.get_sample_shape(latent))

    def get_latent_shape(self, mel_spec_shape) -> Optional[torch.Size]:
        enc = self._encoder()
        if enc is None:
            return None
        latent = enc.get_latent_shape(mel_spec_shape)
        return self.unet.get_latent_shape(latent) if hasattr(self, "unet") else latent

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
        dev, B = unet.device, p.batch_size
        gen = torch.Generator(device=dev).manual_seed(p.seed)
        np_gen = np.random.default_rng(p.seed)
        if p.length is None and fmt is not None and hasattr(getattr(fmt, "config", None), "default_raw_length"):
            p.length = fmt.config.default_raw_length

        emb = None
        if audio_embedding is not None:
            mask = torch.cat((torch.ones(B, dtype=torch.bool), torch.zeros(B, dtype=torch.bool)))
            emb = unet.get_embeddings(audio_embedding, mask)           # (2B, cemb): conditioned rows first
        nb = 2 * B if emb is not None else B
        if x_ref is not None:
            sample_shape = tuple(sample_shape or x_ref.shape)
            ref_in = (x_ref.repeat(2, 1, 1, 1) if emb is not None else x_ref).to(dev, torch.float32).contiguous()
        else:
            if sample_shape is None:      # reference :617-622: the format's shape for the requested length, through the encoder
                if fmt is None:
                    raise ValueError("sample_shape is required when the pipeline has no format")
                mel = self.get_mel_spec_shape(bsz=B, raw_length=p.length)
                sample_shape = self.get_latent_shape(mel) if self._encoder() is not None else mel
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

        if p.seamless_loop:
            return self._decode_seamless(p, steps, sig_table, sample, ref_in, unet, fmt, emb, B, np_gen, draw)
        if self.step_graph and emb is not None and hasattr(unet, "_engine_for") and not unet.training and self._step_graph_ok(steps, sample, nb):
            return self._decode_step_graph(p, steps, sig_table, sample, ref_in, unet, fmt, emb, B, nb, draw)
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

    # the whole CFG + Heun step as ONE hipGraph; False: the eager step loop (UNet plan + lincomb launches per step)
    step_graph = os.environ.get("DDX_STEP_GRAPH", "1") != "0"

    # ddx_sampler_load serves at most 256 UNet rows; the per-step noise is staged up front (steps x sample fp32): beyond this budget the
    # eager loop (one draw per step) runs instead
    STEP_GRAPH_MAX_ROWS = 256
    STEP_GRAPH_NOISE_BYTES = 2 << 30
    STEP_PLAN_CACHE = 4

    def _step_graph_ok(self, steps, sample, nb) -> bool:
        any_noise = any(ng > 0 for (_s, _th, _t, ng) in steps)
        return nb <= self.STEP_GRAPH_MAX_ROWS and (not any_noise or len(steps) * sample.numel() * 4 <= self.STEP_GRAPH_NOISE_BYTES)

    def _decode_step_graph(self, p, steps, sig_table, sample, ref_in, unet, fmt, emb, B, nb, draw) -> torch.Tensor:
        """The sampler loop with one graph launch per step: [copy-in + sigma row, UNet forward, CFG lerp, Heun lerp, copy-in, UNet forward,
        CFG lerp, average, update (+ ancestral noise), step counter] recorded once over static buffers (the UNet engine's own launch
        plan is included twice); the per-step scalars (sigma rows, lerp weights, noise gain) and the step's noise tensor are read on
        the device through a step counter, so nothing runs on the host -- or in ATen -- between two UNet forwards.  Same kernels,
        same arithmetic and the same generator draws, in the same order, as the eager loop.  The recorded plan and its hipGraph are
        kept per (engine, steps, Heun, noise, cfg scale) and reused by later calls: only the tables and the sample are refreshed."""
        from .._lib import Plan
        dev = sample.device
        eng = unet._engine_for(nb, sample.shape[2], sample.shape[3], ref_in is not None)
        eng.prepare(fmt, emb, ref_in)
        n = len(steps)
        coef = torch.tensor([[1.0 - th, th, 1.0 - t, t, ng] for (_s, th, t, ng) in steps], dtype=torch.float32)
        any_noise = any(ng > 0 for (_s, _th, _t, ng) in steps)
        cache = self.__dict__.setdefault("_step_plans", {})
        key = (id(eng), n, bool(p.use_heun), any_noise, float(p.cfg_scale), tuple(sample.shape))
        st = cache.get(key)
        if st is not None and st["eng"] is not eng:      # (an id() recycled by a new engine)
            st = None
        if st is not None:
            cache[key] = cache.pop(key)                  # LRU: a hit moves the entry to the young end (dicts keep insertion order)
        if st is None:
            st = dict(eng=eng, sample=torch.empty_like(sample), coef=torch.empty(n, 5, dtype=torch.float32, device=dev),
                      sig_table=torch.empty_like(sig_table), stepc=torch.zeros(1, dtype=torch.int32, device=dev),
                      # drawn up front, in the eager loop's order (steps x sample: 0.56 GB at B = 16, 100 steps)
                      noise_buf=torch.zeros((n,) + tuple(sample.shape), device=dev, dtype=torch.float32) if any_noise else None)
            cfg, cfg_hat, x_hat = (torch.empty_like(sample) for _ in range(3))
            s_buf, c_buf, t_buf, stepc, noise_buf = st["sample"], st["coef"], st["sig_table"], st["stepc"], st["noise_buf"]
            eng.pb.launch(False)        # one eager pass of the UNet plan outside any capture (kernel attributes of a first launch)
            plan = Plan()
            with plan.record():
                ops.sampler_load(s_buf, eng.x_in, eng.x_pre, eng.sigma, t_buf, stepc, 0)
                plan.include(eng.fplan)
                ops.lincomb3(cfg, eng.out[:B], p.cfg_scale, eng.out[B:], 1.0 - p.cfg_scale)            # uncond.lerp(cond, cfg_scale)
                if p.use_heun:
                    ops.lincomb3_dev(x_hat, c_buf, stepc, cfg, 0, s_buf, 1)                            # lerp(cfg, sample, t_hat)
                    ops.sampler_load(x_hat, eng.x_in, eng.x_pre, eng.sigma, t_buf, stepc, 1)
                    plan.include(eng.fplan)
                    ops.lincomb3(cfg_hat, eng.out[:B], p.cfg_scale, eng.out[B:], 1.0 - p.cfg_scale)
                    ops.lincomb3(cfg, cfg, 0.5, cfg_hat, 0.5)
                ops.lincomb3_dev(s_buf, c_buf, stepc, cfg, 2, s_buf, 3, noise_buf, 4 if any_noise else -1, z_step_stride=s_buf.numel())
                ops.step_advance(stepc)
            plan.keepalive += [c_buf, noise_buf, stepc, cfg, cfg_hat, x_hat, t_buf, s_buf, eng]
            torch.cuda.current_stream().synchronize()
            cap = torch.cuda.Stream(device=dev)          # the legacy default stream cannot be captured
            plan.graph_build(cap.cuda_stream)
            cap.synchronize()
            st["plan"] = plan
            # at most STEP_PLAN_CACHE entries (least recently used goes first), each pinning its hipGraph, five sample-sized buffers and -- with
            # ancestral noise -- a steps x sample fp32 buffer (<= STEP_GRAPH_NOISE_BYTES): the memory bound of the cache is
            # STEP_PLAN_CACHE x (2 GiB + graph); sweeping cfg_scale or the step count re-records (both are part of the key)
            cache.pop(key, None)
            while len(cache) >= self.STEP_PLAN_CACHE:
                cache.pop(next(iter(cache)))
            cache[key] = st
        st["sample"].copy_(sample)
        st["coef"].copy_(coef)
        st["sig_table"].copy_(sig_table)
        st["stepc"].zero_()
        if any_noise:
            for i, (_s, _th, _t, ng) in enumerate(steps):
                if ng > 0:
                    st["noise_buf"][i].copy_(draw(1 + i))
        for _ in range(n):
            st["plan"].graph_launch()
        return st["sample"].clone()

    def _decode_seamless(self, p, steps, sig_table, sample, ref_in, unet, fmt, emb, B, np_gen, draw) -> torch.Tensor:
        """seamless_loop (reference :651-658, :729-732): every step rolls the sample (and the reference input) by a random shift
        along the time axis, pads 32 wrapped columns on either side, denoises the padded tensor and undoes both -- the UNet
        never sees a seam at a fixed place.  Tensor surgery (roll / cat / crop) as in the reference; the step algebra stays on
        ddx_lincomb3."""
        PADW = 32

        def wrap(x, shift):
            x = torch.roll(x, shifts=shift, dims=-1)
            return torch.cat((x[..., -PADW:], x, x[..., :PADW]), dim=-1).contiguous()

        def unwrap(x, shift):
            return torch.roll(x[..., PADW:-PADW], shifts=-shift, dims=-1).contiguous()

        def guided(x, sig_row, ref):
            x2 = x.repeat(2, 1, 1, 1) if emb is not None else x
            y = unet(x2, sig_row, fmt, emb, ref)
            if emb is not None:
                return ops.lincomb3(torch.empty_like(x), y[:B].contiguous(), p.cfg_scale, y[B:].contiguous(), 1.0 - p.cfg_scale)
            return y

        for i, (s_curr, t_hat, t, noise_gain) in enumerate(steps):
            shift = int(np_gen.integers(0, sample.shape[-1]))
            xs = wrap(sample, shift)
            ref = wrap(ref_in, shift) if ref_in is not None else None
            cfg = guided(xs, sig_table[i, 0], ref)
            if p.use_heun:
                x_hat = ops.lincomb3(torch.empty_like(xs), cfg, 1.0 - t_hat, xs, t_hat)
                cfg_hat = guided(x_hat, sig_table[i, 1], ref)
                ops.lincomb3(cfg, cfg, 0.5, cfg_hat, 0.5)
            ops.lincomb3(xs, cfg, 1.0 - t, xs, t)
            sample = unwrap(xs, shift)
            if noise_gain > 0:
                ops.lincomb3(sample, sample, 1.0, draw(1 + i), noise_gain)
        return sample
