"""CPU oracle for the EDM2-UNet denoising path  --  TEST INFRASTRUCTURE ONLY.

This file is a CPU (torch fp32, no nn.Module, state-dict driven) restatement of the reference's
algorithm for the hot path.  Only `tests/`, `__graft_entry__.smoke()` and the `cpu_baseline` leg
of `bench.py` may import it; the product path (`dualdiffusion_amd/`) never does and fails loudly
when its HIP library is missing.

Parity status: PINNED.  `tools/make_golden.py` (build container only) imports the reference from
/root/reference, runs it on seeded inputs, checks this oracle against it and commits the vectors
under `tests/golden/`; `tests/test_oracle_golden.py` re-checks the oracle against those vectors
everywhere.  The reference's own tests hold no golden vectors for this path.

Every function cites the reference lines (under /root/reference/src) whose arithmetic it restates.
Tensors are NCHW fp32 on the CPU; `sd` is a flat {state_dict key: tensor} mapping using the
reference's key names.
"""
from __future__ import annotations

import math
from typing import Optional, Sequence

import torch
import torch.nn.functional as F

EPS_NORM = 1e-4


# ----------------------------------------------------------------------------- op library (a-1 .. a-5)

def rms_normalize(x: torch.Tensor, dims: Optional[Sequence[int]] = None, eps: float = EPS_NORM) -> torch.Tensor:
    """modules/mp_tools.py:42-49.  x / (eps + ||x||_2 * sqrt(n_norm / n_x)), fp32 internally."""
    if dims is None:
        dims = list(range(1, x.ndim))
    xf = x.to(torch.float32)
    nrm = torch.sqrt((xf * xf).sum(dim=list(dims), keepdim=True))
    nrm = eps + nrm * math.sqrt(nrm.numel() / x.numel())
    return (xf / nrm).to(x.dtype)


def silu_mp(x: torch.Tensor) -> torch.Tensor:
    """modules/mp_tools.py:268-269."""
    return x * torch.sigmoid(x) / 0.596


def sum_mp(a: torch.Tensor, b: torch.Tensor, t) -> torch.Tensor:
    """modules/mp_tools.py:274-279.  lerp(a, b, t) / sqrt((1-t)^2 + t^2); t float or tensor."""
    if isinstance(t, torch.Tensor):
        return (a + (b - a) * t) / torch.sqrt((1 - t) ** 2 + t ** 2).to(a.dtype)
    return (a + (b - a) * t) / math.sqrt((1 - t) ** 2 + t ** 2)


def cat_mp_weights(na: int, nb: int, t: float) -> tuple[float, float]:
    """modules/mp_tools.py:294-301 (the two scalars)."""
    c = math.sqrt((na + nb) / ((1 - t) ** 2 + t ** 2))
    return c / math.sqrt(na) * (1 - t), c / math.sqrt(nb) * t


def cat_mp(a: torch.Tensor, b: torch.Tensor, t: float = 0.5) -> torch.Tensor:
    wa, wb = cat_mp_weights(a.shape[1], b.shape[1], t)
    return torch.cat([wa * a, wb * b], dim=1)


def resample2x(x: torch.Tensor, mode: str) -> torch.Tensor:
    """modules/mp_tools.py:71-79.  'down' = 2x2 mean (not rescaled), 'up' = nearest x2."""
    if mode == "keep":
        return x
    if mode == "down":
        b, c, h, w = x.shape
        return x.reshape(b, c, h // 2, 2, w // 2, 2).mean(dim=(3, 5))
    if mode == "up":
        return x.repeat_interleave(2, dim=2).repeat_interleave(2, dim=3)
    raise ValueError(mode)


def fourier_tables(num_channels: int, bandwidth: float = 1.0, eps: float = 1e-3) -> tuple[torch.Tensor, torch.Tensor]:
    """modules/mp_tools.py:318-322: freqs = pi*erfinv(linspace(0,1-eps,C))*bw ; phases = pi/2 on even idx."""
    freqs = torch.pi * torch.linspace(0, 1 - eps, num_channels).erfinv() * bandwidth
    phases = torch.pi / 2 * (torch.arange(num_channels) % 2 == 0).float()
    return freqs, phases


def fourier_mp(x: torch.Tensor, freqs: torch.Tensor, phases: torch.Tensor) -> torch.Tensor:
    """modules/mp_tools.py:324-330 (1-D input form)."""
    y = x.float()[:, None] * freqs.float()[None, :] + phases.float()[None, :]
    return torch.cos(y) * math.sqrt(2.0)


def prepared_weight(w: torch.Tensor, gain=1.0, training: bool = False, weight_norm: bool = True) -> torch.Tensor:
    """modules/mp_tools.py:359-364: optional forced weight-norm, then gain / sqrt(fan_in)."""
    w = w.float()
    if training and weight_norm:
        w = rms_normalize(w)
    return w * (gain / math.sqrt(w[0].numel()))


def conv_mp(x: torch.Tensor, w: torch.Tensor, gain=1.0, groups: int = 1, training: bool = False,
            weight_norm: bool = True) -> torch.Tensor:
    """modules/mp_tools.py:357-373 (no bias on this path).  2-D weight -> x @ w.T, else same-pad conv."""
    wp = prepared_weight(w, gain, training, weight_norm).to(x.dtype)
    if wp.ndim == 2:
        return x @ wp.t()
    return F.conv2d(x, wp, padding=(wp.shape[-2] // 2, wp.shape[-1] // 2), groups=groups)


# ----------------------------------------------------------------------------- attention (inside a-6)

def attention_2d(qk: torch.Tensor, v: torch.Tensor, heads: int) -> torch.Tensor:
    """modules/unets/unet_edm2_b4.py:137-148.  qk: (B, 2C, H, W) with channel = (head, d, {q,k});
    v: (B, C, H, W) with channel = (head, d).  Per-token RMS-normalised q, k, v over d, softmax(q.k/sqrt(d))."""
    b, c2, h, w = qk.shape
    c = c2 // 2
    d = c // heads
    qk5 = rms_normalize(qk.reshape(b, heads, d, 2, h * w), dims=[2])
    q, k = qk5[:, :, :, 0], qk5[:, :, :, 1]                       # (B, heads, d, T)
    v4 = rms_normalize(v.reshape(b, heads, d, h * w), dims=[2])
    s = torch.einsum("bhdq,bhdk->bhqk", q, k) / math.sqrt(d)
    p = torch.softmax(s, dim=-1)
    o = torch.einsum("bhqk,bhdk->bhdq", p, v4)
    return o.reshape(b, c, h, w)


# ----------------------------------------------------------------------------- Block (a-6)

def block_forward(sd: dict, prefix: str, x: torch.Tensor, emb: torch.Tensor, *, flavor: str, resample: str,
                  attention: bool, heads: int, groups: int, res_balance: float = 0.3, attn_balance: float = 0.3,
                  clip: Optional[float] = 256.0, training: bool = False, dropout: float = 0.0,
                  dropout_mask: Optional[torch.Tensor] = None) -> torch.Tensor:
    """modules/unets/unet_edm2_b4.py:110-158.  Dropout (:124-125, training only): `dropout_mask` is the keep mask (1 = kept) of the
    draw, shaped like the hidden tensor -- torch's dropout scales the kept values by 1 / (1 - p), the block then by (1 - p)^0.5."""
    def W(name):
        return sd[f"{prefix}.{name}.weight"]

    x = resample2x(x, resample)
    if flavor == "enc":
        x = conv_mp(x, W("conv_skip"), training=training)
        x = rms_normalize(x, dims=[1])
    y = conv_mp(silu_mp(x), W("conv_res0"), groups=groups, training=training)
    c = conv_mp(emb, W("emb_linear"), gain=sd[f"{prefix}.emb_gain"], groups=groups, training=training) + 1.0
    y = silu_mp(y * c)
    if dropout != 0 and training:
        assert dropout_mask is not None and dropout_mask.shape == y.shape, "training with dropout needs the keep mask of the draw"
        y = y * dropout_mask.to(y.dtype) / (1.0 - dropout) * (1.0 - dropout) ** 0.5
    y = conv_mp(y, W("conv_res1"), groups=groups, training=training)
    if flavor == "dec":
        x = conv_mp(x, W("conv_skip"), training=training)
    x = sum_mp(x, y, res_balance)
    if attention:
        c = conv_mp(emb, W("emb_linear_qk"), gain=sd[f"{prefix}.emb_gain_qk"], training=training) + 1.0
        qk = conv_mp(x * c, W("attn_qk"), training=training)
        v = conv_mp(x, W("attn_v"), training=training)
        y = attention_2d(qk, v, heads)
        c = conv_mp(emb, W("emb_linear_v"), gain=sd[f"{prefix}.emb_gain_v"], training=training) + 1.0
        y = silu_mp(y * c)
        y = conv_mp(y, W("attn_proj"), training=training)
        x = sum_mp(x, y, attn_balance)
    if clip is not None:
        x = x.clamp(-clip, clip)
    return x


# ----------------------------------------------------------------------------- UNet (a-7 .. a-9)

DEFAULT_UNET_CFG = dict(
    in_channels=4, out_channels=4, in_channels_emb=512, dropout=0.0, sigma_max=200.0, sigma_min=0.03,
    sigma_data=1.0, model_channels=256, logvar_channels=128, channel_mult=(1, 2, 3, 4, 5),
    channel_mult_noise=None, channel_mult_emb=None, channels_per_head=64, num_layers_per_block=2,
    label_balance=0.5, concat_balance=0.5, res_balance=0.3, attn_balance=0.3, attn_levels=(3, 4),
    mlp_multiplier=2, mlp_groups=8)


def unet_cfg(**overrides) -> dict:
    cfg = dict(DEFAULT_UNET_CFG)
    cfg.update(overrides)
    return cfg


def unet_topology(cfg: dict) -> dict:
    """modules/unets/unet_edm2_b4.py:172-230: the ordered list of encoder / decoder stages.

    Returns {"cblock","cnoise","cemb","enc":[...],"dec":[...]} where each stage is a dict with
    name, kind ('conv_in' | 'block'), cin, cout, level, flavor, resample, attention, skip_in (dec only).
    """
    mc = cfg["model_channels"]
    cblock = [mc * m for m in cfg["channel_mult"]]
    cnoise = mc * cfg["channel_mult_noise"] if cfg["channel_mult_noise"] is not None else max(cblock)
    cemb = mc * cfg["channel_mult_emb"] if cfg["channel_mult_emb"] is not None else max(cblock)
    attn_levels = set(cfg["attn_levels"])
    enc, dec = [], []
    cout = cfg["in_channels"] + 2
    for level, ch in enumerate(cblock):
        if level == 0:
            enc.append(dict(name="conv_in", kind="conv_in", cin=cout, cout=ch, level=0))
            cout = ch
        else:
            enc.append(dict(name=f"block{level}_down", kind="block", cin=cout, cout=cout, level=level, flavor="enc",
                            resample="down", attention=level in attn_levels))
        for i in range(cfg["num_layers_per_block"]):
            enc.append(dict(name=f"block{level}_layer{i}", kind="block", cin=cout, cout=ch, level=level,
                            flavor="enc", resample="keep", attention=level in attn_levels))
            cout = ch
    skips = [s["cout"] for s in enc]
    top = len(cblock) - 1
    for level in range(top, -1, -1):
        ch = cblock[level]
        if level == top:
            for nm in ("in0", "in1"):
                dec.append(dict(name=f"block{level}_{nm}", kind="block", cin=cout, cout=cout, level=level,
                                flavor="dec", resample="keep", attention=True, skip_in=0))
        else:
            dec.append(dict(name=f"block{level}_up", kind="block", cin=cout, cout=cout, level=level, flavor="dec",
                            resample="up", attention=level in attn_levels, skip_in=0))
        for i in range(cfg["num_layers_per_block"] + 1):
            sk = skips.pop()
            dec.append(dict(name=f"block{level}_layer{i}", kind="block", cin=cout + sk, cout=ch, level=level,
                            flavor="dec", resample="keep", attention=level in attn_levels, skip_in=sk))
            cout = ch
    return dict(cblock=cblock, cnoise=cnoise, cemb=cemb, enc=enc, dec=dec, cout_last=cout)


def unet_param_shapes(cfg: dict) -> dict:
    """State-dict key -> shape for the reference UNet (unet_edm2_b4.py:180-230; checked in make_golden)."""
    topo = unet_topology(cfg)
    g, mm = cfg["mlp_groups"], cfg["mlp_multiplier"]
    cemb = topo["cemb"]
    shapes = {
        "out_gain": (),
        "emb_fourier.freqs": (topo["cnoise"],), "emb_fourier.phases": (topo["cnoise"],),
        "emb_noise.weight": (cemb, topo["cnoise"]),
        "emb_label.weight": (cemb, cfg["in_channels_emb"]),
        "emb_label_unconditional.weight": (cemb, 1),
        "logvar_fourier.freqs": (cfg["logvar_channels"],), "logvar_fourier.phases": (cfg["logvar_channels"],),
        "logvar_linear.weight": (1, cfg["logvar_channels"]),
    }
    for side in ("enc", "dec"):
        for st in topo[side]:
            p = f"{side}.{st['name']}"
            if st["kind"] == "conv_in":
                shapes[f"{p}.weight"] = (st["cout"], st["cin"], 3, 3)
                continue
            cin, cout = st["cin"], st["cout"]
            res0_in = cout if st["flavor"] == "enc" else cin
            shapes[f"{p}.emb_gain"] = ()
            if st["attention"]:
                shapes[f"{p}.emb_gain_qk"] = ()
                shapes[f"{p}.emb_gain_v"] = ()
            shapes[f"{p}.conv_res0.weight"] = (cout * mm, res0_in // g, 3, 3)
            shapes[f"{p}.conv_res1.weight"] = (cout, cout * mm // g, 3, 3)
            shapes[f"{p}.conv_skip.weight"] = (cout, cin, 1, 1)
            shapes[f"{p}.emb_linear.weight"] = (cout * mm, cemb // g, 1, 1)
            if st["attention"]:
                shapes[f"{p}.emb_linear_qk.weight"] = (cout, cemb, 1, 1)
                shapes[f"{p}.emb_linear_v.weight"] = (cout, cemb, 1, 1)
                shapes[f"{p}.attn_qk.weight"] = (cout * 2, cout, 1, 1)
                shapes[f"{p}.attn_v.weight"] = (cout, cout, 1, 1)
                shapes[f"{p}.attn_proj.weight"] = (cout, cout, 1, 1)
    shapes["conv_out.weight"] = (cfg["out_channels"], topo["cout_last"], 3, 3)
    return shapes


def random_unet_state(cfg: dict, seed: int, gain_value: float = 0.7, normalized: bool = True) -> dict:
    """Deterministic synthetic weights: randn per key (sorted order, one generator), forced weight-norm
    (mp_tools.py:375-378) and every 0-d gain set to `gain_value` (SURVEY.md section 0.5a parity trap)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in sorted(unet_param_shapes(cfg).items()):
        if key.endswith(".freqs") or key.endswith(".phases"):
            continue
        if shape == ():
            sd[key] = torch.tensor(gain_value)
        else:
            w = torch.randn(shape, generator=g)
            if normalized and key != "logvar_linear.weight":
                w = rms_normalize(w)
            sd[key] = w
    topo = unet_topology(cfg)
    sd["emb_fourier.freqs"], sd["emb_fourier.phases"] = fourier_tables(topo["cnoise"])
    sd["logvar_fourier.freqs"], sd["logvar_fourier.phases"] = fourier_tables(cfg["logvar_channels"])
    return sd


def hz_to_mel(f: float) -> float:
    """modules/formats/frequency_scale.py:30-31 (numpy float64 in the reference)."""
    return 2595.0 * math.log10(1.0 + f / 700.0)


def mel_points_hz(n: int, fmin: float, fmax: float) -> torch.Tensor:
    """frequency_scale.py:144-149 with scale 'mel': f32 linspace between f64 endpoints, then mel->Hz in f32."""
    mels = torch.linspace(hz_to_mel(fmin), hz_to_mel(fmax), n)
    return 700.0 * (10.0 ** (mels / 2595.0) - 1.0)


def ln_freq_channel(h: int, w: int, batch: int, fmin: float, fmax: float) -> torch.Tensor:
    """unet_edm2_b4.py:244-248 == formats/old/spectrogram.py:240-244: standardised log2 mel-Hz per row."""
    lf = mel_points_hz(h + 2, fmin, fmax)[1:-1].log2()
    lf = lf.view(1, 1, -1, 1).repeat(batch, 1, 1, w)
    return (lf - lf.mean()) / lf.std()


def unet_embeddings(sd: dict, cfg: dict, emb_in: torch.Tensor, cond_mask: torch.Tensor, training: bool = False) -> torch.Tensor:
    """unet_edm2_b4.py:232-235."""
    u = conv_mp(torch.ones(1), sd["emb_label_unconditional.weight"], training=training)
    c = conv_mp(rms_normalize(emb_in.float()), sd["emb_label.weight"], training=training)
    return sum_mp(u, c, cond_mask.float().unsqueeze(1))


def unet_sigma_logvar(sd: dict, cfg: dict, sigma: torch.Tensor) -> torch.Tensor:
    """unet_edm2_b4.py:237-238 (logvar_linear has weight-norm disabled, unet_edm2_b4.py:187)."""
    f = fourier_mp(sigma.flatten().log() / 4, sd["logvar_fourier.freqs"], sd["logvar_fourier.phases"])
    return conv_mp(f, sd["logvar_linear.weight"], weight_norm=False).view(-1, 1, 1, 1).float()


def unet_forward(sd: dict, cfg: dict, x_in: torch.Tensor, sigma: torch.Tensor, embeddings: torch.Tensor,
                 freq_range: tuple[float, float] = (20.0, 16000.0), x_ref: Optional[torch.Tensor] = None,
                 perturbed_input: Optional[torch.Tensor] = None, training: bool = False,
                 collect: Optional[dict] = None, dropout_masks: Optional[dict] = None) -> torch.Tensor:
    """unet_edm2_b4.py:250-296.  `freq_range` = (freq_min, freq_max) of format.ms_freq_scale (mel scale).
    `collect`, if given, receives every stage output by name (for layer-level parity tests).
    `dropout_masks` (training with cfg["dropout"] > 0): {"enc.<block>" | "dec.<block>": keep mask of that block's draw}."""
    topo = unet_topology(cfg)
    sdata = cfg["sigma_data"]
    sig = sigma.float().view(-1, 1, 1, 1)
    c_skip = sdata ** 2 / (sig ** 2 + sdata ** 2)
    c_out = sig * sdata / torch.sqrt(sig ** 2 + sdata ** 2)
    c_in = 1 / torch.sqrt(sdata ** 2 + sig ** 2)
    c_noise = sig.flatten().log() / 4
    x = c_in * (perturbed_input if perturbed_input is not None else x_in).float()

    emb = conv_mp(fourier_mp(c_noise, sd["emb_fourier.freqs"], sd["emb_fourier.phases"]), sd["emb_noise.weight"],
                  training=training)
    emb = sum_mp(emb, embeddings.float(), cfg["label_balance"])
    emb = silu_mp(emb)[:, :, None, None]
    if collect is not None:
        collect["emb"] = emb

    b, _, h, w = x.shape
    x = torch.cat([x, torch.ones_like(x[:, :1]), ln_freq_channel(h, w, b, *freq_range)], dim=1)
    heads_of = lambda cout: cout // cfg["channels_per_head"]
    kw = dict(groups=cfg["mlp_groups"], res_balance=cfg["res_balance"], attn_balance=cfg["attn_balance"],
              training=training, dropout=cfg.get("dropout", 0.0))
    dm = dropout_masks or {}
    skips = []
    for st in topo["enc"]:
        if st["kind"] == "conv_in":
            x = conv_mp(x, sd["enc.conv_in.weight"], training=training)
        else:
            x = block_forward(sd, f"enc.{st['name']}", x, emb, flavor="enc", resample=st["resample"],
                              attention=st["attention"], heads=heads_of(st["cout"]), dropout_mask=dm.get(f"enc.{st['name']}"), **kw)
        skips.append(x)
        if collect is not None:
            collect[f"enc.{st['name']}"] = x
    for st in topo["dec"]:
        if st["skip_in"]:
            x = cat_mp(x, skips.pop(), cfg["concat_balance"])
        x = block_forward(sd, f"dec.{st['name']}", x, emb, flavor="dec", resample=st["resample"],
                          attention=st["attention"], heads=heads_of(st["cout"]), dropout_mask=dm.get(f"dec.{st['name']}"), **kw)
        if collect is not None:
            collect[f"dec.{st['name']}"] = x
    x = conv_mp(x, sd["conv_out.weight"], gain=sd["out_gain"], training=training)
    d_x = c_skip * x_in.float() + c_out * x.float()
    if x_ref is not None:
        d_x = sum_mp(x_ref[:, :-1].float(), d_x, x_ref[:, -1:].float())
    return d_x


def unet_train_loss(sd: dict, cfg: dict, samples: torch.Tensor, audio_embeddings: torch.Tensor, sigma: torch.Tensor, noise: torch.Tensor,
                    cond_mask: torch.Tensor, input_perturbation: Optional[torch.Tensor] = None, input_perturbation_scale: float = 0.0,
                    freq_range: tuple[float, float] = (20.0, 16000.0), *, conditioning_perturbation: Optional[torch.Tensor] = None,
                    conditioning_perturbation_scale: float = 0.0, normalize_latents: bool = False, dynamic_sigma_data: Optional[tuple] = None,
                    ref_samples: Optional[torch.Tensor] = None, dropout_masks: Optional[dict] = None) -> torch.Tensor:
    """Device part of training/module_trainers/unet_trainer.py:203-296 (`train_batch` / `unet_train_batch`, train branch) with the random
    draws given: normalize_latents :205-206, get_embeddings :236, conditioning_perturbation :241-243 (embeddings + draw * scale),
    noised / perturbed input :249-259, UNet forward in training mode :261 (`ref_samples` = x_ref, `dropout_masks` the keep masks of the
    blocks' dropout draws), loss weight and MSE :271-276 with use_dynamic_sigma_data :263-269 (`dynamic_sigma_data` = (min, max, exp)),
    Gaussian NLL with the learned per-sigma log-variance :280-282.  Returns the per-sample loss [B]."""
    if normalize_latents:
        samples = rms_normalize(samples).float()
    emb = unet_embeddings(sd, cfg, audio_embeddings, cond_mask, training=True)
    if conditioning_perturbation is not None and conditioning_perturbation_scale > 0:
        emb = emb + conditioning_perturbation * conditioning_perturbation_scale
    s4 = sigma.float().view(-1, 1, 1, 1)
    x_in = samples + noise * s4
    pert = x_in + input_perturbation * s4 * input_perturbation_scale if input_perturbation is not None else None
    denoised = unet_forward(sd, cfg, x_in, sigma, emb, freq_range, x_ref=ref_samples, perturbed_input=pert, training=True,
                            dropout_masks=dropout_masks)
    if dynamic_sigma_data is not None:
        lo, hi, ex = dynamic_sigma_data
        n = samples.shape[1] * samples.shape[2] * samples.shape[3]
        sdata = (torch.linalg.vector_norm(samples, dim=(1, 2, 3), keepdim=True) / n ** 0.5).clip(min=lo, max=hi) ** ex
    else:
        sdata = cfg["sigma_data"]
    w = (s4 ** 2 + sdata ** 2) / (s4 * sdata) ** 2
    wl = (F.mse_loss(denoised, samples, reduction="none") * w).mean(dim=(1, 2, 3))
    logvar = unet_sigma_logvar(sd, cfg, sigma).flatten()
    return wl / logvar.exp() + logvar


def unet_latent_shape(cfg: dict, shape: Sequence[int]) -> tuple:
    """unet_edm2_b4.py:240-242."""
    q = 2 ** (len(cfg["channel_mult"]) - 1)
    return tuple(shape[0:2]) + ((shape[2] // q) * q, (shape[3] // q) * q)


# ----------------------------------------------------------------------------- schedule (a-16)

def schedule_edm2(steps: int, sigma_max: float, sigma_min: float, rho: float = 7.0) -> torch.Tensor:
    """sampling/schedule.py:34-37,57-59."""
    t = torch.linspace(1, 0, steps + 1)
    return (sigma_max ** (1 / rho) + (1 - t) * (sigma_min ** (1 / rho) - sigma_max ** (1 / rho))) ** rho


# ----------------------------------------------------------------------------- sampler (a-15)

def sampler_edm2(denoise, sample_shape, noises: list, *, num_steps: int, sigma_max: float, sigma_min: float, sigma_data: float = 1.0,
                 rho: float = 7.0, cfg_scale: float = 1.5, use_heun: bool = True, input_perturbation: float = 1.0,
                 input_perturbation_offset: float = 0.0, batch_size: int = 1, conditioned: bool = True,
                 x_ref: Optional[torch.Tensor] = None, seamless_seed: Optional[int] = None):
    """pipelines/dual_diffusion_pipeline.py:589-752 (diffusion_decode), with the random draws injected:
    `noises[0]` is the initial noise, `noises[1:]` the ancestral noise of steps 0..num_steps-2.
    `denoise(x, sigma_vector)` is the UNet call at batch 2B (cond rows first) when `conditioned`, else at batch B;
    with `x_ref` it is called as denoise(x, sigma_vector, ref) (ref repeated to 2B like the sample).
    `seamless_seed`: seamless_loop (:651-658, :729-732) -- per-step random roll (numpy default_rng(seed), as the reference) and
    32 wrapped columns of padding around the sample and the reference input.
    Returns (final sample, sigma schedule list)."""
    import numpy as np
    sched = schedule_edm2(num_steps, sigma_max, sigma_min, rho)
    sig = sched.tolist()
    sample = noises[0] * (sched[0] ** 2 + sigma_data ** 2) ** 0.5
    B = batch_size
    ref = None if x_ref is None else (x_ref.repeat(2, 1, 1, 1) if conditioned else x_ref)
    rng = np.random.default_rng(seamless_seed) if seamless_seed is not None else None

    def guided(x, s):
        extra = () if ref is None else (ref,)
        if conditioned:
            out = denoise(x.repeat(2, 1, 1, 1), torch.tensor([s] * B * 2), *extra).float()
            return torch.lerp(out[B:], out[:B], cfg_scale)
        return denoise(x, torch.tensor([s] * B), *extra).float()

    for i, (s_curr, s_next) in enumerate(zip(sig[:-1], sig[1:])):
        shift = None
        if rng is not None:
            shift = int(rng.integers(0, sample.shape[-1]))
            sample = torch.roll(sample, shifts=shift, dims=-1)
            sample = torch.cat((sample[..., -32:], sample, sample[..., :32]), dim=-1)
            if ref is not None:
                ref = torch.roll(ref, shifts=shift, dims=-1)
                ref = torch.cat((ref[..., -32:], ref, ref[..., :32]), dim=-1)
        old_next = s_next
        ipo = math.log(s_curr) + input_perturbation_offset
        eff = (math.tanh(ipo) / 2 + 0.5) * float(input_perturbation)        # :683-693
        s_next = s_next * (1 - max(min(eff, 1), 0))                          # :695
        out = guided(sample, s_curr)
        if use_heun:                                                         # :705-721
            t_hat = max(old_next, sigma_min) / s_curr
            out_hat = guided(torch.lerp(out, sample, t_hat), t_hat * s_curr)
            out = torch.lerp(out, out_hat, 0.5)
        t = s_next / s_curr if (i + 1) < num_steps else 0
        sample = torch.lerp(out, sample, t)                                  # :723-724
        if shift is not None:                                                # :729-732
            sample = torch.roll(sample[..., 32:-32], shifts=-shift, dims=-1)
            if ref is not None:
                ref = torch.roll(ref[..., 32:-32], shifts=-shift, dims=-1)
        if i + 1 < num_steps:                                                # :734-737
            p = max(old_next ** 2 - s_next ** 2, 0) ** 0.5
            sample = sample + p * noises[1 + i]
    return sample, sig


# ----------------------------------------------------------------------------- VAE (a-14)

DEFAULT_VAE_CFG = dict(in_channels=2, in_num_freqs=256, in_channels_emb=512, out_channels=2, latent_channels=4, dropout=0.0,
                       model_channels=256, channel_mult=(1, 2, 3, 4), channel_mult_emb=None, channels_per_head=64,
                       num_layers_per_block=2, res_balance=0.3, attn_balance=0.3, mlp_multiplier=1, mlp_groups=1,
                       add_mid_block_attention=False, class_id_override=0, target_snr=32.0, label_dim=512)


def vae_cfg(**overrides) -> dict:
    cfg = dict(DEFAULT_VAE_CFG)
    cfg.update(overrides)
    return cfg


def vae_topology(cfg: dict) -> dict:
    """modules/old/vaes/vae_edm2.py:176-228: ordered encoder / decoder stages (no skip connections, no attention unless
    add_mid_block_attention; a block has a skip conv only when its channel count changes, :84)."""
    cblock = [cfg["model_channels"] * m for m in cfg["channel_mult"]]
    cemb = cfg["model_channels"] * cfg["channel_mult_emb"] if cfg["channel_mult_emb"] is not None else max(cblock)
    enc, dec = [], []
    cout = cfg["in_channels"] + 2
    for level, ch in enumerate(cblock):
        if level == 0:
            enc.append(dict(name="conv_in", kind="conv_in", cin=cout, cout=ch))
            cout = ch
        else:
            enc.append(dict(name=f"block{level}_down", kind="block", cin=cout, cout=cout, flavor="enc", resample="down", attention=False))
        for i in range(cfg["num_layers_per_block"]):
            enc.append(dict(name=f"block{level}_layer{i}", kind="block", cin=cout, cout=ch, flavor="enc", resample="keep", attention=False))
            cout = ch
    c_lat = cout
    top = len(cblock) - 1
    for level in range(top, -1, -1):
        ch = cblock[level]
        if level == top:
            for nm in ("in0", "in1"):
                dec.append(dict(name=f"block{level}_{nm}", kind="block", cin=cout, cout=cout, flavor="dec", resample="keep",
                                attention=cfg["add_mid_block_attention"]))
        else:
            dec.append(dict(name=f"block{level}_up", kind="block", cin=cout, cout=cout, flavor="dec", resample="up", attention=False))
        for i in range(cfg["num_layers_per_block"] + 1):
            dec.append(dict(name=f"block{level}_layer{i}", kind="block", cin=cout, cout=ch, flavor="dec", resample="keep", attention=False))
            cout = ch
    return dict(cblock=cblock, cemb=cemb, enc=enc, dec=dec, c_lat=c_lat, cout_last=cout)


def vae_param_shapes(cfg: dict) -> dict:
    topo = vae_topology(cfg)
    g, mm, cemb = cfg["mlp_groups"], cfg["mlp_multiplier"], topo["cemb"]
    shapes = {"latents_out_gain": (), "out_gain": (), "emb_label.weight": (cemb, cfg["label_dim"]),
              "recon_loss_logvar": (1,), "latents_logvar": (1,)}
    for side in ("enc", "dec"):
        for st in topo[side]:
            p = f"{side}.{st['name']}"
            if st["kind"] == "conv_in":
                shapes[f"{p}.weight"] = (st["cout"], st["cin"], 3, 3)
                continue
            cin, cout = st["cin"], st["cout"]
            res0_in = cout if st["flavor"] == "enc" else cin
            shapes[f"{p}.emb_gain"] = ()
            shapes[f"{p}.conv_res0.weight"] = (cout * mm, res0_in // g, 3, 3)
            shapes[f"{p}.conv_res1.weight"] = (cout, cout * mm // g, 3, 3)
            if cin != cout:
                shapes[f"{p}.conv_skip.weight"] = (cout, cin, 1, 1)
            shapes[f"{p}.emb_linear.weight"] = (cout * mm, cemb // g)
            if st["attention"]:
                shapes[f"{p}.emb_gain_qk"] = ()
                shapes[f"{p}.emb_gain_v"] = ()
                shapes[f"{p}.emb_linear_qk.weight"] = (cout, cemb, 1, 1)
                shapes[f"{p}.emb_linear_v.weight"] = (cout, cemb, 1, 1)
                shapes[f"{p}.attn_qk.weight"] = (cout * 2, cout, 1, 1)
                shapes[f"{p}.attn_v.weight"] = (cout, cout, 1, 1)
                shapes[f"{p}.attn_proj.weight"] = (cout, cout, 1, 1)
    shapes["conv_latents_out.weight"] = (cfg["latent_channels"], topo["c_lat"], 3, 3)
    shapes["conv_latents_in.weight"] = (topo["c_lat"], cfg["latent_channels"] + 2, 3, 3)
    shapes["conv_out.weight"] = (cfg["out_channels"], topo["cout_last"], 3, 3)
    return shapes


def random_vae_state(cfg: dict, seed: int, gain_value: float = 0.7) -> dict:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in sorted(vae_param_shapes(cfg).items()):
        if shape == ():
            sd[key] = torch.tensor(gain_value)
        elif key in ("recon_loss_logvar", "latents_logvar"):
            sd[key] = torch.zeros(1)
        else:
            sd[key] = rms_normalize(torch.randn(shape, generator=g))
    return sd


def vae_block_forward(sd: dict, prefix: str, x: torch.Tensor, emb: torch.Tensor, *, flavor: str, resample: str, groups: int,
                      res_balance: float = 0.3, clip: Optional[float] = 256.0, training: bool = False) -> torch.Tensor:
    """modules/old/vaes/vae_edm2.py:96-156 without attention: like the UNet block, but the skip conv exists only when the
    channel count changes (:84) and emb_linear is a plain linear on the (B, cemb) embedding (:87-88, :111-112)."""
    skip = sd.get(f"{prefix}.conv_skip.weight")
    x = resample2x(x, resample)
    if flavor == "enc":
        if skip is not None:
            x = conv_mp(x, skip, training=training)
        x = rms_normalize(x, dims=[1])
    y = conv_mp(silu_mp(x), sd[f"{prefix}.conv_res0.weight"], groups=groups, training=training)
    c = conv_mp(emb, sd[f"{prefix}.emb_linear.weight"], gain=sd[f"{prefix}.emb_gain"], training=training) + 1.0
    y = silu_mp(y * c[:, :, None, None])
    y = conv_mp(y, sd[f"{prefix}.conv_res1.weight"], groups=groups, training=training)
    if flavor == "dec" and skip is not None:
        x = conv_mp(x, skip, training=training)
    x = sum_mp(x, y, res_balance)
    return x.clamp(-clip, clip) if clip is not None else x


def vae_embeddings(sd: dict, labels_like: torch.Tensor) -> torch.Tensor:
    """vae_edm2.py:230-239 with the random draw injected: mp_silu(emb_label(normalize(labels_like)))."""
    return silu_mp(conv_mp(rms_normalize(labels_like.float()), sd["emb_label.weight"]))


def vae_encode(sd: dict, cfg: dict, x: torch.Tensor, emb: torch.Tensor, freq_range=(20.0, 16000.0), training: bool = False):
    """vae_edm2.py:259-269: returns (latent mean, constant noise logvar)."""
    topo = vae_topology(cfg)
    b, _, h, w = x.shape
    x = torch.cat([x, torch.ones_like(x[:, :1]), ln_freq_channel(h, w, b, *freq_range)], dim=1)
    for st in topo["enc"]:
        if st["kind"] == "conv_in":
            x = conv_mp(x, sd["enc.conv_in.weight"], training=training)
        else:
            x = vae_block_forward(sd, f"enc.{st['name']}", x, emb, flavor="enc", resample=st["resample"], groups=cfg["mlp_groups"],
                                  res_balance=cfg["res_balance"], training=training)
    mean = conv_mp(x, sd["conv_latents_out.weight"], gain=sd["latents_out_gain"], training=training)
    return mean, math.log(1 / (cfg["target_snr"] ** 2 + 1))


def vae_decode(sd: dict, cfg: dict, z: torch.Tensor, emb: torch.Tensor, freq_range=(20.0, 16000.0), training: bool = False) -> torch.Tensor:
    """vae_edm2.py:271-279."""
    topo = vae_topology(cfg)
    b, _, h, w = z.shape
    x = torch.cat([z, torch.ones_like(z[:, :1]), ln_freq_channel(h, w, b, *freq_range)], dim=1)
    x = conv_mp(x, sd["conv_latents_in.weight"], training=training)
    for st in topo["dec"]:
        x = vae_block_forward(sd, f"dec.{st['name']}", x, emb, flavor="dec", resample=st["resample"], groups=cfg["mlp_groups"],
                              res_balance=cfg["res_balance"], training=training)
    return conv_mp(x, sd["conv_out.weight"], gain=sd["out_gain"], training=training)
