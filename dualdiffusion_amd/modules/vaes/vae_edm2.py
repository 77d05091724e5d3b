"""EDM2 KL-VAE (mel spectrogram <-> latents) on the HIP kernels.

Drop-in for reference src/modules/old/vaes/vae_edm2.py:151-279 (`AutoencoderKL_EDM2`, the VAE named by the default
`model_index.json`): same config dataclass fields, `state_dict()` keys, `encode / decode / get_embeddings / get_*_shape`
signatures.  Encoder and decoder are launch plans over the same conv / pixel-norm kernels as the UNet; blocks without a
skip conv (channel count unchanged, reference :84) use the stand-alone resample kernel for their residual.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Union

import torch

from ... import ops
from ..._lib import DDXError
from ...engine import PlanBuilder
from ..unets.unet_edm2_b4 import MPConvWeight
from .vae import DualDiffusionVAE, DualDiffusionVAEConfig, IsotropicGaussianDistribution


@dataclass
class DualDiffusionVAE_EDM2Config(DualDiffusionVAEConfig):
    model_channels: int = 256
    channel_mult: list = (1, 2, 3, 4)
    channel_mult_emb: Optional[int] = None
    channels_per_head: int = 64
    num_layers_per_block: int = 2
    res_balance: float = 0.3
    attn_balance: float = 0.3
    mlp_multiplier: int = 1
    mlp_groups: int = 1
    add_mid_block_attention: bool = False
    class_id_override: Optional[int] = 0
    target_snr: float = 32
    label_dim: int = 512


class VAEBlockWeights(torch.nn.Module):
    """Parameters of one VAE block (reference vae_edm2.py:53-94): skip conv only when the channel count changes,
    emb_linear is a plain linear layer."""

    def __init__(self, level: int, in_channels: int, out_channels: int, emb_channels: int, flavor: str, resample_mode: str,
                 use_attention: bool, cfg: DualDiffusionVAE_EDM2Config):
        super().__init__()
        self.level, self.in_channels, self.out_channels = level, in_channels, out_channels
        self.flavor, self.resample_mode, self.use_attention = flavor, resample_mode, use_attention
        self.num_heads = out_channels // cfg.channels_per_head
        mm, g = cfg.mlp_multiplier, cfg.mlp_groups
        self.conv_res0 = MPConvWeight(out_channels if flavor == "enc" else in_channels, out_channels * mm, (3, 3), groups=g)
        self.conv_res1 = MPConvWeight(out_channels * mm, out_channels, (3, 3), groups=g)
        self.conv_skip = MPConvWeight(in_channels, out_channels, (1, 1)) if in_channels != out_channels else None
        self.emb_gain = torch.nn.Parameter(torch.zeros([]))
        self.emb_linear = MPConvWeight(emb_channels, out_channels * mm, (), groups=g)
        if use_attention:
            self.emb_gain_qk = torch.nn.Parameter(torch.zeros([]))
            self.emb_gain_v = torch.nn.Parameter(torch.zeros([]))
            self.emb_linear_qk = MPConvWeight(emb_channels, out_channels, (1, 1))
            self.emb_linear_v = MPConvWeight(emb_channels, out_channels, (1, 1))
            self.attn_qk = MPConvWeight(out_channels, out_channels * 2, (1, 1))
            self.attn_v = MPConvWeight(out_channels, out_channels, (1, 1))
            self.attn_proj = MPConvWeight(out_channels, out_channels, (1, 1))


class AutoencoderKL_EDM2(DualDiffusionVAE):

    config_class = DualDiffusionVAE_EDM2Config

    def __init__(self, config: DualDiffusionVAE_EDM2Config) -> None:
        super().__init__()
        self.config = config
        cblock = [config.model_channels * m for m in config.channel_mult]
        cemb = config.model_channels * config.channel_mult_emb if config.channel_mult_emb is not None else max(cblock)
        self.num_levels = len(config.channel_mult)
        noise_std = (1 / (config.target_snr ** 2 + 1)) ** 0.5
        self.latents_out_gain = torch.nn.Parameter(torch.tensor((1 - noise_std ** 2) ** 0.5))
        self.out_gain = torch.nn.Parameter(torch.ones([]))
        self.emb_label = MPConvWeight(config.label_dim, cemb, ())
        self.emb_dim = cemb
        self.recon_loss_logvar = torch.nn.Parameter(torch.zeros(1))
        self.latents_logvar = torch.nn.Parameter(torch.zeros(1))

        self.enc = torch.nn.ModuleDict()
        cout = config.in_channels + 2
        for level, ch in enumerate(cblock):
            if level == 0:
                self.enc["conv_in"] = MPConvWeight(cout, ch, (3, 3))
                cout = ch
            else:
                self.enc[f"block{level}_down"] = VAEBlockWeights(level, cout, cout, cemb, "enc", "down", False, config)
            for i in range(config.num_layers_per_block):
                self.enc[f"block{level}_layer{i}"] = VAEBlockWeights(level, cout, ch, cemb, "enc", "keep", False, config)
                cout = ch
        self.conv_latents_out = MPConvWeight(cout, config.latent_channels, (3, 3))
        self.conv_latents_in = MPConvWeight(config.latent_channels + 2, cout, (3, 3))
        self.dec = torch.nn.ModuleDict()
        for level, ch in reversed(list(enumerate(cblock))):
            if level == len(cblock) - 1:
                for nm in ("in0", "in1"):
                    self.dec[f"block{level}_{nm}"] = VAEBlockWeights(level, cout, cout, cemb, "dec", "keep",
                                                                     config.add_mid_block_attention, config)
            else:
                self.dec[f"block{level}_up"] = VAEBlockWeights(level, cout, cout, cemb, "dec", "up", False, config)
            for i in range(config.num_layers_per_block + 1):
                self.dec[f"block{level}_layer{i}"] = VAEBlockWeights(level, cout, ch, cemb, "dec", "keep", False, config)
                cout = ch
        self.conv_out = MPConvWeight(cout, config.out_channels, (3, 3))
        self._engines: dict = {}
        self._use_graph = False

    def _on_placement_change(self) -> None:
        self._engines = {}

    def _require_device(self) -> None:
        if self.device.type != "cuda":
            raise DDXError("VAE is not on a ROCm device: dualdiffusion_amd runs only on its HIP kernels (no CPU fallback)")

    # ------------------------------------------------------------------ reference API
    @torch.no_grad()
    def get_embeddings(self, class_labels: torch.Tensor, labels_like: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference vae_edm2.py:230-239: the labels are REPLACED by a random draw of the same shape (by design there);
        `labels_like` injects that draw for reproducible tests."""
        self._require_device()
        if class_labels.shape[-1] == 512 and self.config.label_dim != 512:
            class_labels = torch.zeros((class_labels.shape[0], self.config.label_dim))
        dev = self.device
        draw = labels_like if labels_like is not None else torch.randn(class_labels.shape, device=dev)
        x = ops.pixelnorm(draw.to(device=dev, dtype=torch.float32).contiguous())
        B = x.shape[0]
        lin = torch.empty(B, self.emb_dim, device=dev, dtype=torch.float32)
        tab = ops.make_linear_jobs([(self.emb_label.weight, None, lin, 1.0, 0.0, 1, self.training)], dev)
        ops.linear_small(tab, 1, self.emb_dim, x, B, self.emb_label.weight.dtype)
        out = torch.empty_like(lin)
        ops.mpsum_rows(lin, lin, out, t=0.0, silu=True)       # mp_silu
        torch.cuda.current_stream().synchronize()             # the job table is a temporary
        return out.to(self.dtype)

    def get_recon_loss_logvar(self) -> torch.Tensor:
        return self.recon_loss_logvar

    def get_target_snr(self) -> float:
        return self.config.target_snr

    def get_latent_shape(self, sample_shape: Union[torch.Size, tuple]) -> torch.Size:
        if len(sample_shape) != 4:
            raise ValueError(f"Invalid sample shape: {sample_shape}")
        q = 2 ** (self.num_levels - 1)
        return torch.Size((sample_shape[0], self.config.latent_channels, sample_shape[2] // q, sample_shape[3] // q))

    def get_sample_shape(self, latent_shape: Union[torch.Size, tuple]) -> torch.Size:
        if len(latent_shape) != 4:
            raise ValueError(f"Invalid latent shape: {latent_shape}")
        q = 2 ** (self.num_levels - 1)
        return torch.Size((latent_shape[0], self.config.out_channels, latent_shape[2] * q, latent_shape[3] * q))

    def _engine(self, kind: str, shape) -> "_VAEEngine":
        self._require_device()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise DDXError("autograd through the HIP VAE is not available yet: call under torch.no_grad()")
        B, _, H, W = shape
        key = (kind, B, H, W, self.dtype, self.training)
        eng = self._engines.get(key)
        if eng is None:
            eng = self._engines[key] = _VAEEngine(self, kind, B, H, W, self.training)
        return eng

    # A launch plan keeps every activation of its batch alive (~17 GB per 45 s sample for the decoder at the mel resolution):
    # larger batches run as chunks of this many samples through one plan (config 5: B = 16 -> 4 chunks).
    max_plan_batch = 4

    def _run_chunked(self, kind: str, x: torch.Tensor, emb: torch.Tensor, format) -> torch.Tensor:
        B, n = x.shape[0], self.max_plan_batch
        if B <= n:
            return self._engine(kind, x.shape).run(x, emb, format, self._use_graph)
        outs = []
        for i in range(0, B, n):
            xc, ec = x[i:i + n], emb[i:i + n]
            outs.append(self._engine(kind, xc.shape).run(xc, ec, format, self._use_graph, full_batch=B))
        return torch.cat(outs, dim=0)

    def encode(self, x: torch.Tensor, class_embeddings: torch.Tensor, format) -> IsotropicGaussianDistribution:
        """reference vae_edm2.py:259-269."""
        if torch.compiler.is_compiling():
            from ... import compile_ops
            mean = compile_ops.vae_encode(x, class_embeddings, compile_ops.handle_of(self), compile_ops.handle_of(format))
        else:
            mean = self._run_chunked("enc", x, class_embeddings, format)
        logvar = torch.tensor(math.log(1 / (self.config.target_snr ** 2 + 1)), device=mean.device, dtype=mean.dtype)
        return IsotropicGaussianDistribution(mean, logvar)

    def decode(self, x: torch.Tensor, class_embeddings: torch.Tensor, format) -> torch.Tensor:
        """reference vae_edm2.py:271-279."""
        if torch.compiler.is_compiling():
            from ... import compile_ops
            return compile_ops.vae_decode(x, class_embeddings, compile_ops.handle_of(self), compile_ops.handle_of(format))
        return self._run_chunked("dec", x, class_embeddings, format)

    def forward(self, x, class_embeddings, format):
        return self.decode(self.encode(x, class_embeddings, format).mode(), class_embeddings, format)


class _VAEEngine:
    """Launch plan of VAE.encode or VAE.decode for a fixed (B, H, W, dtype, training)."""

    def __init__(self, vae: AutoencoderKL_EDM2, kind: str, B: int, H: int, W: int, training: bool):
        cfg = vae.config
        q = 2 ** (vae.num_levels - 1)
        if kind == "enc" and (H % q or W % q):
            raise DDXError(f"sample size {H}x{W} must be a multiple of {q}")
        self.v, self.B, self.H, self.W = vae, B, H, W
        pb = self.pb = PlanBuilder(vae.device, vae.dtype, B, training)
        self._lnf_key = None
        cin = cfg.in_channels if kind == "enc" else cfg.latent_channels
        self.x_in = pb.f32(B, cin, H, W)
        self.emb = pb.f32(B, vae.emb_dim)
        self.lnf = pb.f32(H)
        self.zero_sigma = torch.zeros(B, device=vae.device, dtype=torch.float32)   # c_in(0) = 1: plain copy + extra channels
        pb.keep.append(self.zero_sigma)
        Cpad = 8
        x0 = pb.act(H, W, Cpad)
        bk = dict(mlp_multiplier=cfg.mlp_multiplier, res_balance=cfg.res_balance, attn_balance=cfg.attn_balance)
        h, w = H, W
        if kind == "enc":
            first, blocks, last, last_gain = vae.enc["conv_in"], [b for n, b in vae.enc.items() if n != "conv_in"], vae.conv_latents_out, vae.latents_out_gain
            names = [n for n in vae.enc if n != "conv_in"]
        else:
            first, blocks, last, last_gain = vae.conv_latents_in, list(vae.dec.values()), vae.conv_out, vae.out_gain
            names = list(vae.dec)
        self.stages: dict = {}      # "enc.<block>" / "dec.<block>" -> NHWC output buffer (static: valid after any run)
        pw_first = pb.prep(first, cg_pad=Cpad, npix=B * H * W)
        x = pb.act(H, W, first.out_channels)
        # decoder blocks read mp_silu(x) twins written by their producers; encoder blocks normalise first and need none
        x_act = pb.act(H, W, first.out_channels) if kind == "dec" else None
        first_kw = dict(out2=x_act, out2_scale=1.0) if x_act is not None else {}       # bound now: x_act is re-assigned below
        pb.step(lambda x=x, first_kw=first_kw: ops.conv2d(x0, pw_first, out=x, **first_kw))
        for bi, blk in enumerate(blocks):
            if blk.resample_mode == "down":
                h, w = h // 2, w // 2
            elif blk.resample_mode == "up":
                h, w = h * 2, w * 2
            tw = 1.0 if (kind == "dec" and bi + 1 < len(blocks)) else None
            x, x_act = pb.block(blk, x, None, 1.0, 1.0, h, w, act0=x_act, twin_scale=tw, **bk)
            self.stages[f"{kind}.{names[bi]}"] = x
        # a 2-channel conv_out would fall to the scalar kernel: prepare it with zero rows up to 8 and read back the real ones
        cout = last.out_channels
        cpad = cout if cout % 4 == 0 else (cout + 7) // 8 * 8
        pw_last = pb.prep(last, gain_param=last_gain, npix=B * h * w, cout_pad=cpad)
        y = pb.act(h, w, cpad)
        self.out = pb.f32(B, cout, h, w)
        pb.step(lambda x=x: ops.conv2d(x, pw_last, out=y))
        pb.step(lambda: ops.nhwc_to_nchw(y, out=self.out, channels=cout))
        pb.finalize(self.emb, vae.emb_dim, pre_steps=lambda: ops.unet_input_prep(self.x_in, self.zero_sigma, self.lnf, x0, 1.0))

    def run(self, x, emb, format, use_graph: bool, full_batch: Optional[int] = None) -> torch.Tensor:
        # the reference standardises ln_freqs over the WHOLE batch tensor (unbiased std: depends on the element count), so a
        # chunk of a larger batch asks for the table of the full batch size
        nb = full_batch or self.B
        fs = getattr(format, "freq_scale", None)          # by value: an id() can be recycled by a different format object
        lkey = (type(format).__name__, type(fs).__name__, getattr(fs, "freq_scale", None), float(getattr(fs, "freq_min", 0.0)),
                float(getattr(fs, "freq_max", 0.0)), int(getattr(fs, "num_filters", 0)), nb) if fs is not None else (id(format), nb)
        if lkey != self._lnf_key:
            rows = format.get_ln_freqs(torch.empty(nb, 1, self.H, self.W))[0, 0, :, 0]   # host-side table, same dtype sequence
            self.lnf.copy_(rows.float())
            self._lnf_key = lkey
        self.pb.refresh_weights(self.v.parameters())
        self.x_in.copy_(x)
        self.emb.copy_(emb)
        self.pb.launch(use_graph)
        return self.out.clone()
