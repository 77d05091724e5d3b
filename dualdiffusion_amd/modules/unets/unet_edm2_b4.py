"""EDM2 UNet (default dualdiffusion denoiser) executed by hand-written HIP kernels on MI355X.

Drop-in for reference src/modules/unets/unet_edm2_b4.py: same `UNetConfig` fields, constructor `UNet(config)`,
method signatures and `state_dict()` keys, so checkpoints and callers are interchangeable.
What differs is everything underneath: the module holds only parameters; `forward` replays a launch plan of
libddx_hip kernels (NHWC activations, fused prologue/epilogue convs, flash attention) compiled once per input
shape, optionally as a single hipGraph.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Union

import torch

from ... import ops
from ..._lib import DDXError, Plan
from ..._lib import EPI_MPSUM, PRO_NONE, PRO_SCALE, PRO_SCALE_SILU, PRO_SILU, RESAMPLE_DOWN, RESAMPLE_KEEP, RESAMPLE_UP  # noqa: F401
from .unet import DualDiffusionUNet, DualDiffusionUNetConfig


@dataclass
class UNetConfig(DualDiffusionUNetConfig):
    model_channels: int = 256
    logvar_channels: int = 128
    channel_mult: list = (1, 2, 3, 4, 5)
    channel_mult_noise: Optional[int] = None
    channel_mult_emb: Optional[int] = None
    channels_per_head: int = 64
    num_layers_per_block: int = 2
    label_balance: float = 0.5
    concat_balance: float = 0.5
    res_balance: float = 0.3
    attn_balance: float = 0.3
    attn_levels: list = (3, 4)
    mlp_multiplier: int = 2
    mlp_groups: int = 8


class MPConvWeight(torch.nn.Module):
    """Parameter holder for one magnitude-preserving conv / linear layer (reference MPConv, mp_tools.py:332-378).
    Key name `weight`, init randn, attribute `weight.conv_groups` as in the reference."""

    def __init__(self, in_channels: int, out_channels: int, kernel: tuple, groups: int = 1, disable_weight_norm: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.groups = in_channels, out_channels, groups
        self.disable_weight_norm = disable_weight_norm
        self.weight = torch.nn.Parameter(torch.randn(out_channels, in_channels // groups, *kernel))
        self.weight.conv_groups = groups

    @torch.no_grad()
    def normalize_weights(self) -> None:
        if not self.disable_weight_norm:
            if not self.weight.is_cuda:
                raise DDXError("normalize_weights needs the module on a ROCm device (no CPU path)")
            ops.normalize_weights_(self.weight.data)


class FourierTable(torch.nn.Module):
    """Buffers of MPFourier (mp_tools.py:316-322)."""

    def __init__(self, num_channels: int, bandwidth: float = 1., eps: float = 1e-3):
        super().__init__()
        self.register_buffer("freqs", torch.pi * torch.linspace(0, 1 - eps, num_channels).erfinv() * bandwidth)
        self.register_buffer("phases", torch.pi / 2 * (torch.arange(num_channels) % 2 == 0).float())


class BlockWeights(torch.nn.Module):
    """Parameters of one UNet block (reference Block.__init__, unet_edm2_b4.py:62-108)."""

    def __init__(self, level: int, in_channels: int, out_channels: int, emb_channels: int, flavor: str, resample_mode: str,
                 use_attention: bool, cfg: UNetConfig):
        super().__init__()
        self.level, self.in_channels, self.out_channels = level, in_channels, out_channels
        self.flavor, self.resample_mode, self.use_attention = flavor, resample_mode, use_attention
        self.num_heads = out_channels // cfg.channels_per_head
        mm, g = cfg.mlp_multiplier, cfg.mlp_groups
        self.conv_res0 = MPConvWeight(out_channels if flavor == "enc" else in_channels, out_channels * mm, (3, 3), groups=g)
        self.conv_res1 = MPConvWeight(out_channels * mm, out_channels, (3, 3), groups=g)
        self.conv_skip = MPConvWeight(in_channels, out_channels, (1, 1))
        self.emb_gain = torch.nn.Parameter(torch.zeros([]))
        self.emb_linear = MPConvWeight(emb_channels, out_channels * mm, (1, 1), groups=g)
        if use_attention:
            self.emb_gain_qk = torch.nn.Parameter(torch.zeros([]))
            self.emb_gain_v = torch.nn.Parameter(torch.zeros([]))
            self.emb_linear_qk = MPConvWeight(emb_channels, out_channels, (1, 1))
            self.emb_linear_v = MPConvWeight(emb_channels, out_channels, (1, 1))
            self.attn_qk = MPConvWeight(out_channels, out_channels * 2, (1, 1))
            self.attn_v = MPConvWeight(out_channels, out_channels, (1, 1))
            self.attn_proj = MPConvWeight(out_channels, out_channels, (1, 1))


class UNet(DualDiffusionUNet):

    config_class = UNetConfig

    def __init__(self, config: UNetConfig) -> None:
        super().__init__()
        self.config = config
        cblock = [config.model_channels * m for m in config.channel_mult]
        cnoise = config.model_channels * config.channel_mult_noise if config.channel_mult_noise is not None else max(cblock)
        cemb = config.model_channels * config.channel_mult_emb if config.channel_mult_emb is not None else max(cblock)
        self.num_levels = len(config.channel_mult)
        self.cnoise, self.cemb = cnoise, cemb

        self.emb_fourier = FourierTable(cnoise)
        self.emb_noise = MPConvWeight(cnoise, cemb, ())
        self.emb_label = MPConvWeight(config.in_channels_emb, cemb, ())
        self.emb_label_unconditional = MPConvWeight(1, cemb, ())
        self.logvar_fourier = FourierTable(config.logvar_channels)
        self.logvar_linear = MPConvWeight(config.logvar_channels, 1, (), disable_weight_norm=True)

        attn = set(config.attn_levels)
        self.enc = torch.nn.ModuleDict()
        cout = config.in_channels + 2
        for level, ch in enumerate(cblock):
            if level == 0:
                self.enc["conv_in"] = MPConvWeight(cout, ch, (3, 3))
                cout = ch
            else:
                self.enc[f"block{level}_down"] = BlockWeights(level, cout, cout, cemb, "enc", "down", level in attn, config)
            for i in range(config.num_layers_per_block):
                self.enc[f"block{level}_layer{i}"] = BlockWeights(level, cout, ch, cemb, "enc", "keep", level in attn, config)
                cout = ch
        skips = [m.out_channels for m in self.enc.values()]
        self.dec = torch.nn.ModuleDict()
        for level, ch in reversed(list(enumerate(cblock))):
            if level == len(cblock) - 1:
                self.dec[f"block{level}_in0"] = BlockWeights(level, cout, cout, cemb, "dec", "keep", True, config)
                self.dec[f"block{level}_in1"] = BlockWeights(level, cout, cout, cemb, "dec", "keep", True, config)
            else:
                self.dec[f"block{level}_up"] = BlockWeights(level, cout, cout, cemb, "dec", "up", level in attn, config)
            for i in range(config.num_layers_per_block + 1):
                cin = cout + skips.pop()
                self.dec[f"block{level}_layer{i}"] = BlockWeights(level, cin, ch, cemb, "dec", "keep", level in attn, config)
                cout = ch
        self.out_gain = torch.nn.Parameter(torch.zeros([]))
        self.conv_out = MPConvWeight(cout, config.out_channels, (3, 3))

        self._engines: dict = {}
        self._use_graph = False
        self._small: Optional["_SmallOps"] = None

    # ------------------------------------------------------------------ reference API
    def _on_placement_change(self) -> None:
        self._engines = {}
        self._small = None

    def _require_device(self) -> None:
        if self.device.type != "cuda":
            raise DDXError("UNet is not on a ROCm device: dualdiffusion_amd runs only on its HIP kernels (no CPU fallback)")

    @torch.no_grad()
    def get_embeddings(self, emb_in: torch.Tensor, conditioning_mask: torch.Tensor) -> torch.Tensor:
        """reference unet_edm2_b4.py:232-235."""
        self._require_device()
        if self._small is None:
            self._small = _SmallOps(self)
        return self._small.embeddings(emb_in, conditioning_mask)

    @torch.no_grad()
    def get_sigma_loss_logvar(self, sigma: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference unet_edm2_b4.py:237-238."""
        self._require_device()
        if self._small is None:
            self._small = _SmallOps(self)
        return self._small.logvar(sigma)

    def get_latent_shape(self, latent_shape: Union[torch.Size, tuple]) -> torch.Size:
        q = 2 ** (self.num_levels - 1)
        return torch.Size(tuple(latent_shape[0:2]) + ((latent_shape[2] // q) * q, (latent_shape[3] // q) * q))

    @torch.no_grad()
    def get_ln_freqs_rows(self, format, B: int, H: int, W: int) -> torch.Tensor:
        """Per-row values of the frequency-axis embedding channel (reference get_ln_freqs, unet_edm2_b4.py:244-248),
        computed once per shape on the host exactly as the reference does (same dtype sequence, same global mean / unbiased std)."""
        ln = format.ms_freq_scale.get_unscaled(H + 2, device="cpu")[1:-1].log2()
        full = ln.view(1, 1, -1, 1).repeat(B, 1, 1, W)
        return ((ln - full.mean()) / full.std()).float().contiguous()

    def forward(self, x_in: torch.Tensor, sigma: torch.Tensor, format, embeddings: torch.Tensor,
                x_ref: Optional[torch.Tensor] = None, perturbed_input: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference unet_edm2_b4.py:250-296.  Returns float32 NCHW like the reference."""
        self._require_device()
        if torch.is_grad_enabled() and any(p.requires_grad for p in self.parameters()):
            raise DDXError("autograd through the HIP UNet is not available yet: call under torch.no_grad()")
        B, _, H, W = x_in.shape
        key = (B, H, W, self.dtype, self.training, x_ref is not None)
        eng = self._engines.get(key)
        if eng is None:
            eng = _UNetEngine(self, B, H, W, self.training, x_ref is not None)
            self._engines[key] = eng
        return eng.run(x_in, sigma, format, embeddings, x_ref, perturbed_input, use_graph=self._use_graph)


# ====================================================================================================== engines

class _SmallOps:
    """Eager helpers for the small host-called methods (get_embeddings, get_sigma_loss_logvar)."""

    def __init__(self, unet: UNet):
        self.u = unet

    def embeddings(self, emb_in: torch.Tensor, mask: torch.Tensor) -> torch.Tensor:
        u, dev = self.u, self.u.device
        B = emb_in.shape[0]
        x = emb_in.to(device=dev, dtype=torch.float32).contiguous()
        xn = ops.pixelnorm(x)
        ones = torch.ones(1, 1, device=dev, dtype=torch.float32)
        uemb = torch.empty(1, u.cemb, device=dev, dtype=torch.float32)
        cemb = torch.empty(B, u.cemb, device=dev, dtype=torch.float32)
        t1 = ops.make_linear_jobs([(u.emb_label_unconditional.weight, None, uemb, 1.0, 0.0, 1, u.training)], dev)
        ops.linear_small(t1, 1, u.cemb, ones, 1, u.emb_label_unconditional.weight.dtype)
        t2 = ops.make_linear_jobs([(u.emb_label.weight, None, cemb, 1.0, 0.0, 1, u.training)], dev)
        ops.linear_small(t2, 1, u.cemb, xn, B, u.emb_label.weight.dtype)
        out = torch.empty(B, u.cemb, device=dev, dtype=torch.float32)
        ops.mpsum_rows(uemb, cemb, out, t_rows=mask.to(device=dev, dtype=torch.float32).contiguous())
        torch.cuda.current_stream().synchronize()  # job tables are temporaries
        return out.to(u.dtype)

    def logvar(self, sigma: torch.Tensor) -> torch.Tensor:
        u, dev = self.u, self.u.device
        s = sigma.flatten().to(device=dev, dtype=torch.float32).contiguous()
        B = s.numel()
        f = torch.empty(B, u.config.logvar_channels, device=dev, dtype=torch.float32)
        ops.mpfourier(s, u.logvar_fourier.freqs.float(), u.logvar_fourier.phases.float(), f, True)
        out = torch.empty(B, 1, device=dev, dtype=torch.float32)
        t = ops.make_linear_jobs([(u.logvar_linear.weight, None, out, 1.0, 0.0, 1, False)], dev)
        ops.linear_small(t, 1, 1, f, B, u.logvar_linear.weight.dtype)
        torch.cuda.current_stream().synchronize()
        return out.view(-1, 1, 1, 1)


class _UNetEngine:
    """Launch plan of one UNet forward for a fixed (B, H, W, dtype, training) -- built once, replayed per call."""

    def __init__(self, unet: UNet, B: int, H: int, W: int, training: bool, with_xref: bool):
        cfg = unet.config
        q = 2 ** (unet.num_levels - 1)
        if H % q or W % q:
            raise DDXError(f"latent size {H}x{W} must be a multiple of {q} (see get_latent_shape)")
        self.u, self.B, self.H, self.W, self.training = unet, B, H, W, training
        self.dev, self.dt = unet.device, unet.dtype
        dev, dt = self.dev, self.dt
        f32 = dict(device=dev, dtype=torch.float32)
        self.keep: list = []
        self._weights_key = None
        self._lnf_key = None

        # ---- static I/O
        C_in = cfg.in_channels
        self.x_in = torch.empty(B, C_in, H, W, **f32)
        self.x_pre = torch.empty(B, C_in, H, W, **f32)       # perturbed_input or x_in: what the body sees
        self.sigma = torch.empty(B, **f32)
        self.emb_in = torch.empty(B, unet.cemb, **f32)
        self.x_ref = torch.empty(B, cfg.out_channels + 1, H, W, **f32) if with_xref else None
        self.out = torch.empty(B, cfg.out_channels, H, W, **f32)
        self.lnf = torch.empty(H, **f32)

        # ---- weight preparation plan
        self.gains: list = []          # 0-d gain parameters, mirrored into one fp32 vector for the kernels
        self.convs: list = []          # (MPConvWeight, PreparedWeight holder dict)
        self.wplan = Plan()
        self.fplan = Plan()
        self._build(with_xref)

    # -------------------------------------------------------------------------------------------- helpers
    def _gain_slot(self, p: torch.nn.Parameter) -> int:
        self.gains.append(p)
        return len(self.gains) - 1

    def _act(self, Hh: int, Ww: int, Cc: int) -> torch.Tensor:
        t = torch.empty(self.B, Hh, Ww, Cc, device=self.dev, dtype=self.dt)
        self.keep.append(t)
        return t

    def _prep(self, conv: MPConvWeight, gain_param=None, qk_head_dim: int = 0, cg_pad: Optional[int] = None, npix: int = 0):
        """Allocate the prepared-weight buffer and queue its wprep; returns a PreparedWeight (filled when wplan runs)."""
        w = conv.weight
        Cg = w.shape[1]
        ks = w.shape[2]
        CK = ops.pick_ck(cg_pad or Cg, ks, self.dt, npix)
        nbytes = ops.lib().ddx_wprep_bytes(w.shape[0], Cg, ks, conv.groups, CK, ops.dtype_code(self.dt))
        buf = torch.empty(nbytes, dtype=torch.uint8, device=self.dev)
        self.keep.append(buf)
        spec = dict(conv=conv, buf=buf, CK=CK, qk=qk_head_dim, cg_pad=cg_pad,
                    gain_slot=self._gain_slot(gain_param) if gain_param is not None else None)
        self.convs.append(spec)
        return ops.PreparedWeight(buf, w.shape[0], Cg, ks, conv.groups, CK, self.dt, None)

    # -------------------------------------------------------------------------------------------- plan build
    def _build(self, with_xref: bool) -> None:
        u, cfg, B, H, W = self.u, self.u.config, self.B, self.H, self.W
        dev = self.dev
        f32 = dict(device=dev, dtype=torch.float32)
        cemb = u.cemb
        wa_wb = lambda na, nb, t: ((math.sqrt((na + nb) / ((1 - t) ** 2 + t ** 2)) / math.sqrt(na) * (1 - t)),
                                   (math.sqrt((na + nb) / ((1 - t) ** 2 + t ** 2)) / math.sqrt(nb) * t))

        # prepared weights + per-block emb_linear jobs are declared while walking the topology
        lin_jobs: list = []   # (MPConvWeight, gain_slot, out tensor, groups)

        def cvec(conv: MPConvWeight, gain_param) -> torch.Tensor:
            out = torch.empty(B, conv.out_channels, **f32)
            self.keep.append(out)
            lin_jobs.append((conv, self._gain_slot(gain_param), out, conv.groups))
            return out

        steps: list = []  # closures executed inside fplan.record()

        # ---- front end
        Cpad = 8
        x0 = self._act(H, W, Cpad)
        four = torch.empty(B, u.cnoise, **f32)
        e0 = torch.empty(B, cemb, **f32)
        emb = torch.empty(B, cemb, **f32)
        self.keep += [four, e0, emb]
        self.emb = emb
        freqs = u.emb_fourier.freqs.float().contiguous()
        phases = u.emb_fourier.phases.float().contiguous()
        self.keep += [freqs, phases]
        self._noise_job = (u.emb_noise, e0)

        pw_in = self._prep(u.enc["conv_in"], cg_pad=Cpad)

        def front():
            ops.unet_input_prep(self.x_pre, self.sigma, self.lnf, x0, cfg.sigma_data)
            ops.mpfourier(self.sigma, freqs, phases, four, True)
            ops.linear_small(self.noise_table, 1, cemb, four, B, u.emb_noise.weight.dtype)
            ops.mpsum_rows(e0, self.emb_in, emb, t=cfg.label_balance, silu=True)
            ops.linear_small(self.emb_table, self.emb_njobs, self.emb_max_o, emb, B, self.emb_wdtype, x_stride=cemb)
        steps.append(front)

        # ---- encoder
        x = self._act(H, W, u.enc["conv_in"].out_channels)
        steps.append(lambda x=x: ops.conv2d(x0, pw_in, out=x))
        skips = [x]
        cur_h, cur_w = H, W

        def block(blk: BlockWeights, src0, src1, s0, s1, h, w):
            """Queue one Block (reference unet_edm2_b4.py:110-158); returns the output tensor."""
            cout = blk.out_channels
            mm = cfg.mlp_multiplier
            rs = {"keep": RESAMPLE_KEEP, "up": RESAMPLE_UP, "down": RESAMPLE_DOWN}[blk.resample_mode]
            c_emb = cvec(blk.emb_linear, blk.emb_gain)
            npix = B * h * w
            pw_res0, pw_res1, pw_skip = (self._prep(blk.conv_res0, npix=npix), self._prep(blk.conv_res1, npix=npix),
                                         self._prep(blk.conv_skip, npix=npix))
            y0 = self._act(h, w, cout * mm)
            xo = self._act(h, w, cout)
            last_clip = 0.0 if blk.use_attention else 256.0
            if blk.flavor == "enc":
                x1 = self._act(h, w, cout)
                steps.append(lambda: ops.conv2d(src0, pw_skip, out_hw=(h, w), resample=rs, out=x1))
                steps.append(lambda: ops.pixelnorm(x1, out=x1))
                steps.append(lambda: ops.conv2d(x1, pw_res0, prologue=PRO_SILU, out=y0))
                steps.append(lambda: ops.conv2d(y0, pw_res1, prologue=PRO_SCALE_SILU, chan_scale=c_emb, residual=x1,
                                                res_t=cfg.res_balance, clip=last_clip, out=xo))
            else:
                sk = self._act(h, w, cout)
                steps.append(lambda: ops.conv2d(src0, pw_res0, out_hw=(h, w), src1=src1, scale0=s0, scale1=s1, resample=rs,
                                                prologue=PRO_SILU, out=y0))
                steps.append(lambda: ops.conv2d(src0, pw_skip, out_hw=(h, w), src1=src1, scale0=s0, scale1=s1, resample=rs, out=sk))
                steps.append(lambda: ops.conv2d(y0, pw_res1, prologue=PRO_SCALE_SILU, chan_scale=c_emb, residual=sk,
                                                res_t=cfg.res_balance, clip=last_clip, out=xo))
            if not blk.use_attention:
                return xo
            c_qk, c_v = cvec(blk.emb_linear_qk, blk.emb_gain_qk), cvec(blk.emb_linear_v, blk.emb_gain_v)
            hd = cout // blk.num_heads
            pw_qk = self._prep(blk.attn_qk, qk_head_dim=hd, npix=npix)
            pw_v, pw_proj = self._prep(blk.attn_v, npix=npix), self._prep(blk.attn_proj, npix=npix)
            qk, vv, ao, xa = self._act(h, w, 2 * cout), self._act(h, w, cout), self._act(h, w, cout), self._act(h, w, cout)
            steps.append(lambda: ops.conv2d(xo, pw_qk, prologue=PRO_SCALE, chan_scale=c_qk, out=qk))
            steps.append(lambda: ops.conv2d(xo, pw_v, out=vv))
            steps.append(lambda: ops.attention(qk, vv, blk.num_heads, out=ao))
            steps.append(lambda: ops.conv2d(ao, pw_proj, prologue=PRO_SCALE_SILU, chan_scale=c_v, residual=xo,
                                            res_t=cfg.attn_balance, clip=256.0, out=xa))
            return xa

        for name, blk in u.enc.items():
            if name == "conv_in":
                continue
            if blk.resample_mode == "down":
                cur_h, cur_w = cur_h // 2, cur_w // 2
            x = block(blk, x, None, 1.0, 1.0, cur_h, cur_w)
            skips.append(x)
        # ---- decoder
        for name, blk in u.dec.items():
            if blk.resample_mode == "up":
                cur_h, cur_w = cur_h * 2, cur_w * 2
            if "layer" in name:
                sk = skips.pop()
                s0, s1 = wa_wb(x.shape[3], sk.shape[3], cfg.concat_balance)
                x = block(blk, x, sk, s0, s1, cur_h, cur_w)
            else:
                x = block(blk, x, None, 1.0, 1.0, cur_h, cur_w)
        # ---- output
        pw_out = self._prep(u.conv_out, gain_param=u.out_gain)
        y = self._act(H, W, cfg.out_channels)
        xl = x
        steps.append(lambda: ops.conv2d(xl, pw_out, out=y))
        steps.append(lambda: ops.unet_output_combine(y, self.x_in, self.sigma, self.x_ref, self.out, cfg.sigma_data))

        # ---- gains vector + job tables
        self.gain_f32 = torch.zeros(max(len(self.gains), 1), **f32)
        gp = lambda slot: self.gain_f32[slot:slot + 1] if slot is not None else None
        self.noise_table = ops.make_linear_jobs([(u.emb_noise.weight, None, e0, 1.0, 0.0, 1, self.training)], dev)
        self.emb_table = ops.make_linear_jobs(
            [(c.weight, gp(slot), out, 1.0, 1.0, groups, self.training) for (c, slot, out, groups) in lin_jobs], dev)
        self.emb_njobs = len(lin_jobs)
        self.emb_max_o = max(c.out_channels for (c, _, _, _) in lin_jobs)
        self.emb_wdtype = lin_jobs[0][0].weight.dtype

        with self.wplan.record():
            for sp in self.convs:
                conv = sp["conv"]
                ops.wprep(conv.weight, conv.groups, self.dt, gain_ptr=gp(sp["gain_slot"]),
                          normalize=self.training and not conv.disable_weight_norm, qk_head_dim=sp["qk"], CK=sp["CK"],
                          cg_pad=sp["cg_pad"], out=sp["buf"])
        with self.fplan.record():
            for st in steps:
                st()

    # -------------------------------------------------------------------------------------------- run
    def _refresh_weights(self) -> None:
        key = tuple(p._version for p in self.u.parameters()) if not self.training else None
        if self.training or key != self._weights_key:
            if self.gains:
                self.gain_f32.copy_(torch.stack([g.detach().float() for g in self.gains]))
            self.wplan.run()
            self._weights_key = key

    def run(self, x_in, sigma, format, embeddings, x_ref, perturbed_input, use_graph: bool) -> torch.Tensor:
        lkey = (id(format),)
        if lkey != self._lnf_key:
            self.lnf.copy_(self.u.get_ln_freqs_rows(format, self.B, self.H, self.W))
            self._lnf_key = lkey
        self._refresh_weights()
        self.x_in.copy_(x_in)
        self.x_pre.copy_(perturbed_input if perturbed_input is not None else x_in)
        self.sigma.copy_(sigma.flatten())
        self.emb_in.copy_(embeddings)
        if x_ref is not None:
            self.x_ref.copy_(x_ref)
        if use_graph:
            if not self.fplan.has_graph:
                self.fplan.run()                     # warm-up outside capture (function attributes, lazy module load)
                torch.cuda.current_stream().synchronize()
                cap = torch.cuda.Stream(device=self.dev)   # the legacy default stream cannot be captured
                self.fplan.graph_build(cap.cuda_stream)
                cap.synchronize()
            self.fplan.graph_launch()
        else:
            self.fplan.run()
        return self.out.clone()
