"""DAE_G1 autoencoder (the live stage between the UNet and the diffusion decoder) on the MI355X kernels.

Drop-in for reference src/modules/daes/dae_edm2_g1.py: same `DAE_G1_Config` fields, constructor `(config)`, `state_dict()` keys
(5-D `MPConv3D_E` weights) and methods (`get_embeddings`, `encode`, `decode`, `tiled_encode`, `get_latent_shape`,
`get_mel_spec_shape`, `normalize_weights`).  As in the diffusion decoder (unets/unet_edm2_ddec_mclt_b1.py) the stereo depth axis
of the reference's 5-D tensors is folded into the image batch (image n = 2 b + z, NHWC) and every layer runs on the 2-D conv kernels:
  * (1,3,3) encoder kernels: 3x3 convs with mirrored columns (`REFLECT_W`) and zero rows;
  * (2,3,3) decoder kernels: out[z] = W[0] x[z] + W[1] x[1 - z] -- ONE two-source conv over [x | pair-swapped x] (`SWAP_SRC1`) with the
    weights [W[..,0] | W[..,1]], the nearest upsample of `up` blocks folded into the source addressing;
  * (1,5,5) input / output convs (2 -> C and C -> 1 channels at the full mel resolution): the scalar kernel;
  * attention (levels in `attn_levels`): tokens along h for every (b, z, w) column -- the axis-folded attention kernel
    (ddx_attn_fold_fwd) on the maps as they lie, no transposes (reference :209-228).
Eager launches (no launch plan yet).  Activations are applied by the producer where the consumer is a conv (mp_silu twins).
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Union

import torch

from ... import ops
from ..._lib import DDXError, RESAMPLE_DOWN, RESAMPLE_KEEP, RESAMPLE_UP, bump_weights_epoch, weights_epoch
from .dae import DualDiffusionDAE, DualDiffusionDAEConfig


@dataclass
class DAE_G1_Config(DualDiffusionDAEConfig):
    in_channels: int = 1
    out_channels: int = 1
    in_channels_emb: int = 1024
    in_num_freqs: int = 256
    latent_channels: int = 4
    model_channels: int = 32
    channel_mult_enc: int = 1
    channel_mult_dec: list = (1, 2, 4, 8)
    channel_mult_emb: int = 4
    num_attn_heads: int = 8
    num_enc_layers: int = 6
    num_dec_layers_per_block: int = 3
    res_balance: float = 0.3
    attn_balance: float = 0.3
    attn_levels: list = ()
    mlp_multiplier: int = 2
    mlp_groups: int = 1
    emb_linear_groups: int = 1
    add_constant_channel: bool = True
    add_pixel_norm: bool = False


class MPConv3D_E(torch.nn.Module):
    """Parameter holder of one MPConv3D_E (reference :68-93): key `weight`, init randn; `normalize_weights` over dim 1 (:123-126)."""

    def __init__(self, in_channels: int, out_channels: int, kernel: tuple, groups: int = 1, disable_weight_norm: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.groups, self.kernel = in_channels, out_channels, groups, tuple(kernel)
        self.disable_weight_norm = disable_weight_norm
        self.weight = torch.nn.Parameter(torch.randn(out_channels, in_channels // groups, *kernel))

    @torch.no_grad()
    def normalize_weights(self) -> None:
        if self.disable_weight_norm:
            return
        if not self.weight.is_cuda:
            raise DDXError("normalize_weights needs the module on a ROCm device (no CPU path)")
        w = self.weight.data
        # normalize(w, dim=1): one norm per (output channel, kernel position) over the input channels
        rows = w.movedim(1, -1).contiguous()
        ops.normalize_weights_(rows.view(-1, w.shape[1]))
        w.copy_(rows.movedim(-1, 1))
        bump_weights_epoch()


class G1Block(torch.nn.Module):
    """Parameters of one block (reference :128-180)."""

    def __init__(self, level: int, in_channels: int, out_channels: int, emb_channels: int, flavor: str, resample_mode: str,
                 use_attention: bool, cfg: DAE_G1_Config):
        super().__init__()
        self.level, self.in_channels, self.out_channels, self.flavor, self.resample_mode = level, in_channels, out_channels, flavor, resample_mode
        self.use_attention = use_attention
        mm, g = cfg.mlp_multiplier, cfg.mlp_groups
        kernel = (1, 3, 3) if flavor == "enc" else (2, 3, 3)
        self.conv_res0 = MPConv3D_E(out_channels if flavor == "enc" else in_channels, out_channels * mm, kernel, groups=g)
        self.conv_res1 = MPConv3D_E(out_channels * mm, out_channels, kernel, groups=g)
        self.conv_skip = MPConv3D_E(in_channels, out_channels, (1, 1, 1)) if (in_channels != out_channels or g > 1) else None
        self.emb_gain = torch.nn.Parameter(torch.zeros([]))
        self.emb_linear = MPConv3D_E(emb_channels, out_channels * mm, (1, 1, 1), groups=cfg.emb_linear_groups) if emb_channels != 0 else None
        if use_attention:
            self.attn_qkv = MPConv3D_E(out_channels, out_channels * 3, (1, 1, 1))
            self.attn_proj = MPConv3D_E(out_channels, out_channels, (1, 1, 1))


class DAE_G1(DualDiffusionDAE):

    config_class = DAE_G1_Config
    supports_channels_last = "3d"

    def __init__(self, config: DAE_G1_Config) -> None:
        super().__init__()
        self.config = c = config
        if c.in_channels != 1 or c.out_channels != 1:
            raise NotImplementedError("DAE_G1: one channel per stereo slice (the default config)")
        if c.add_pixel_norm:
            raise NotImplementedError("DAE_G1: add_pixel_norm is not built (default False)")
        cemb = c.model_channels * c.channel_mult_emb * c.mlp_multiplier if c.in_channels_emb > 0 else 0
        self.num_levels = len(c.channel_mult_dec)
        self.downsample_ratio = 2 ** (self.num_levels - 1)
        self.out_gain = torch.nn.Parameter(torch.ones([]))
        self.recon_loss_logvar = torch.nn.Parameter(torch.zeros([]))
        self.emb_label = MPConv3D_E(c.in_channels_emb, cemb, ()) if c.in_channels_emb > 0 else None
        self.emb_dim = cemb
        in_channels = 1 + int(c.add_constant_channel)
        enc_ch = c.model_channels * c.channel_mult_enc
        dec_ch = [c.model_channels * m for m in c.channel_mult_dec]
        self.enc = torch.nn.ModuleDict()
        self.enc["conv_in"] = MPConv3D_E(in_channels, enc_ch, (1, 5, 5))
        for i in range(c.num_enc_layers):
            self.enc[f"block0_layer{i}"] = G1Block(0, enc_ch, enc_ch, 0, "enc", "keep", False, c)
        self.conv_latents_out = MPConv3D_E(enc_ch, c.latent_channels, (1, 3, 3))
        self.conv_latents_in = MPConv3D_E(c.latent_channels + int(c.add_constant_channel), dec_ch[-1], (2, 3, 3))
        self.dec = torch.nn.ModuleDict()
        cin = dec_ch[-1]
        for level in reversed(range(self.num_levels)):
            cout = dec_ch[level]
            attn = level in c.attn_levels
            if level == self.num_levels - 1:
                self.dec[f"block{level}_in0"] = G1Block(level, cin, cout, cemb, "dec", "keep", attn, c)
            else:
                self.dec[f"block{level}_up"] = G1Block(level, cin, cout, cemb, "dec", "up", attn, c)
            for i in range(c.num_dec_layers_per_block):
                self.dec[f"block{level}_layer{i}"] = G1Block(level, cout, cout, cemb, "dec", "keep", attn, c)
            cin = cout
        self.conv_out = MPConv3D_E(cin, c.out_channels, (1, 5, 5))
        self._prepared, self._prepared_key = {}, None

    # ------------------------------------------------------------------ reference API
    def _on_placement_change(self) -> None:
        self._prepared, self._prepared_key = {}, None

    def _require_device(self) -> None:
        if self.device.type != "cuda":
            raise DDXError("DAE_G1 is not on a ROCm device: dualdiffusion_amd runs only on its HIP kernels (no CPU fallback)")

    @torch.no_grad()
    def get_embeddings(self, emb_in: torch.Tensor) -> Optional[torch.Tensor]:
        """reference :305-309: emb_label(normalize(emb_in))."""
        if self.emb_label is None:
            return None
        self._require_device()
        dev = self.device
        xn = ops.pixelnorm(emb_in.to(device=dev, dtype=torch.float32).contiguous())
        out = torch.empty(emb_in.shape[0], self.emb_dim, device=dev, dtype=torch.float32)
        w = self.emb_label.weight
        table = ops.make_linear_jobs([(w, None, out, 1.0, 0.0, 1, False)], dev)
        ops.linear_small(table, 1, self.emb_dim, xn, emb_in.shape[0], w.dtype)
        torch.cuda.current_stream().synchronize()      # (the job table is a temporary)
        return out.to(self.dtype)

    def get_recon_loss_logvar(self) -> torch.Tensor:
        return self.recon_loss_logvar

    def get_latent_shape(self, mel_spec_shape: Union[torch.Size, tuple]) -> tuple:
        if len(mel_spec_shape) != 4:
            raise ValueError(f"Invalid sample shape: {mel_spec_shape}")
        q = 2 ** (self.num_levels - 1)
        return (mel_spec_shape[0], self.config.latent_channels * 2, mel_spec_shape[2] // q, mel_spec_shape[3] // q)

    def get_mel_spec_shape(self, latent_shape: Union[torch.Size, tuple]) -> tuple:
        if len(latent_shape) != 4:
            raise ValueError(f"Invalid latent shape: {latent_shape}")
        q = 2 ** (self.num_levels - 1)
        return (latent_shape[0], 2, latent_shape[2] * q, latent_shape[3] * q)

    # ------------------------------------------------------------------ weight preparation (once per weight version)
    def _prep(self) -> dict:
        key = (weights_epoch(),) + tuple(p._version for p in self.parameters()) + (self.dtype,)
        if key == self._prepared_key:
            return self._prepared
        dt, G, dev = self.dtype, self.config.mlp_groups, self.device
        P: dict = {}

        def pad_in(w4: torch.Tensor, mult: int = 8) -> torch.Tensor:
            cin = w4.shape[1]
            cp = (cin + mult - 1) // mult * mult
            if cp == cin:
                return w4.contiguous()
            z = torch.zeros(w4.shape[0], cp - cin, *w4.shape[2:], dtype=w4.dtype, device=w4.device)
            return torch.cat([w4, z], 1).contiguous()

        def pair(w5: torch.Tensor, mult: int = 8) -> torch.Tensor:
            """[Cout, Cin, 2, k, k] -> [Cout, 2 * Cin_padded, k, k]: depth tap 0 on the image itself, tap 1 on the pair-swapped image."""
            return torch.cat([pad_in(w5[:, :, 0], mult), pad_in(w5[:, :, 1], mult)], dim=1).contiguous()

        def prep(w4: torch.Tensor, groups: int, true_fan: int, **kw):
            # weight scaling is gain / sqrt(fan_in of the 5-D kernel): zero padding columns / rows must not count
            return ops.wprep(w4, groups, dt, gain=math.sqrt(w4[0].numel() / true_fan), **kw)

        def pad_out(w4: torch.Tensor, rows: int) -> torch.Tensor:
            if w4.shape[0] >= rows:
                return w4
            z = torch.zeros(rows - w4.shape[0], *w4.shape[1:], dtype=w4.dtype, device=w4.device)
            return torch.cat([w4, z], 0).contiguous()

        w = self.enc["conv_in"].weight.data
        self._cin_pad = (w.shape[1] + 7) // 8 * 8
        P["enc.conv_in"] = prep(pad_in(w[:, :, 0]), 1, w[0].numel())
        for name, blk in self.enc.items():
            if name == "conv_in":
                continue
            for cn in ("conv_res0", "conv_res1"):
                w = getattr(blk, cn).weight.data
                P[f"enc.{name}.{cn}"] = prep(w[:, :, 0].contiguous(), G, w[0].numel())
            if blk.conv_skip is not None:
                w = blk.conv_skip.weight.data
                P[f"enc.{name}.conv_skip"] = prep(w[:, :, 0].contiguous(), 1, w[0].numel())
        w = self.conv_latents_out.weight.data
        P["conv_latents_out"] = prep(pad_out(w[:, :, 0].contiguous(), 8), 1, w[0].numel())      # 4 output channels -> one 16-byte NHWC vector
        w = self.conv_latents_in.weight.data
        self._lat_pad = (w.shape[1] + 7) // 8 * 8
        P["conv_latents_in"] = prep(pair(w), 1, w[0].numel())
        self._blocks = []
        for name, blk in self.dec.items():
            pre = f"dec.{name}"
            self._blocks.append((pre, blk))
            if G != 1:
                raise NotImplementedError("DAE_G1: grouped (2,3,3) decoder kernels are not built (mlp_groups = 1 in every shipped config)")
            for cn in ("conv_res0", "conv_res1"):
                w = getattr(blk, cn).weight.data
                P[f"{pre}.{cn}"] = prep(pair(w), 1, w[0].numel())
            if blk.conv_skip is not None:
                w = blk.conv_skip.weight.data
                P[f"{pre}.conv_skip"] = prep(w[:, :, 0].contiguous(), 1, w[0].numel())
            if blk.use_attention:
                C_, heads = blk.out_channels, self.config.num_attn_heads
                d = C_ // heads
                w = blk.attn_qkv.weight.data[:, :, 0]                       # rows: head * 3d + dd * 3 + s   (reference :214-216)
                idx = torch.arange(3 * C_, device=w.device).view(heads, d, 3)
                qk_rows = idx[:, :, :2].permute(0, 2, 1).reshape(-1)        # -> (head, {q, k}, dd): what the attention kernel reads
                v_rows = idx[:, :, 2].reshape(-1)                           # -> (head, dd)
                wq = w[torch.cat([qk_rows, v_rows])].contiguous()
                P[f"{pre}.attn_qkv"] = prep(wq, 1, blk.attn_qkv.weight.data[0].numel())
                w = blk.attn_proj.weight.data
                P[f"{pre}.attn_proj"] = prep(w[:, :, 0].contiguous(), 1, w[0].numel())
        w = self.conv_out.weight.data
        self._out_gain = self.out_gain.data.float().reshape(1)
        P["conv_out"] = ops.wprep(pad_out(w[:, :, 0].contiguous(), 8), 1, dt, gain_ptr=self._out_gain)
        self._gain32 = {pre: b.emb_gain.data.float().reshape(1) for pre, b in self._blocks if b.emb_linear is not None}
        self._emb_tables: dict = {}
        self._ones: dict = {}
        self._prepared, self._prepared_key = P, key
        return P

    def _emb_scales(self, emb2: torch.Tensor) -> dict:
        """c = emb_linear(emb) * emb_gain + 1 of every decoder block (:190-192), {block: [N, Cmid] fp32}, one launch."""
        N = emb2.shape[0]
        blocks = [(pre, b) for pre, b in self._blocks if b.emb_linear is not None]
        if N not in self._emb_tables:
            outs = {pre: torch.empty(N, b.conv_res0.out_channels, dtype=torch.float32, device=emb2.device) for pre, b in blocks}
            table = ops.make_linear_jobs([(b.emb_linear.weight, self._gain32[pre], outs[pre], 1.0, 1.0, self.config.emb_linear_groups, False)
                                          for pre, b in blocks], emb2.device)
            self._emb_tables[N] = (table, outs, max(b.conv_res0.out_channels for _, b in blocks))
        table, outs, max_o = self._emb_tables[N]
        ops.linear_small(table, len(blocks), max_o, emb2, N, blocks[0][1].emb_linear.weight.dtype)
        return outs

    # ------------------------------------------------------------------ encode
    @torch.no_grad()
    def encode(self, x: torch.Tensor, embeddings: Optional[torch.Tensor] = None, normalize_latents: bool = True) -> torch.Tensor:
        """reference :331-349.  x [B, 2, H, W] mel spectrogram -> latents [B, 2 * latent_channels, H / ds, W / ds] (fp32)."""
        self._require_device()
        c, dt = self.config, self.dtype
        P = self._prep()
        x = x.to(self.device, torch.float32).contiguous()
        B, _, H, W = x.shape
        if H % self.downsample_ratio or W % self.downsample_ratio:
            raise DDXError(f"sample size {H}x{W} must be a multiple of the downsample ratio {self.downsample_ratio}")
        img = ops.stereo_to_images(x, 1, self._cin_pad, c.add_constant_channel, dt)
        h = ops.conv2d(img, P["enc.conv_in"], reflect_w=True, path="direct")
        act = ops.silu_scale_fwd(h, None, 1.0)
        for name, blk in self.enc.items():
            if name == "conv_in":
                continue
            pre = f"enc.{name}"
            if blk.conv_skip is not None:
                h = ops.conv2d(h, P[pre + ".conv_skip"])
                act = ops.silu_scale_fwd(h, None, 1.0)
            y = ops.conv2d(act, P[pre + ".conv_res0"], reflect_w=True, out_act=True)       # mp_silu(conv) (no embedding in the encoder, :193-194)
            act = torch.empty_like(h)
            h = ops.conv2d(y, P[pre + ".conv_res1"], residual=h, res_t=c.res_balance, clip=256.0, reflect_w=True, out2=act, out2_scale=1.0)
            if getattr(self, "collect", None) is not None:
                self.collect[pre] = h
        z8 = ops.conv2d(h, P["conv_latents_out"], reflect_w=True)                          # [2B, H, W, 8] (4 latent channels used)
        # avg_pool2d(ds) as repeated 2x2 means on the NHWC images, then tensor_5d_to_4d: latent channel c of slice z -> channel c * 2 + z
        ds = self.downsample_ratio
        while ds > 1:
            z8 = ops.resample2d(z8, torch.empty(z8.shape[0], z8.shape[1] // 2, z8.shape[2] // 2, z8.shape[3], dtype=z8.dtype, device=z8.device),
                                RESAMPLE_DOWN)
            ds //= 2
        lat = ops.images_to_stereo(z8, c.latent_channels)                                   # [B, 8, H / ds, W / ds] fp32
        if normalize_latents:
            lat = self._normalize_latents(lat)
        return lat

    def _normalize_latents(self, lat: torch.Tensor) -> torch.Tensor:
        B = lat.shape[0]
        flat = lat.reshape(B, -1).contiguous()
        return ops.pixelnorm(flat).view(lat.shape)

    # ------------------------------------------------------------------ decode
    def _dec_block(self, P: dict, pre: str, blk: G1Block, x: torch.Tensor, x_act: torch.Tensor, cvec: Optional[torch.Tensor]):
        c = self.config
        rs = RESAMPLE_UP if blk.resample_mode == "up" else RESAMPLE_KEEP
        y = ops.conv2d(x_act, P[pre + ".conv_res0"], src1=x_act, swap_src1=True, resample=rs, reflect_w=True, out_act=True, out_scale=cvec)
        if blk.conv_skip is not None:
            res = ops.conv2d(x, P[pre + ".conv_skip"], resample=rs)
        elif rs == RESAMPLE_UP:
            N, H, W, Cn = x.shape
            res = ops.resample2d(x, torch.empty(N, 2 * H, 2 * W, Cn, dtype=x.dtype, device=x.device), RESAMPLE_UP)
        else:
            res = x
        attn = blk.use_attention
        out = ops.conv2d(y, P[pre + ".conv_res1"], src1=y, swap_src1=True, reflect_w=True, residual=res, res_t=c.res_balance,
                         clip=0.0 if attn else 256.0)
        if attn:
            Cn, heads = blk.out_channels, c.num_attn_heads
            qkv = ops.conv2d(out, P[pre + ".attn_qkv"])
            N = qkv.shape[0]
            ones = self._ones.get((N, Cn))
            if ones is None:
                ones = self._ones[(N, Cn)] = torch.ones(N, Cn, dtype=torch.float32, device=self.device)
            ao = ops.attention_fold(qkv[..., :2 * Cn], qkv[..., 2 * Cn:], heads, out_scale=ones)        # mp_silu(attention)
            out = ops.conv2d(ao, P[pre + ".attn_proj"], residual=out, res_t=c.attn_balance, clip=256.0)
        return out, ops.silu_scale_fwd(out, None, 1.0)

    @torch.no_grad()
    def decode(self, x: torch.Tensor, embeddings: Optional[torch.Tensor]) -> torch.Tensor:
        """reference :351-364.  latents [B, 2 * latent_channels, h, w] -> mel spectrogram [B, 2, h * ds, w * ds] (fp32)."""
        self._require_device()
        c = self.config
        P = self._prep()
        x = x.to(self.device, torch.float32).contiguous()
        B = x.shape[0]
        img = ops.stereo_to_images(x, c.latent_channels, self._lat_pad, c.add_constant_channel, self.dtype)
        h = ops.conv2d(img, P["conv_latents_in"], src1=img, swap_src1=True, reflect_w=True)
        act = ops.silu_scale_fwd(h, None, 1.0)
        cs = {}
        if embeddings is not None and self.emb_label is not None:
            emb2 = embeddings.to(self.device, torch.float32).repeat_interleave(2, dim=0).contiguous()
            cs = self._emb_scales(emb2)
        coll = getattr(self, "collect", None)       # tests: {} -> block outputs [2B, H, W, C] (image n = 2b + z)
        for pre, blk in self._blocks:
            h, act = self._dec_block(P, pre, blk, h, act, cs.get(pre))
            if coll is not None:
                coll[pre] = h
        y8 = ops.conv2d(h, P["conv_out"], reflect_w=True, path="direct")                     # [2B, H, W, 8] (1 channel used)
        return ops.images_to_stereo(y8, 1)                                                    # [B, 2, H, W] fp32

    def forward(self, samples: torch.Tensor, dae_embeddings: torch.Tensor, add_latents_noise: float = 0):
        """reference :366-373."""
        pre = self.encode(samples, dae_embeddings, normalize_latents=False)
        latents = self._normalize_latents(pre)
        if add_latents_noise > 0:
            latents = self._normalize_latents(latents + torch.randn_like(latents) * add_latents_noise)
        return latents, self.decode(latents, dae_embeddings), pre

    @torch.no_grad()
    def tiled_encode(self, x: torch.Tensor, embeddings: Optional[torch.Tensor], max_chunk: int = 6144, overlap: int = 256) -> torch.Tensor:
        """reference :375-427: encode in overlapping chunks along the time axis (activation memory), keep each chunk's interior."""
        x_w, ds = x.shape[-1], self.downsample_ratio
        assert max_chunk % ds == 0, "max_chunk must be divisible by downsample ratio"
        assert overlap % ds == 0, "overlap must be divisible by downsample ratio"
        assert x_w % ds == 0, "sample length must be divisible by downsample ratio"
        if x_w <= max_chunk:
            return self.encode(x, embeddings)
        min_chunk_len, out_overlap = overlap * 3, overlap // ds
        latents = torch.zeros(x.shape[0], self.config.latent_channels * 2, x.shape[-2] // ds, x_w // ds, device=self.device, dtype=torch.float32)
        for w_start in range(0, x_w, max_chunk - overlap * 2):
            chunk_start, chunk_end = max(0, w_start), min(x_w, w_start + max_chunk)
            if chunk_end - chunk_start < min_chunk_len:
                chunk_start -= min_chunk_len - (chunk_end - chunk_start)
            chunk = self.encode(x[:, :, :, chunk_start:chunk_end], embeddings, normalize_latents=False)
            first, last = w_start == 0, chunk_end == x_w
            valid_start = 0 if first else out_overlap
            valid_end = chunk.shape[3] if last else chunk.shape[3] - out_overlap
            dest_start = chunk_start // ds if first else chunk_start // ds + out_overlap
            dest_end = chunk_end // ds if last else chunk_end // ds - out_overlap
            latents[:, :, :, dest_start:dest_end] = chunk[:, :, :, valid_start:valid_end]
        return self._normalize_latents(latents)
