"""MCLT diffusion-decoder UNet on the MI355X kernels (drop-in for the reference's `DDec_MCLT_UNet_B1`).

Mirrors reference src/modules/unets/unet_edm2_ddec_mclt_b1.py:46-326 and `MPConv3D` (modules/daes/dae_edm2_d3.py:43-93):
same config dataclass, constructor `(config)`, state-dict keys (5-D conv weights) and forward signature.  The reference
keeps the stereo pair on a depth axis of 5-D tensors and convolves with (1,3,3) / (2,1,1) / (2,3,3) kernels; here the
depth axis is folded into the image batch (image n = 2*b + z, NHWC) and every layer runs on the 2-D conv kernels:
  * (1,3,3): the same 3x3 conv on both depth slices, with mirrored columns (`pad_mode = REFLECT_W`) and zero rows;
  * (2,1,1) / (2,3,3): out[z] = W[0] x[z] + W[1] x[1-z] (the reflected depth row behind a 2-deep tensor is the other channel)
    = ONE two-source conv over [x | x with the stereo pair swapped] with the weights [W[..,0] | W[..,1]].
First version of this row: eager launches; the pair-swapped copies and the materialised mp_cat are extra HBM passes
(`ddx_cat2_swap`, one read + two writes per block input) that a swapped second-source addressing mode in the conv kernel
would remove.
"""
from __future__ import annotations

import math
from dataclasses import dataclass
from typing import Optional, Union

import torch

from ... import ops
from ..._lib import weights_epoch, DDXError, RESAMPLE_DOWN, RESAMPLE_KEEP, RESAMPLE_UP, check, current_stream, dtype_code, lib, ptr
from ...engine import mp_cat_weights
from .unet import DualDiffusionUNet, DualDiffusionUNetConfig


@dataclass
class DDec_MCLT_UNet_B1_Config(DualDiffusionUNetConfig):
    in_channels: int = 1
    out_channels: int = 1
    in_channels_emb: int = 0
    in_num_freqs: int = 256
    in_psd_freqs: int = 4096
    model_channels: int = 32
    logvar_channels: int = 128
    channel_mult: tuple = (1, 2, 3, 4)
    double_midblock: bool = True
    midblock_attn: bool = False
    channel_mult_noise: Optional[int] = 4
    channel_mult_emb: Optional[int] = 4
    channels_per_head: int = 64
    num_layers_per_block: int = 3
    label_balance: float = 0.5
    concat_balance: float = 0.5
    res_balance: float = 0.3
    attn_balance: float = 0.3
    attn_levels: tuple = ()
    mlp_multiplier: int = 1
    mlp_groups: int = 1
    emb_linear_groups: int = 1
    add_constant_channel: bool = True


class MPConv3DWeight(torch.nn.Module):
    """Parameter holder of one MPConv3D (dae_edm2_d3.py:45-60): key `weight`, init randn."""

    def __init__(self, in_channels: int, out_channels: int, kernel: tuple, groups: int = 1, disable_weight_norm: bool = False):
        super().__init__()
        self.in_channels, self.out_channels, self.groups, self.kernel = in_channels, out_channels, groups, tuple(kernel)
        self.disable_weight_norm = disable_weight_norm
        self.weight = torch.nn.Parameter(torch.randn(out_channels, in_channels // groups, *kernel))


class DDecBlockWeights(torch.nn.Module):
    """Parameters of one block (unet_edm2_ddec_mclt_b1.py:75-125, attention not built)."""

    def __init__(self, in_channels: int, out_channels: int, emb_channels: int, flavor: str, resample_mode: str, cfg: DDec_MCLT_UNet_B1_Config):
        super().__init__()
        self.in_channels, self.out_channels, self.flavor, self.resample_mode = in_channels, out_channels, flavor, resample_mode
        mm, g = cfg.mlp_multiplier, cfg.mlp_groups
        self.conv_res0 = MPConv3DWeight(out_channels if flavor == "enc" else in_channels, out_channels * mm, (1, 3, 3), groups=g)
        self.conv_res1 = MPConv3DWeight(out_channels * mm, out_channels, (1, 3, 3), groups=g)
        self.conv_skip = MPConv3DWeight(in_channels, out_channels, (2, 1, 1))
        self.emb_gain = torch.nn.Parameter(torch.zeros([]))
        self.emb_linear = MPConv3DWeight(emb_channels, out_channels * mm, (1, 1, 1), groups=cfg.emb_linear_groups)


class DDec_MCLT_UNet_B1(DualDiffusionUNet):

    config_class = DDec_MCLT_UNet_B1_Config
    supports_channels_last = "3d"

    def __init__(self, config: DDec_MCLT_UNet_B1_Config) -> None:
        super().__init__()
        self.config = config
        if len(config.attn_levels) > 0 or config.midblock_attn or config.in_channels_emb > 0:
            raise NotImplementedError("DDec_MCLT_UNet_B1: attention levels / label embeddings are not built (default config has neither)")
        if config.in_channels != 1 or config.out_channels != 1:
            raise NotImplementedError("DDec_MCLT_UNet_B1: one channel per stereo slice (the default config)")
        cblock = [config.model_channels * m for m in config.channel_mult]
        cnoise = config.model_channels * config.channel_mult_noise if config.channel_mult_noise is not None else max(cblock)
        cemb = (config.model_channels * config.channel_mult_emb if config.channel_mult_emb is not None else max(cblock)) * config.mlp_multiplier
        self.num_levels, self.cnoise, self.cemb = len(cblock), cnoise, cemb
        assert config.in_psd_freqs % config.in_num_freqs == 0
        self.psd_freqs_per_freq = config.in_psd_freqs // config.in_num_freqs
        from .unet_edm2_b4 import FourierTable
        self.emb_fourier = FourierTable(cnoise)
        self.emb_noise = MPConv3DWeight(cnoise, cemb, ())
        self.logvar_fourier = FourierTable(config.logvar_channels)
        self.logvar_linear = MPConv3DWeight(config.logvar_channels, 1, (), disable_weight_norm=True)
        self.enc = torch.nn.ModuleDict()
        cout = config.in_channels + self.psd_freqs_per_freq + int(config.add_constant_channel)
        for level, ch in enumerate(cblock):
            if level == 0:
                self.enc["conv_in"] = MPConv3DWeight(cout, ch, (2, 3, 3))
                cout = ch
            else:
                self.enc[f"block{level}_down"] = DDecBlockWeights(cout, cout, cemb, "enc", "down", config)
            for i in range(config.num_layers_per_block):
                self.enc[f"block{level}_layer{i}"] = DDecBlockWeights(cout, ch, cemb, "enc", "keep", config)
                cout = ch
        skips = [m.out_channels for m in self.enc.values()]
        self.dec = torch.nn.ModuleDict()
        for level, ch in reversed(list(enumerate(cblock))):
            if level == len(cblock) - 1:
                self.dec[f"block{level}_in0"] = DDecBlockWeights(cout, cout, cemb, "dec", "keep", config)
                if config.double_midblock:
                    self.dec[f"block{level}_in1"] = DDecBlockWeights(cout, cout, cemb, "dec", "keep", config)
            else:
                self.dec[f"block{level}_up"] = DDecBlockWeights(cout, cout, cemb, "dec", "up", config)
            for i in range(config.num_layers_per_block + 1):
                self.dec[f"block{level}_layer{i}"] = DDecBlockWeights(cout + skips.pop(), ch, cemb, "dec", "keep", config)
                cout = ch
        self.out_gain = torch.nn.Parameter(torch.zeros([]))
        self.conv_out = MPConv3DWeight(cout, config.out_channels, (2, 3, 3))
        self._prepared: dict = {}
        self._prepared_key = None

    # ------------------------------------------------------------------ reference API
    def _on_placement_change(self) -> None:
        self._prepared, self._prepared_key = {}, None

    def get_embeddings(self, emb_in, conditioning_mask):
        return None     # in_channels_emb == 0 (unet_edm2_ddec_mclt_b1.py:256-262)

    @torch.no_grad()
    def get_sigma_loss_logvar(self, sigma: Optional[torch.Tensor] = None) -> torch.Tensor:
        self._require_device()
        dev = self.device
        s = sigma.flatten().to(device=dev, dtype=torch.float32).contiguous()
        f = torch.empty(s.numel(), self.config.logvar_channels, device=dev, dtype=torch.float32)
        ops.mpfourier(s, self.logvar_fourier.freqs.float(), self.logvar_fourier.phases.float(), f, True)
        out = torch.empty(s.numel(), 1, device=dev, dtype=torch.float32)
        w = self.logvar_linear.weight
        ops.linear_small(ops.make_linear_jobs([(w, None, out, 1.0, 0.0, 1, False)], dev), 1, 1, f, s.numel(), w.dtype)
        torch.cuda.current_stream().synchronize()
        return out.view(-1, 1, 1, 1)

    def get_latent_shape(self, latent_shape: Union[torch.Size, tuple]) -> torch.Size:
        q = 2 ** (self.num_levels - 1)
        return torch.Size(tuple(latent_shape[0:2]) + ((latent_shape[2] // q) * q, (latent_shape[3] // q) * q))

    def _require_device(self) -> None:
        if self.device.type != "cuda":
            raise DDXError("DDec_MCLT_UNet_B1 is not on a ROCm device: dualdiffusion_amd runs only on its HIP kernels (no CPU fallback)")

    # ------------------------------------------------------------------ weight preparation (once per weight version)
    def _prep(self) -> dict:
        key = (weights_epoch(),) + tuple(p._version for p in self.parameters()) + (self.dtype,)
        if key == self._prepared_key:
            return self._prepared
        dt, G = self.dtype, self.config.mlp_groups
        P: dict = {}

        def pair(w5: torch.Tensor, cpad: int = 0) -> torch.Tensor:
            """[Cout, Cin, 2, k, k] -> [Cout, 2*(Cin+pad), k, k]: depth tap 0 on the image itself, tap 1 on the swapped pair."""
            a, b = w5[:, :, 0], w5[:, :, 1]
            if cpad:
                z = torch.zeros(a.shape[0], cpad, *a.shape[2:], dtype=a.dtype, device=a.device)
                a, b = torch.cat([a, z], 1), torch.cat([b, z], 1)
            return torch.cat([a, b], dim=1).contiguous()

        cat = self._cat_plan()
        w_in = self.enc["conv_in"].weight.data
        cin = w_in.shape[1]
        self._cin_pad = (cin + 7) // 8 * 8
        w2 = pair(w_in, self._cin_pad - cin)
        # weight scaling is gain / sqrt(fan_in of the 5-D kernel); the zero padding columns must not count
        P["conv_in"] = ops.wprep(w2, 1, dt, gain=math.sqrt(w2[0].numel() / w_in[0].numel()))
        for side in ("enc", "dec"):
            for name, blk in getattr(self, side).items():
                if name == "conv_in":
                    continue
                pre = f"{side}.{name}"
                P[pre + ".res0"] = ops.wprep(blk.conv_res0.weight.data[:, :, 0].contiguous(), G, dt)
                P[pre + ".res1"] = ops.wprep(blk.conv_res1.weight.data[:, :, 0].contiguous(), G, dt)
                w_skip = blk.conv_skip.weight.data
                if pre in cat:      # mp_cat operand never materialised: its weights wa | wb go into the columns of both depth taps
                    Cx, Cs, wa, wb, _ = cat[pre]
                    col = torch.cat([torch.full((Cx,), wa), torch.full((Cs,), wb)]).to(device=w_skip.device, dtype=torch.float32)
                    w_skip = w_skip.float() * col[None, :, None, None, None]     # (fp32 master for the preparation: no extra rounding)
                P[pre + ".skip"] = ops.wprep(pair(w_skip), 1, dt)
        w_out = self.conv_out.weight.data
        w8 = torch.zeros(8, *w_out.shape[1:], dtype=w_out.dtype, device=w_out.device)   # 1 output channel -> one 16-byte NHWC vector
        w8[:w_out.shape[0]] = w_out
        self._out_gain = self.out_gain.data.float().reshape(1)
        P["conv_out"] = ops.wprep(pair(w8), 1, dt, gain_ptr=self._out_gain)
        # emb_linear of every block in ONE launch: gains as device scalars, job tables cached per image count
        self._blocks = [(f"{side}.{n}", b) for side in ("enc", "dec") for n, b in getattr(self, side).items() if n != "conv_in"]
        self._gain32 = {pre: b.emb_gain.data.float().reshape(1) for pre, b in self._blocks}
        self._emb_tables: dict = {}
        self._prepared, self._prepared_key = P, key
        return P

    def _cat_plan(self) -> dict:
        """{decoder block with a skip input: (Cx, Cskip, wa, wb, index of the skip in encoder order)} (forward :317-322)."""
        skips = [m.out_channels for m in self.enc.values()]
        plan, cout = {}, skips[-1]
        for name, blk in self.dec.items():
            if "layer" in name:
                cs = skips.pop()
                wa, wb = mp_cat_weights(cout, cs, self.config.concat_balance)
                plan["dec." + name] = (cout, cs, float(wa), float(wb), len(skips))
            cout = blk.out_channels
        return plan

    def _emb_scales(self, emb2: torch.Tensor) -> dict:
        """c = emb_linear(emb) * emb_gain + 1 of every block (unet_edm2_ddec_mclt_b1.py:104-105), {block: [N, Cmid] fp32}."""
        N = emb2.shape[0]
        if N not in self._emb_tables:
            outs = {pre: torch.empty(N, b.conv_res0.out_channels, dtype=torch.float32, device=emb2.device) for pre, b in self._blocks}
            table = ops.make_linear_jobs([(b.emb_linear.weight, self._gain32[pre], outs[pre], 1.0, 1.0, self.config.emb_linear_groups, False)
                                          for pre, b in self._blocks], emb2.device)
            self._emb_tables[N] = (table, outs, max(b.conv_res0.out_channels for _, b in self._blocks))
        table, outs, max_o = self._emb_tables[N]
        ops.linear_small(table, len(self._blocks), max_o, emb2, N, self._blocks[0][1].emb_linear.weight.dtype)
        return outs

    # ------------------------------------------------------------------ forward
    def _block(self, P: dict, pre: str, blk: DDecBlockWeights, x: torch.Tensor, x_act: Optional[torch.Tensor], c: torch.Tensor,
               skip: Optional[torch.Tensor] = None, skip_act: Optional[torch.Tensor] = None, twin_scale: Optional[float] = None):
        """x (| skip: mp_cat operand, never materialised) -> (block output, mp_silu(twin_scale * output) | None).

        Every operand is consumed as stored: activations are applied by the PRODUCER (pixel-norm and the previous convs write
        the mp_silu'd twins the 3x3 convs read, with the consumer's mp_cat weight inside the activation), so all convs run on
        the LDS-DMA kernel.  The (2,1,1) skip conv mixes the stereo pair through DDX_PAD_SWAP_SRC1 / _PAIRED (second depth tap =
        the same tensors, image b ^ 1; the mp_cat weights live in its prepared weights) and a nearest upsample is folded into
        the source addressing of both convs -- no swapped copies, no concatenated or upsampled tensor.
        x_act: mp_silu(x) (blocks without skip) / mp_silu(wa * x) (with skip); skip_act: mp_silu(wb * skip)."""
        cfg = self.config
        rs = RESAMPLE_UP if blk.resample_mode == "up" else RESAMPLE_KEEP
        if blk.resample_mode == "down":
            N, H, W, Cn = x.shape
            x = ops.resample2d(x, torch.empty((N, H // 2, W // 2, Cn), dtype=x.dtype, device=x.device), RESAMPLE_DOWN)
        if blk.flavor == "enc":
            Cn, Co = x.shape[-1], blk.out_channels
            if x.dtype == torch.bfloat16 and Co <= 64 and Co % 8 == 0 and Cn % 32 == 0:
                # pixel norm in the skip conv's epilogue (fp32 accumulators, all channels of a pixel in one wave), twin alongside
                x_act = torch.empty(*x.shape[:-1], Co, dtype=x.dtype, device=x.device)
                x = ops.conv2d(x, P[pre + ".skip"], src1=x, swap_src1=True, pixelnorm_eps=1e-4, out2=x_act)
            else:
                xs = ops.conv2d(x, P[pre + ".skip"], src1=x, swap_src1=True)
                x_act = torch.empty_like(xs)
                x = ops.pixelnorm(xs, out_act=x_act)
            y = ops.conv2d(x_act, P[pre + ".res0"], reflect_w=True, out_act=True, out_scale=c)
        else:
            if x_act is None or (skip is not None and skip_act is None):
                raise DDXError("DDec block: decoder blocks read pre-activated operands")
            y = ops.conv2d(x_act, P[pre + ".res0"], src1=skip_act, resample=rs, reflect_w=True, out_act=True, out_scale=c)
            if skip is not None:
                x = ops.conv2d(x, P[pre + ".skip"], src1=skip, swap_paired=True)
            else:
                x = ops.conv2d(x, P[pre + ".skip"], src1=x, swap_src1=True, resample=rs)
        twin = torch.empty_like(x) if twin_scale is not None else None
        out = ops.conv2d(y, P[pre + ".res1"], residual=x, res_t=cfg.res_balance, clip=256.0, reflect_w=True, out2=twin,
                         out2_scale=twin_scale if twin_scale is not None else 1.0)
        return out, twin

    @torch.no_grad()
    def forward(self, x_in: torch.Tensor, sigma: torch.Tensor, format, embeddings: Optional[torch.Tensor] = None,
                x_ref: Optional[torch.Tensor] = None, perturbed_input: Optional[torch.Tensor] = None) -> torch.Tensor:
        """reference unet_edm2_ddec_mclt_b1.py:275-326.  x_in [B, 2, H, W], x_ref [B, 2, in_psd_freqs, W] -> float32 [B, 2, H, W]."""
        self._require_device()
        cfg, dev, dt = self.config, self.device, self.dtype
        if x_ref is None:
            raise DDXError("DDec_MCLT_UNet_B1.forward needs x_ref (the un-mel'd PSD conditioning)")
        B, _, H, W = x_in.shape
        if H != cfg.in_num_freqs:
            raise DDXError(f"x_in has {H} frequency rows, the config says in_num_freqs = {cfg.in_num_freqs}")
        P = self._prep()
        x_in = x_in.to(dev, torch.float32).contiguous()
        sig = sigma.flatten().to(dev, torch.float32).contiguous()
        src = perturbed_input.to(dev, torch.float32).contiguous() if perturbed_input is not None else x_in
        xr = x_ref.to(dev, torch.float32).contiguous()
        ppf = self.psd_freqs_per_freq
        if tuple(xr.shape) != (B, 2, H * ppf, W):
            raise DDXError(f"x_ref must be [B, 2, {H * ppf}, W], got {tuple(xr.shape)}")
        # ---- input assembly: channels [c_in * x, psd chunk 0..ppf-1, 1] of image n = 2b + z (zero padded to 8k)
        x0 = torch.empty(B * 2, H, W, self._cin_pad, dtype=dt, device=dev)
        check(lib().ddx_ddec_input_prep(ptr(src), ptr(xr), ptr(sig), ptr(x0), None, B, H, W, ppf, self._cin_pad, cfg.sigma_data,
                                        int(cfg.add_constant_channel), dtype_code(dt), current_stream()), "ddec_input_prep")
        # ---- embedding: emb_noise(fourier(ln sigma / 4)) (no label path), one row per image
        four = torch.empty(B, self.cnoise, dtype=torch.float32, device=dev)
        ops.mpfourier(sig, self.emb_fourier.freqs.float().contiguous(), self.emb_fourier.phases.float().contiguous(), four, True)
        emb = torch.empty(B, self.cemb, dtype=torch.float32, device=dev)
        w_n = self.emb_noise.weight
        t_n = ops.make_linear_jobs([(w_n, None, emb, 1.0, 0.0, 1, False)], dev)
        ops.linear_small(t_n, 1, self.cemb, four, B, w_n.dtype)
        if dt == torch.bfloat16:
            emb = emb.to(dt).float()            # the reference casts emb to bfloat16 before the emb_linear layers (:305)
        emb2 = emb.repeat_interleave(2, dim=0).contiguous()
        cs = self._emb_scales(emb2)
        # ---- encoder / decoder.  Every tensor is written once raw and once as the activated twin its 3x3 consumer reads:
        # encoder outputs as mp_silu(wb * skip) for the decoder block that concatenates them, decoder outputs as mp_silu(x) or
        # mp_silu(wa * x) for the next block.
        cat = self._cat_plan()
        skip_scale = {idx: wb for (_, _, _, wb, idx) in cat.values()}
        enc_names = [n for n in self.enc if n != "conv_in"]
        dec_names = list(self.dec)
        tw0 = torch.empty(B * 2, H, W, self.enc["conv_in"].out_channels, dtype=dt, device=dev)
        x = ops.conv2d(x0, P["conv_in"], src1=x0, swap_src1=True, reflect_w=True, out2=tw0, out2_scale=skip_scale[0])
        skips, skip_acts = [x], [tw0]
        coll = getattr(self, "collect", None)       # tests: {} -> block outputs [2B, H, W, C] (image n = 2b + z)
        if coll is not None:
            coll["enc.conv_in"] = x
        for name in enc_names:
            x, tw = self._block(P, "enc." + name, self.enc[name], x, None, cs["enc." + name], twin_scale=skip_scale[len(skips)])
            skips.append(x)
            skip_acts.append(tw)
            if coll is not None:
                coll["enc." + name] = x
        # the last encoder output is also the first decoder block's input: that block reads the unit-scale twin
        x_act = ops.silu_scale_fwd(x, None, 1.0)
        for i, name in enumerate(dec_names):
            pre = "dec." + name
            nxt = "dec." + dec_names[i + 1] if i + 1 < len(dec_names) else None
            tws = None if nxt is None else (cat[nxt][2] if nxt in cat else 1.0)
            if pre in cat:
                x, x_act = self._block(P, pre, self.dec[name], x, x_act, cs[pre], skips.pop(), skip_acts.pop(), twin_scale=tws)
            else:
                x, x_act = self._block(P, pre, self.dec[name], x, x_act, cs[pre], twin_scale=tws)
            if coll is not None:
                coll[pre] = x
        y8 = ops.conv2d(x, P["conv_out"], src1=x, swap_src1=True, reflect_w=True)
        out = torch.empty(B, 2, H, W, dtype=torch.float32, device=dev)
        check(lib().ddx_ddec_output_combine(ptr(y8), y8.shape[-1], ptr(x_in), ptr(sig), ptr(out), B, 2 * H * W, cfg.sigma_data, dtype_code(dt),
                                            current_stream()), "ddec_output_combine")
        torch.cuda.current_stream().synchronize()    # the emb_noise job table of this call is a temporary
        return out
