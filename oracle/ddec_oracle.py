"""CPU restatement of the reference's MCLT diffusion-decoder UNet (TEST INFRASTRUCTURE ONLY).

Follows /root/reference/src/modules/unets/unet_edm2_ddec_mclt_b1.py:46-326 (`DDec_MCLT_UNet_B1`, eval-mode forward) and
`MPConv3D` /root/reference/src/modules/daes/dae_edm2_d3.py:43-93: 5-D activations (B, C, 2, H, W) with the stereo pair on the
depth axis, kernels (1,3,3) / (2,1,1) / (2,3,3) with reflection padding on W, zero padding on H and a reflected depth row
behind (for depth 2: the other channel).  The reference hard-codes bfloat16 for the body (:295-305); `compute_dtype`
reproduces that (default) or keeps fp32 for the fp32 HIP parity path.  Pinned against the reference by tools/make_golden.py
(tests/golden/ddec_small.safetensors); only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline may import this.
"""
import math
from typing import Optional

import torch
import torch.nn.functional as F

from .edm2_oracle import cat_mp, fourier_mp, fourier_tables, rms_normalize

DEFAULT_DDEC_CFG = dict(
    in_channels=1, out_channels=1, in_channels_emb=0, in_num_freqs=256, in_psd_freqs=4096, model_channels=32, logvar_channels=128,
    channel_mult=(1, 2, 3, 4), double_midblock=True, midblock_attn=False, channel_mult_noise=4, channel_mult_emb=4, channels_per_head=64,
    num_layers_per_block=3, label_balance=0.5, concat_balance=0.5, res_balance=0.3, attn_balance=0.3, attn_levels=(), mlp_multiplier=1,
    mlp_groups=1, emb_linear_groups=1, add_constant_channel=True, sigma_data=1.0, dropout=0.0)


def silu_mp(x: torch.Tensor) -> torch.Tensor:
    """mp_tools.py:268-269.  Written with the primitive the reference uses: in bfloat16 (this model's body) `F.silu` and
    `x * sigmoid(x)` round differently, and the fixture is matched bit for bit."""
    return F.silu(x) / 0.596


def sum_mp(a: torch.Tensor, b: torch.Tensor, t: float) -> torch.Tensor:
    """mp_tools.py:274-279 (float t), through `lerp` for the same reason."""
    return a.lerp(b, t) / ((1 - t) ** 2 + t ** 2) ** 0.5


def ddec_cfg(**over) -> dict:
    cfg = dict(DEFAULT_DDEC_CFG)
    cfg.update(over)
    return cfg


def conv3d_mp(x: torch.Tensor, w: torch.Tensor, gain=1.0, groups: int = 1) -> torch.Tensor:
    """MPConv3D.forward in eval mode (dae_edm2_d3.py:68-84)."""
    w = w.float()
    w = w * (gain / math.sqrt(w[0].numel()))
    w = w.to(x.dtype)
    if w.ndim == 2:
        return x @ w.t()
    kz, kw = w.shape[2], w.shape[4]
    if kz // 2 or kw // 2:
        x = F.pad(x, (kw // 2, kw // 2, 0, 0, 0, kz // 2), mode="reflect")
    return F.conv3d(x, w, padding=(0, w.shape[3] // 2, 0), groups=groups)


def resample3(x: torch.Tensor, mode: str) -> torch.Tensor:
    """mp_tools.py:81-93 (H and W only)."""
    if mode == "keep":
        return x
    if mode == "down":
        s = x.shape
        return F.avg_pool2d(x.reshape(s[0] * s[1], s[2], s[3], s[4]), 2).view(s[0], s[1], s[2], s[3] // 2, s[4] // 2)
    return x.repeat_interleave(2, dim=-1).repeat_interleave(2, dim=-2)


def ddec_topology(cfg: dict) -> dict:
    """unet_edm2_ddec_mclt_b1.py:196-254: ordered encoder / decoder stages (no attention in the default config)."""
    mc = cfg["model_channels"]
    cblock = [mc * m for m in cfg["channel_mult"]]
    cnoise = mc * cfg["channel_mult_noise"] if cfg["channel_mult_noise"] is not None else max(cblock)
    cemb = (mc * cfg["channel_mult_emb"] if cfg["channel_mult_emb"] is not None else max(cblock)) * cfg["mlp_multiplier"]
    ppf = cfg["in_psd_freqs"] // cfg["in_num_freqs"]
    enc, dec = [], []
    cout = cfg["in_channels"] + ppf + int(cfg["add_constant_channel"])
    for level, ch in enumerate(cblock):
        if level == 0:
            enc.append(dict(name="conv_in", kind="conv_in", cin=cout, cout=ch))
            cout = ch
        else:
            enc.append(dict(name=f"block{level}_down", kind="block", cin=cout, cout=cout, flavor="enc", resample="down"))
        for i in range(cfg["num_layers_per_block"]):
            enc.append(dict(name=f"block{level}_layer{i}", kind="block", cin=cout, cout=ch, flavor="enc", resample="keep"))
            cout = ch
    skips = [st["cout"] for st in enc]
    for level, ch in reversed(list(enumerate(cblock))):
        if level == len(cblock) - 1:
            dec.append(dict(name=f"block{level}_in0", kind="block", cin=cout, cout=cout, flavor="dec", resample="keep", skip_in=0))
            if cfg["double_midblock"]:
                dec.append(dict(name=f"block{level}_in1", kind="block", cin=cout, cout=cout, flavor="dec", resample="keep", skip_in=0))
        else:
            dec.append(dict(name=f"block{level}_up", kind="block", cin=cout, cout=cout, flavor="dec", resample="up", skip_in=0))
        for i in range(cfg["num_layers_per_block"] + 1):
            sk = skips.pop()
            dec.append(dict(name=f"block{level}_layer{i}", kind="block", cin=cout + sk, cout=ch, flavor="dec", resample="keep", skip_in=sk))
            cout = ch
    return dict(cblock=cblock, cnoise=cnoise, cemb=cemb, ppf=ppf, enc=enc, dec=dec, cout_last=cout)


def ddec_param_shapes(cfg: dict) -> dict:
    topo = ddec_topology(cfg)
    mm, cemb = cfg["mlp_multiplier"], topo["cemb"]
    shapes = {"out_gain": (), "emb_noise.weight": (cemb, topo["cnoise"]), "logvar_linear.weight": (1, cfg["logvar_channels"]),
              "conv_out.weight": (cfg["out_channels"], topo["cout_last"], 2, 3, 3)}
    for side in ("enc", "dec"):
        for st in topo[side]:
            pre = f"{side}.{st['name']}"
            if st["kind"] == "conv_in":
                shapes[f"{pre}.weight"] = (st["cout"], st["cin"], 2, 3, 3)
                continue
            cres_in = st["cout"] if st["flavor"] == "enc" else st["cin"]
            shapes[f"{pre}.conv_res0.weight"] = (st["cout"] * mm, cres_in // cfg["mlp_groups"], 1, 3, 3)
            shapes[f"{pre}.conv_res1.weight"] = (st["cout"], st["cout"] * mm // cfg["mlp_groups"], 1, 3, 3)
            shapes[f"{pre}.conv_skip.weight"] = (st["cout"], st["cin"], 2, 1, 1)
            shapes[f"{pre}.emb_linear.weight"] = (st["cout"] * mm, cemb // cfg["emb_linear_groups"], 1, 1, 1)
            shapes[f"{pre}.emb_gain"] = ()
    return shapes


def random_ddec_state(cfg: dict, seed: int, gain_value: float = 0.7) -> dict:
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for key, shape in sorted(ddec_param_shapes(cfg).items()):
        sd[key] = torch.tensor(gain_value) if shape == () else rms_normalize(torch.randn(shape, generator=g))
    topo = ddec_topology(cfg)
    sd["emb_fourier.freqs"], sd["emb_fourier.phases"] = fourier_tables(topo["cnoise"])
    sd["logvar_fourier.freqs"], sd["logvar_fourier.phases"] = fourier_tables(cfg["logvar_channels"])
    return sd


def ddec_block(sd: dict, pre: str, x: torch.Tensor, emb: torch.Tensor, st: dict, cfg: dict, clip: float = 256.0) -> torch.Tensor:
    """Block.forward, unet_edm2_ddec_mclt_b1.py:127-173 (no attention, dropout 0)."""
    G = cfg["mlp_groups"]
    x = resample3(x, st["resample"])
    if st["flavor"] == "enc":
        x = conv3d_mp(x, sd[f"{pre}.conv_skip.weight"])
        x = rms_normalize(x, dims=[1])
    y = conv3d_mp(silu_mp(x), sd[f"{pre}.conv_res0.weight"], groups=G)
    c = conv3d_mp(emb, sd[f"{pre}.emb_linear.weight"], gain=sd[f"{pre}.emb_gain"], groups=cfg["emb_linear_groups"]) + 1.0
    y = silu_mp(y * c)
    y = conv3d_mp(y, sd[f"{pre}.conv_res1.weight"], groups=G)
    if st["flavor"] == "dec":
        x = conv3d_mp(x, sd[f"{pre}.conv_skip.weight"])
    x = sum_mp(x, y, cfg["res_balance"])
    return x.clamp(-clip, clip)


def ddec_forward(sd: dict, cfg: dict, x_in: torch.Tensor, sigma: torch.Tensor, x_ref: torch.Tensor,
                 perturbed_input: Optional[torch.Tensor] = None, compute_dtype: torch.dtype = torch.bfloat16,
                 collect: Optional[dict] = None) -> torch.Tensor:
    """DDec_MCLT_UNet_B1.forward, unet_edm2_ddec_mclt_b1.py:275-326 (in_channels_emb = 0: no label embedding, no mp_silu on emb)."""
    topo = ddec_topology(cfg)
    sdata = cfg["sigma_data"]
    sig = sigma.float().view(-1, 1, 1, 1, 1)
    c_skip = sdata ** 2 / (sig ** 2 + sdata ** 2)
    c_out = sig * sdata / torch.sqrt(sig ** 2 + sdata ** 2)
    c_in = 1 / torch.sqrt(sdata ** 2 + sig ** 2)
    c_noise = sig.flatten().log() / 4
    B = x_in.shape[0]
    xr = x_ref.view(B, x_ref.shape[1], cfg["in_num_freqs"], topo["ppf"], x_ref.shape[3]).permute(0, 3, 1, 2, 4).to(compute_dtype)
    src = perturbed_input if perturbed_input is not None else x_in
    x = (c_in * src.reshape(B, cfg["in_channels"], -1, src.shape[2], src.shape[3])).to(compute_dtype)
    emb = conv3d_mp(fourier_mp(c_noise, sd["emb_fourier.freqs"], sd["emb_fourier.phases"]), sd["emb_noise.weight"])
    emb = emb[:, :, None, None, None].to(compute_dtype)
    parts = (x, xr, torch.ones_like(x[:, :1])) if cfg["add_constant_channel"] else (x, xr)
    x = torch.cat(parts, dim=1)
    skips = []
    for st in topo["enc"]:
        pre = f"enc.{st['name']}"
        x = conv3d_mp(x, sd[f"{pre}.weight"]) if st["kind"] == "conv_in" else ddec_block(sd, pre, x, emb, st, cfg)
        skips.append(x)
        if collect is not None:
            collect[pre] = x
    for st in topo["dec"]:
        pre = f"dec.{st['name']}"
        if st["skip_in"]:
            x = cat_mp(x, skips.pop(), cfg["concat_balance"])
        x = ddec_block(sd, pre, x, emb, st, cfg)
        if collect is not None:
            collect[pre] = x
    x = conv3d_mp(x, sd["conv_out.weight"], gain=sd["out_gain"])
    d_x = c_skip * x_in.float().reshape(B, cfg["out_channels"], -1, x_in.shape[2], x_in.shape[3]) + c_out * x.float()
    return d_x.reshape(B, d_x.shape[1] * d_x.shape[2], d_x.shape[3], d_x.shape[4])
