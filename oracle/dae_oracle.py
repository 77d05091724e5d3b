"""CPU restatement of the reference's DAE_G1 autoencoder (TEST INFRASTRUCTURE ONLY -- only tests/, bench.py's cpu_baseline leg and
__graft_entry__.smoke() may import this package).

Follows /root/reference/src/modules/daes/dae_edm2_g1.py: `MPConv3D_E` :68-126 (eval mode), `Block` :128-233 (encoder flavour:
(1,3,3) kernels; decoder flavour: (2,3,3) kernels over the stereo depth pair; optional attention folded over (b, z, w) with the
tokens along h, :209-228), `DAE_G1` :235-427 (encode, decode, tiled_encode).  5-D activations (B, C, 2, H, W), reflection padding
on W, zero padding on H, a reflected depth row behind (for depth 2: the other channel).  Pinned by tools/make_golden.py gen_dae
against the reference module (tests/golden/dae_g1_small.safetensors).
"""
from __future__ import annotations

import math
from typing import Optional

import torch
import torch.nn.functional as F

from .ddec_oracle import resample3
from .edm2_oracle import rms_normalize, silu_mp, sum_mp

DEFAULT_DAE_G1_CFG = dict(
    in_channels=1, out_channels=1, in_channels_emb=1024, in_num_freqs=256, latent_channels=4, model_channels=32, channel_mult_enc=1,
    channel_mult_dec=(1, 2, 4, 8), channel_mult_emb=4, num_attn_heads=8, num_enc_layers=6, num_dec_layers_per_block=3, res_balance=0.3,
    attn_balance=0.3, attn_levels=(), mlp_multiplier=2, mlp_groups=1, emb_linear_groups=1, add_constant_channel=True, add_pixel_norm=False)


def dae_cfg(**over) -> dict:
    cfg = dict(DEFAULT_DAE_G1_CFG)
    cfg.update(over)
    return cfg


def conv3d_e(x: torch.Tensor, w: torch.Tensor, gain=1.0, groups: int = 1) -> torch.Tensor:
    """MPConv3D_E.forward, eval mode (:95-121)."""
    w = w.float() * (gain / math.sqrt(w[0].numel()))
    w = w.to(x.dtype)
    if w.ndim == 2:
        return x @ w.t()
    kz, kw = w.shape[2], w.shape[4]
    if kz // 2 or kw // 2:
        x = F.pad(x, (kw // 2, kw // 2, 0, 0, 0, kz // 2), mode="reflect")
    return F.conv3d(x, w, padding=(0, w.shape[3] // 2, 0), groups=groups)


def dae_topology(cfg: dict) -> dict:
    mc = cfg["model_channels"]
    cemb = mc * cfg["channel_mult_emb"] * cfg["mlp_multiplier"] if cfg["in_channels_emb"] > 0 else 0
    enc_ch = mc * cfg["channel_mult_enc"]
    dec_ch = [mc * m for m in cfg["channel_mult_dec"]]
    L = len(dec_ch)
    enc = [f"block0_layer{i}" for i in range(cfg["num_enc_layers"])]
    dec, cin = [], dec_ch[-1]
    for level in reversed(range(L)):
        cout = dec_ch[level]
        first = f"block{level}_in0" if level == L - 1 else f"block{level}_up"
        dec.append(dict(name=first, cin=cin, cout=cout, resample="keep" if level == L - 1 else "up", attn=level in cfg["attn_levels"]))
        for i in range(cfg["num_dec_layers_per_block"]):
            dec.append(dict(name=f"block{level}_layer{i}", cin=cout, cout=cout, resample="keep", attn=level in cfg["attn_levels"]))
        cin = cout
    return dict(cemb=cemb, enc_ch=enc_ch, dec_ch=dec_ch, enc=enc, dec=dec, levels=L, cout_last=cin)


def dae_param_shapes(cfg: dict) -> dict:
    t = dae_topology(cfg)
    mm, g = cfg["mlp_multiplier"], cfg["mlp_groups"]
    sh = {"out_gain": (), "recon_loss_logvar": ()}
    if cfg["in_channels_emb"] > 0:
        sh["emb_label.weight"] = (t["cemb"], cfg["in_channels_emb"])
    cin0 = 1 + int(cfg["add_constant_channel"])
    e = t["enc_ch"]
    sh["enc.conv_in.weight"] = (e, cin0, 1, 5, 5)
    for n in t["enc"]:
        sh[f"enc.{n}.conv_res0.weight"] = (e * mm, e // g, 1, 3, 3)
        sh[f"enc.{n}.conv_res1.weight"] = (e, e * mm // g, 1, 3, 3)
        if g > 1:
            sh[f"enc.{n}.conv_skip.weight"] = (e, e, 1, 1, 1)
        sh[f"enc.{n}.emb_gain"] = ()
    sh["conv_latents_out.weight"] = (cfg["latent_channels"], e, 1, 3, 3)
    sh["conv_latents_in.weight"] = (t["dec_ch"][-1], cfg["latent_channels"] + int(cfg["add_constant_channel"]), 2, 3, 3)
    for d in t["dec"]:
        p, ci, co = f"dec.{d['name']}", d["cin"], d["cout"]
        sh[f"{p}.conv_res0.weight"] = (co * mm, ci // g, 2, 3, 3)
        sh[f"{p}.conv_res1.weight"] = (co, co * mm // g, 2, 3, 3)
        if ci != co or g > 1:
            sh[f"{p}.conv_skip.weight"] = (co, ci, 1, 1, 1)
        sh[f"{p}.emb_gain"] = ()
        if t["cemb"]:
            sh[f"{p}.emb_linear.weight"] = (co * mm, t["cemb"] // cfg["emb_linear_groups"], 1, 1, 1)
        if d["attn"]:
            sh[f"{p}.attn_qkv.weight"] = (co * 3, co, 1, 1, 1)
            sh[f"{p}.attn_proj.weight"] = (co, co, 1, 1, 1)
    sh["conv_out.weight"] = (cfg["out_channels"], t["cout_last"], 1, 5, 5)
    return sh


def random_dae_state(cfg: dict, seed: int, gain_value: float = 0.7) -> dict:
    """randn per key (sorted), weight-normalised over all dims but 0 (MPConv3D_E.normalize_weights, norm_dim = 1 means `dim=1`?
    -- no: normalize(w, dim=norm_dim) with norm_dim = 1 normalises over the INPUT-channel axis only; reproduced here)."""
    g = torch.Generator().manual_seed(seed)
    sd = {}
    for k, shape in sorted(dae_param_shapes(cfg).items()):
        if shape == ():
            sd[k] = torch.tensor(1.0 if k == "out_gain" else (0.0 if k == "recon_loss_logvar" else gain_value))
        else:
            w = torch.randn(shape, generator=g)
            sd[k] = rms_normalize(w, dims=[1])
    return sd


def axis_attention(qkv: torch.Tensor, heads: int) -> torch.Tensor:
    """:211-224: qkv (B, 3C, Z, H, W) -> attention along H for every (b, z, w) column; channel index = head * (d * 3) + dd * 3 + s."""
    B, C3, Z, H, W = qkv.shape
    q = qkv.permute(0, 2, 4, 1, 3).reshape(B * Z * W, heads, -1, 3, H)
    q, k, v = rms_normalize(q, dims=[2]).unbind(3)
    y = F.scaled_dot_product_attention(q.transpose(-1, -2), k.transpose(-1, -2), v.transpose(-1, -2)).transpose(-1, -2)
    return y.reshape(B, Z, W, C3 // 3, H).permute(0, 3, 1, 4, 2).contiguous()


def dae_block(sd: dict, p: str, x: torch.Tensor, emb: Optional[torch.Tensor], *, flavor: str, resample: str, cfg: dict, attn: bool) -> torch.Tensor:
    g = cfg["mlp_groups"]
    x = resample3(x, resample)
    skip = sd.get(f"{p}.conv_skip.weight")
    if flavor == "enc":
        if skip is not None:
            x = conv3d_e(x, skip)
        if cfg["add_pixel_norm"]:
            x = rms_normalize(x, dims=[1])
    y = conv3d_e(silu_mp(x), sd[f"{p}.conv_res0.weight"], groups=g)
    if f"{p}.emb_linear.weight" in sd and emb is not None:
        c = conv3d_e(emb, sd[f"{p}.emb_linear.weight"], gain=sd[f"{p}.emb_gain"], groups=cfg["emb_linear_groups"]) + 1.0
        y = silu_mp(y * c)
    else:
        y = silu_mp(y)
    y = conv3d_e(y, sd[f"{p}.conv_res1.weight"], groups=g)
    if flavor == "dec" and skip is not None:
        x = conv3d_e(x, skip)
    x = sum_mp(x, y, cfg["res_balance"])
    if attn:
        y = axis_attention(conv3d_e(x, sd[f"{p}.attn_qkv.weight"]), cfg["num_attn_heads"])
        y = conv3d_e(silu_mp(y), sd[f"{p}.attn_proj.weight"])
        x = sum_mp(x, y, cfg["attn_balance"])
    return x.clamp(-256, 256)


def dae_embeddings(sd: dict, emb_in: torch.Tensor) -> Optional[torch.Tensor]:
    """:305-309."""
    if "emb_label.weight" not in sd:
        return None
    return conv3d_e(rms_normalize(emb_in.float()), sd["emb_label.weight"])


def dae_encode(sd: dict, cfg: dict, x: torch.Tensor, emb: Optional[torch.Tensor] = None, normalize_latents: bool = True,
               collect: Optional[dict] = None) -> torch.Tensor:
    """:331-349."""
    t = dae_topology(cfg)
    x = x.float().reshape(x.shape[0], 1, -1, x.shape[2], x.shape[3])
    if cfg["add_constant_channel"]:
        x = torch.cat((x, torch.ones_like(x[:, :1])), dim=1)
    x = conv3d_e(x, sd["enc.conv_in.weight"])
    if collect is not None:
        collect["enc.conv_in"] = x
    for n in t["enc"]:
        x = dae_block(sd, f"enc.{n}", x, None, flavor="enc", resample="keep", cfg=cfg, attn=False)
        if collect is not None:
            collect[f"enc.{n}"] = x
    z = conv3d_e(x, sd["conv_latents_out.weight"])
    z = z.reshape(z.shape[0], z.shape[1] * z.shape[2], z.shape[3], z.shape[4])
    z = F.avg_pool2d(z, 2 ** (t["levels"] - 1))
    return rms_normalize(z) if normalize_latents else z


def dae_decode(sd: dict, cfg: dict, z: torch.Tensor, emb: Optional[torch.Tensor], collect: Optional[dict] = None) -> torch.Tensor:
    """:351-364."""
    t = dae_topology(cfg)
    x = z.float().reshape(z.shape[0], cfg["latent_channels"], -1, z.shape[2], z.shape[3])
    if cfg["add_constant_channel"]:
        x = torch.cat((x, torch.ones_like(x[:, :1])), dim=1)
    x = conv3d_e(x, sd["conv_latents_in.weight"])
    e5 = emb[:, :, None, None, None] if emb is not None else None
    for d in t["dec"]:
        x = dae_block(sd, f"dec.{d['name']}", x, e5, flavor="dec", resample=d["resample"], cfg=cfg, attn=d["attn"])
        if collect is not None:
            collect[f"dec.{d['name']}"] = x
    y = conv3d_e(x, sd["conv_out.weight"], gain=sd["out_gain"])
    return y.reshape(y.shape[0], y.shape[1] * y.shape[2], y.shape[3], y.shape[4])


def dae_tiled_encode(sd: dict, cfg: dict, x: torch.Tensor, emb, max_chunk: int = 6144, overlap: int = 256) -> torch.Tensor:
    """:375-427."""
    ds = 2 ** (len(cfg["channel_mult_dec"]) - 1)
    x_w = x.shape[-1]
    if x_w <= max_chunk:
        return dae_encode(sd, cfg, x, emb)
    min_chunk, out_ov = overlap * 3, overlap // ds
    lat = torch.zeros(x.shape[0], cfg["latent_channels"] * 2, x.shape[-2] // ds, x_w // ds)
    for w_start in range(0, x_w, max_chunk - overlap * 2):
        c0, c1 = max(0, w_start), min(x_w, w_start + max_chunk)
        if c1 - c0 < min_chunk:
            c0 -= min_chunk - (c1 - c0)
        lc = dae_encode(sd, cfg, x[..., c0:c1], emb, normalize_latents=False)
        first, last = w_start == 0, c1 == x_w
        vs, ve = (0 if first else out_ov), (lc.shape[3] if last else lc.shape[3] - out_ov)
        ds0, ds1 = (c0 // ds if first else c0 // ds + out_ov), (c1 // ds if last else c1 // ds - out_ov)
        lat[:, :, :, ds0:ds1] = lc[:, :, :, vs:ve]
    return rms_normalize(lat)
