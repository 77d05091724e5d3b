"""Which data-gradient convs of a UNet train batch take the fused activation-backward epilogue (ddx_mpconv2d_dgrad_act) and which fall back
to conv + silu_scale_bwd launches?  Prints one line per distinct (shape, path) with counts.   python tools/dgrad_fusion_census.py [B]"""
import collections
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd import ops  # noqa: E402
from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig  # noqa: E402
from dualdiffusion_amd.training.unet_grad import UNetTrainer  # noqa: E402
from tools.default_configs import DEFAULT_UNET  # noqa: E402
from tools.train_bench import Fmt  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 8
unet = UNet(UNetConfig(**DEFAULT_UNET)).requires_grad_(False).to(device="cuda", dtype=torch.float32).train(True)
unet.normalize_weights()
tr = UNetTrainer(unet)
g = torch.Generator(device="cuda").manual_seed(1)
H, W = 32, 688
samples, noise = torch.randn(B, 4, H, W, device="cuda", generator=g), torch.randn(B, 4, H, W, device="cuda", generator=g)
sigma = torch.exp(torch.randn(B, device="cuda", generator=g) * 1.2 - 0.4)
clap, mask = torch.randn(B, 512, device="cuda", generator=g), torch.ones(B, dtype=torch.bool, device="cuda")
tr.train_batch(samples, clap, sigma, noise, mask, Fmt())
census = collections.Counter()
orig = ops.conv2d_dgrad_act


def spy(dy, pw_t, y0, **kw):
    before = ops._dgrad_act_fused_calls
    out = orig(dy, pw_t, y0, **kw)
    fused = ops._dgrad_act_fused_calls > before
    census[(tuple(dy.shape), pw_t.Cout, pw_t.groups, pw_t.ksize, kw.get("y1") is not None, kw.get("dchan_scale") is not None, kw.get("add") is not None,
            "fused" if fused else "conv + silu_scale_bwd")] += 1
    return out


ops.conv2d_dgrad_act = spy
import dualdiffusion_amd.training.block_grad as BG  # noqa: E402
BG.ops.conv2d_dgrad_act = spy
tr.train_batch(samples, clap, sigma, noise, mask, Fmt())
for k, v in sorted(census.items(), key=lambda kv: (kv[0][-1], -kv[0][0][1] * kv[0][0][2])):
    print(f"{v:3d} x dy {k[0]} -> {k[1]} ch, groups {k[2]}, k{k[3]}, two parts {k[4]}, dchan {k[5]}, add {k[6]}: {k[7]}")
