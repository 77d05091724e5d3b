"""Timing of one diffusion-decoder forward (default DDec_MCLT_UNet_B1, bf16) at the full mel resolution (GPU box only)."""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd.modules.unets.unet_edm2_ddec_mclt_b1 import DDec_MCLT_UNet_B1, DDec_MCLT_UNet_B1_Config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
W = int(sys.argv[2]) if len(sys.argv) > 2 else 5504
cfg = DDec_MCLT_UNet_B1_Config()
unet = DDec_MCLT_UNet_B1(cfg).requires_grad_(False).train(False).to(device="cuda", dtype=torch.bfloat16)
unet.normalize_weights()
for n, p in unet.named_parameters():
    if p.ndim == 0:
        p.data.fill_(0.7)
x = torch.randn(B, 2, cfg.in_num_freqs, W, device="cuda")
xr = torch.randn(B, 2, cfg.in_psd_freqs, W, device="cuda").abs()
sig = torch.full((B,), 1.5, device="cuda")
for _ in range(2):
    out = unet(x, sig, None, None, x_ref=xr)
torch.cuda.synchronize()
t0 = time.perf_counter()
n = 5
for _ in range(n):
    out = unet(x, sig, None, None, x_ref=xr)
torch.cuda.synchronize()
ms = (time.perf_counter() - t0) / n * 1e3
assert torch.isfinite(out).all()
print(f"ddec forward B={B} (2,{cfg.in_num_freqs},{W}) bf16: {ms:.1f} ms/step = {B / ms * 1e3:.2f} samples/s; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")
