"""Timing of the fused MSS loss kernel (value + gradient) at the mel-spectrogram size (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd.training.loss.multiscale_spectral import MSSLoss2D, MSSLoss2DConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 2
mss = MSSLoss2D(MSSLoss2DConfig(), torch.device("cuda"))
x = torch.randn(B, 2, 256, 5504, device="cuda")
y = torch.randn(B, 2, 256, 5504, device="cuda")
for need_grad in (False, True):
    for _ in range(2):
        mss._launch(x, y, need_grad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    n = 5
    for _ in range(n):
        mss._launch(x, y, need_grad)
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / n
    print(f"mss_loss B={B} (2,256,5504) grad={need_grad}: {ms:.2f} ms = {ms / B:.2f} ms/sample; algorithmic bytes {(3 if need_grad else 2) * x.numel() * 4 / 1e6:.0f} MB -> {(3 if need_grad else 2) * x.numel() * 4 / ms / 1e6:.1f} GB/s")
