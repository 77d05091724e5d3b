"""Timing of the dual-window mel-spectrogram kernel (MS_MDCT_DualFormat.raw_to_mel_spec) at the 45 s stereo size (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd.modules.formats.ms_mdct_dual import MS_MDCT_DualFormat, MS_MDCT_DualFormatConfig  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
fmt = MS_MDCT_DualFormat(MS_MDCT_DualFormatConfig()).to(device="cuda")
audio = torch.randn(B, 2, fmt.get_raw_crop_width(), device="cuda") * 0.1
for _ in range(2):
    mel = fmt.raw_to_mel_spec(audio)
torch.cuda.synchronize()
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
e0.record()
n = 5
for _ in range(n):
    mel = fmt.raw_to_mel_spec(audio)
e1.record()
torch.cuda.synchronize()
ms = e0.elapsed_time(e1) / n
by = (audio.numel() + mel.numel()) * 4
print(f"ms_mel_spec B={B} {tuple(audio.shape)} -> {tuple(mel.shape)}: {ms:.2f} ms = {ms / B:.3f} ms/sample; algorithmic bytes {by / 1e6:.0f} MB -> {by / ms / 1e6:.1f} GB/s; "
      f"2 FFT-4096 per frame: {2 * 5 * 4096 * 12 * mel.shape[-1] * B / ms / 1e9:.2f} TFLOP/s fp32")
