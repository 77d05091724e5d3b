"""Timing of the FGLA phase reconstruction (sample_to_raw) and of the mel-STFT at the 45 s stereo size (GPU box only)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd.modules.formats import spectrogram as _sp  # noqa: E402
from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig  # noqa: E402

_sp.FGLA_FUSED_ITER = os.environ.get("FGLA_FUSED", "1") != "0"      # 0: the three-launch loop (A/B of ddx_fgla_iter)

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 50
fmt = SpectrogramFormat(SpectrogramFormatConfig()).to(device="cuda")
audio = torch.randn(B, 2, 256 * 5503, device="cuda") * 0.1
mel = fmt.raw_to_sample(audio)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(3): mel = fmt.raw_to_sample(audio)
torch.cuda.synchronize(); t1 = time.perf_counter()
out = fmt.sample_to_raw(mel, n_fgla_iters=4, quiet=True)
torch.cuda.synchronize(); t2 = time.perf_counter()
out = fmt.sample_to_raw(mel, n_fgla_iters=iters, quiet=True)
torch.cuda.synchronize(); t3 = time.perf_counter()
print(f"B={B}: mel-STFT {(t1 - t0) / 3 * 1e3:.1f} ms; FGLA {(t3 - t2) / iters * 1e3:.2f} ms per iteration ({iters} iterations, {t3 - t2:.2f} s)")
