"""Per-op profile of the default VAE's (config/models/default/vae.json: 96 x (1,2,3,5) x 3) decode and encode plans at the 45 s mel
size (GPU box only); the totals must read 10.10 / 4.48 TFLOP per sample."""
import os, sys
import collections
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd.modules.formats.spectrogram import SpectrogramFormat, SpectrogramFormatConfig  # noqa: E402
from dualdiffusion_amd.modules.vaes.vae_edm2 import AutoencoderKL_EDM2, DualDiffusionVAE_EDM2Config  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 1
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from default_configs import DEFAULT_VAE, VAE_DECODE_TFLOP, VAE_ENCODE_TFLOP  # noqa: E402
vae = AutoencoderKL_EDM2(DualDiffusionVAE_EDM2Config(**DEFAULT_VAE)).requires_grad_(False).train(False).to(device="cuda", dtype=torch.bfloat16)
vae.normalize_weights()
for n, p in vae.named_parameters():
    if p.ndim == 0: p.data.fill_(0.7)
fmt = SpectrogramFormat(SpectrogramFormatConfig()).to(device="cuda")
lat = torch.randn(B, 4, 32, 688, device="cuda")
mel = torch.randn(B, 2, 256, 5504, device="cuda")
with torch.no_grad():
    emb = vae.get_embeddings(torch.randn(B, DEFAULT_VAE["label_dim"], device="cuda"))
    for _ in range(2): out = vae.decode(lat.bfloat16(), emb, fmt)
    for _ in range(2): enc = vae.encode(mel.bfloat16(), emb, fmt)
torch.cuda.synchronize()
for key, eng in vae._engines.items():
    prof = eng.pb.fplan.profile(reps=3)
    fam = collections.OrderedDict()
    tot = 0.0
    for i, (tag, fl, by, ms) in enumerate(prof):
        if tag in ("fork", "join"): continue
        tot += ms
        print(f"# op {i:3d} {tag:14s} {ms * 1e3:9.1f} us {fl / 1e9:9.2f} GFLOP {fl / max(ms, 1e-9) / 1e9:8.1f} TFLOP/s {by / max(ms, 1e-9) / 1e6:8.1f} GB/s")
    tf = sum(p[1] for p in prof) / 1e12
    print(key, "total ms", tot, "TFLOP", tf, "=", tf / B, "per sample (expected", VAE_ENCODE_TFLOP if key[0] == "enc" else VAE_DECODE_TFLOP, "+ the zero-padded 8-channel input / output convs)")
