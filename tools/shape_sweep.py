"""Robustness sweep: default UNet forward on the HIP kernels at several batch sizes / latent shapes, bf16 against the fp32 path of
the same kernels (GPU box only).  Catches tile-selection / ragged-edge problems that the fixed benchmark shape does not."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale  # noqa: E402
from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig  # noqa: E402


class Fmt:
    ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)


def main():
    torch.manual_seed(0)
    cfg = UNetConfig(model_channels=128)          # 46 M parameters: same structure, quick in fp32
    sd = None
    outs = {}
    shapes = [(1, 32, 688), (3, 32, 688), (5, 16, 80), (2, 32, 48), (7, 48, 112), (4, 64, 1376)]
    for dt in (torch.float32, torch.bfloat16):
        unet = UNet(cfg).requires_grad_(False).train(False)
        if sd is None:
            sd = {k: v.clone() for k, v in unet.state_dict().items()}
            for k in sd:
                if sd[k].ndim == 0: sd[k].fill_(0.7)
        unet.load_state_dict(sd)
        unet = unet.to(device="cuda", dtype=dt)
        unet.normalize_weights()
        for (B, H, W) in shapes:
            g = torch.Generator(device="cuda").manual_seed(B * 1000 + H + W)
            x = torch.randn(B, 4, H, W, device="cuda", generator=g)
            sigma = torch.exp(torch.randn(B, device="cuda", generator=g))
            clap = torch.randn(B, 512, device="cuda", generator=g)
            with torch.no_grad():
                emb = unet.get_embeddings(clap, torch.ones(B, dtype=torch.bool, device="cuda"))
                y = unet(x * (sigma.view(-1, 1, 1, 1) ** 2 + 1).sqrt(), sigma, Fmt(), emb)
            torch.cuda.synchronize()
            assert torch.isfinite(y).all(), (dt, B, H, W)
            outs[(dt, B, H, W)] = y.float()
    worst = 0.0
    for (B, H, W) in shapes:
        a, b = outs[(torch.float32, B, H, W)], outs[(torch.bfloat16, B, H, W)]
        e = float((a - b).norm() / a.norm())
        worst = max(worst, e)
        print(f"B={B} latent (4,{H},{W}): bf16 vs fp32 rel-L2 {e:.3e}")
    assert worst < 5e-2, worst
    print("shape sweep ok")


if __name__ == "__main__":
    main()
