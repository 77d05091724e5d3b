"""Where does a denoise step's wall time go: hipGraph replay alone vs the module call (input copies, output clone, host work)."""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import bench as Bn  # noqa


sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from default_configs import DEFAULT_UNET  # noqa: E402


def main():
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    class Fmt: ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)
    torch.manual_seed(0)
    unet = UNet(UNetConfig(**DEFAULT_UNET)).requires_grad_(False).train(False).to(device="cuda", dtype=torch.bfloat16)
    unet.normalize_weights()
    for n, p in unet.named_parameters():
        if p.ndim == 0: p.data.fill_(0.7)
    unet.compile()
    B, H, W = 4, 32, 688
    x = torch.randn(B, 4, H, W, device="cuda"); sigma = torch.full((B,), 1.0, device="cuda")
    emb = unet.get_embeddings(torch.randn(B, 512, device="cuda"), torch.ones(B, dtype=torch.bool))
    with torch.no_grad():
        for _ in range(5): out = unet(x, sigma, Fmt(), emb)
    torch.cuda.synchronize()
    eng = next(iter(unet._engines.values()))
    n = 50
    t0 = time.perf_counter()
    for _ in range(n): eng.pb.launch(True)
    torch.cuda.synchronize(); t1 = time.perf_counter()
    with torch.no_grad():
        for _ in range(n): out = unet(x, sigma, Fmt(), emb)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    # host-only cost of a module call (GPU idle): time the call without syncing, then drain
    with torch.no_grad():
        h0 = time.perf_counter()
        for _ in range(n): out = unet(x, sigma, Fmt(), emb)
        h1 = time.perf_counter()
    torch.cuda.synchronize()
    print(f"graph replay only: {(t1 - t0) / n * 1e3:.3f} ms/step; module call: {(t2 - t1) / n * 1e3:.3f} ms/step; host time to enqueue a call: {(h1 - h0) / n * 1e3:.3f} ms")


main()
