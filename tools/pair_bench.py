"""conv_res0 -> conv_res1 of a level-0 encoder block: the fused launch (csrc/conv_pair.hip) against the two LDS-DMA launches, graph replay (GPU box).
    python tools/pair_bench.py [B] [H] [W]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd import ops  # noqa: E402

B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
H = int(sys.argv[2]) if len(sys.argv) > 2 else 32
W = int(sys.argv[3]) if len(sys.argv) > 3 else 688
dt, dev, G, Cn = torch.bfloat16, "cuda", 8, 256
torch.manual_seed(0)
x = torch.randn(B, H, W, Cn, device=dev).to(dt)
xa = (torch.nn.functional.silu(x.float()) / 0.596).to(dt)
pw0 = ops.wprep(torch.randn(2 * Cn, Cn // G, 3, 3, device=dev), G, dt, normalize=True)
pw1 = ops.wprep(torch.randn(Cn, 2 * Cn // G, 3, 3, device=dev), G, dt, normalize=True)
c = torch.rand(B, 2 * Cn, device=dev) + 0.5
y0, out, tw = torch.empty(B, H, W, 2 * Cn, device=dev, dtype=dt), torch.empty_like(x), torch.empty_like(x)


def two(twin):
    ops.conv2d(xa, pw0, out_act=True, out_scale=c, out=y0)
    ops.conv2d(y0, pw1, residual=x, res_t=0.3, clip=256.0, out=out, **(dict(out2=tw, out2_scale=1.0) if twin else {}))


def one(twin):
    ops.conv_pair(xa, pw0, pw1, c, x, 0.3, clip=256.0, out=out, out2=tw if twin else None, out2_scale=1.0)


def graph_time(fn, reps=20):
    fn(); fn()
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    with torch.cuda.graph(g):
        for _ in range(reps):
            fn()
    g.replay()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        g.replay()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / (5 * reps) * 1e3


px = B * H * W
gflop = 2 * px * (2 * Cn) * (Cn // G) * 9 * 2 / 1e9
for twin in (False, True):
    t2, t1 = graph_time(lambda: two(twin)), graph_time(lambda: one(twin))
    mb = px * Cn * 2 * (4 if twin else 3) / 1e6
    print(f"B={B} {H}x{W} twin={twin}: two launches {t2:7.1f} us, fused {t1:7.1f} us ({gflop / t1 * 1e3 / 1e3:.0f} TFLOP/s, {mb / t1:.2f} TB/s of algorithmic bytes)")
