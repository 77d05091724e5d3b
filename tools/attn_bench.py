"""Timing of the attention kernel at the UNet's level-3 / level-4 shapes and nearby token counts (GPU box; graph of back-to-back launches)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd import _lib as L  # noqa: E402
from dualdiffusion_amd import ops  # noqa: E402

dt = torch.bfloat16
pre = os.environ.get("ATTN_PRENORM", "1") != "0"      # operands normalised by their producer (the UNet path): the key-split kernel up to 384 tokens
for (B, H, W, heads) in ([(4, 4, 86, 16)] if os.environ.get("ATTN_ONE") else [(4, 4, 86, 16), (4, 4, 64, 16), (4, 4, 32, 16), (4, 4, 96, 16), (4, 2, 43, 20), (4, 4, 86, 20), (8, 4, 86, 16)]):
    C = heads * 64
    qkv = torch.randn(B, H, W, 3 * C, device="cuda").to(dt)
    qk, v = qkv[..., :2 * C], qkv[..., 2 * C:]
    cs = torch.rand(B, C, device="cuda") + 0.5
    out = torch.empty(B, H, W, C, device="cuda", dtype=dt)
    for _ in range(3):
        ops.attention(qk, v, heads, out=out, out_scale=cs, prenorm=pre)
    torch.cuda.synchronize()
    plan = L.Plan()
    with plan.record():
        for _ in range(40):
            ops.attention(qk, v, heads, out=out, out_scale=cs, prenorm=pre)
    cap = torch.cuda.Stream()
    plan.graph_build(cap.cuda_stream)
    cap.synchronize()
    plan.graph_launch()
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); plan.graph_launch(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 40 * 1e3)
    T = H * W
    fl = 4.0 * B * heads * T * T * 64
    print(f"B={B} T={T:4d} heads={heads}: {best:6.1f} us  {fl / best / 1e6:6.1f} TFLOP/s  ({-(-T // 128)} query tiles x {-(-T // 128)} key chunks, {B * heads * -(-T // 128)} workgroups)")
