"""A/B of the 1024-pixel (8-fragment) conv_dma variant against the default tiles (GPU box only).

    DDX_DMA_BIG=0 python tools/conv_big_ab.py --save /tmp/ab.pt
    DDX_DMA_BIG=2 python tools/conv_big_ab.py --check /tmp/ab.pt

The knob is read once per process, so the two variants run in two processes: the first stores its outputs, the second
compares (same inputs from the same seeds; both must agree to bf16 rounding of the same fp32 accumulation order per element).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd import ops  # noqa: E402

# name: (B, H, W, C0, C1, Cout, groups, resample, residual, out_act+chan_scale, twin)
CASES = {
    "L0 32->64 x8": (4, 32, 688, 256, 0, 512, 8, 0, False, True, False),
    "L0 64->64 x8": (4, 32, 688, 512, 0, 512, 8, 0, True, False, True),
    "L0 96->64 x8 cat": (4, 32, 688, 512, 256, 512, 8, 0, False, True, False),
    "L0 64->128 x8 up": (4, 32, 688, 512, 0, 1024, 8, 1, False, True, False),
    "L0 64->64 x8 plain": (4, 32, 688, 512, 0, 512, 8, 0, False, False, False),
    "L0 64->64 x8 res": (4, 32, 688, 512, 0, 512, 8, 0, True, False, False),
    "L0 64->64 x8 act": (4, 32, 688, 512, 0, 512, 8, 0, False, True, False),
    "L0 64->64 x8 twin": (4, 32, 688, 512, 0, 512, 8, 0, False, False, True),
    "L0 32->64 x8 plain": (4, 32, 688, 256, 0, 512, 8, 0, False, False, False),
    "L0 64->32 x8 plain": (4, 32, 688, 512, 0, 256, 8, 0, False, False, False),
    "L0 64->32 x8 res twin": (4, 32, 688, 512, 0, 256, 8, 0, True, False, True),
    "L0 128->64 x8 plain": (4, 32, 688, 1024, 0, 512, 8, 0, False, False, False),
    "L0 128->64 x8 res": (4, 32, 688, 1024, 0, 512, 8, 0, True, False, True),
    "L1 128->64 x8 res": (4, 16, 344, 1024, 0, 512, 8, 0, True, False, False),
    "L1 96->128 x8": (4, 16, 344, 768, 0, 1024, 8, 0, False, True, False),
    "L1 64->128 x8 act": (4, 16, 344, 512, 0, 1024, 8, 0, False, True, False),
    "L2 96->192 x8 act": (4, 8, 172, 768, 0, 1536, 8, 0, False, True, False),
    "L2 192->96 x8 res twin": (4, 8, 172, 1536, 0, 768, 8, 0, True, False, True),
    "ragged 30x70 64->64": (2, 30, 70, 128, 0, 128, 2, 0, True, False, True),
    "ddec L0 32->32 act": (2, 256, 5504, 32, 0, 32, 1, 0, False, True, False),
    "ddec L0 32->32 res": (2, 256, 5504, 32, 0, 32, 1, 0, True, False, False),
    "ddec L0 64->32 act": (2, 256, 5504, 64, 0, 32, 1, 0, False, True, False),
    "ddec L1 64->64 res": (2, 128, 2752, 64, 0, 64, 1, 0, True, False, False),
    "ddec L2 96->96 act": (2, 64, 1376, 96, 0, 96, 1, 0, False, True, False),
    "B8 L0 64->64 x8": (8, 32, 688, 512, 0, 512, 8, 0, True, False, False),
}


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--save")
    ap.add_argument("--check")
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--cases", default="", help="comma-separated substrings; empty = all")
    ap.add_argument("--c16", action="store_true", help="channel-blocked sources / conv_res0-style outputs / twins (as the inference plans mark them)")
    a = ap.parse_args()
    dt, dev = torch.bfloat16, "cuda"
    saved = torch.load(a.check) if a.check else {}
    outs = {}
    for name, (B, H, W, C0, C1, Cout, G, rs, has_res, act, twin) in CASES.items():
        if a.cases and not any(c == name for c in a.cases.split(",")):
            continue
        g = torch.Generator(device=dev).manual_seed(hash(name) % 1000 if False else len(name) * 7 + B)
        sh, sw = (H // 2, W // 2) if rs == 1 else (H, W)
        a0 = torch.randn(B, sh, sw, C0, device=dev, generator=g).to(dt)
        a1 = torch.randn(B, sh, sw, C1, device=dev, generator=g).to(dt) if C1 else None
        w = torch.randn(Cout, (C0 + C1) // G, 3, 3, device=dev, generator=g)
        res = torch.randn(B, H, W, Cout, device=dev, generator=g).to(dt) if has_res else None
        ocs = (torch.rand(B, Cout, device=dev, generator=g) + 0.5) if act else None
        out = torch.empty(B, H, W, Cout, device=dev, dtype=dt)
        out2 = torch.empty_like(out) if twin else None
        pw = ops.wprep(w, G, dt, npix=B * H * W)
        if a.c16:
            a0 = ops.to_c16(a0)
            a1 = ops.to_c16(a1) if a1 is not None else None
            if not has_res:
                ops.mark_c16(out)
            if twin:
                ops.mark_c16(out2)
        kw = dict(out_hw=(H, W), src1=a1, resample=rs, residual=res, res_t=0.3, clip=256.0, out=out, path="dma")
        if act:
            kw.update(out_act=True, out_scale=ocs)
        if twin:
            kw.update(out2=out2, out2_scale=0.9)
        for _ in range(3):
            ops.conv2d(a0, pw, **kw)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(a.iters):
            ops.conv2d(a0, pw, **kw)
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) / a.iters * 1e3
        fl = 2.0 * B * H * W * Cout * ((C0 + C1) // G) * 9
        msg = f"{name:22s} {us:8.1f} us {fl / us / 1e6:7.1f} TFLOP/s"
        outs[name] = (out.cpu(), out2.cpu() if twin else None)
        if name in saved:
            r0, r1 = saved[name]
            d = (out.cpu().float() - r0.float()).norm() / r0.float().norm()
            msg += f"  vs saved: rel-L2 {float(d):.2e} max {float((out.cpu().float() - r0.float()).abs().max()):.3f}"
            if twin:
                msg += f" twin {float((out2.cpu().float() - r1.float()).norm() / r1.float().norm()):.2e}"
        print(msg)
    if a.save:
        torch.save(outs, a.save)


if __name__ == "__main__":
    main()
