"""Micro-benchmark of single conv layers of the default UNet through the C ABI (GPU box only).

    python tools/conv_bench.py [--cases name,name] [--iters 50] [--dtype bf16]

Used to iterate on conv_mfma.hip and as the target command for rocprofv3 --pmc runs.
"""
import argparse
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd import _lib as L  # noqa: E402
from dualdiffusion_amd import ops  # noqa: E402

# name: (B, H, W, C0, C1, Cout, groups, ksize, resample, prologue, residual)
CASES = {
    "L0_res0_enc": (4, 32, 688, 256, 0, 512, 8, 3, 0, L.PRO_SILU, False),
    "L0_res1_enc": (4, 32, 688, 512, 0, 256, 8, 3, 0, L.PRO_SCALE_SILU, True),
    "L0_up_res0": (4, 32, 688, 512, 0, 1024, 8, 3, 1, L.PRO_SILU, False),
    "L0_up_res1": (4, 32, 688, 1024, 0, 512, 8, 3, 0, L.PRO_SCALE_SILU, True),
    "L0_up_res1_raw": (4, 32, 688, 1024, 0, 512, 8, 3, 0, L.PRO_NONE, True),
    "L0_res0_enc_raw": (4, 32, 688, 256, 0, 512, 8, 3, 0, L.PRO_NONE, False),
    "L1_res1_raw": (4, 16, 344, 1024, 0, 512, 8, 3, 0, L.PRO_NONE, True),
    "L0_res1_enc_raw": (4, 32, 688, 512, 0, 256, 8, 3, 0, L.PRO_NONE, True),
    "L0_dec_res0_raw": (4, 32, 688, 512, 256, 512, 8, 3, 0, L.PRO_NONE, False),
    "L0_up_res0_raw": (4, 32, 688, 512, 0, 1024, 8, 3, 1, L.PRO_NONE, False),
    "L1_dec_res0_raw": (4, 16, 344, 768, 512, 1024, 8, 3, 0, L.PRO_NONE, False),
    "L2_res0_raw": (4, 8, 172, 768, 0, 1536, 8, 3, 0, L.PRO_NONE, False),
    "L2_res1_raw": (4, 8, 172, 1536, 0, 768, 8, 3, 0, L.PRO_NONE, True),
    "L3_res0_raw": (4, 4, 86, 1024, 0, 2048, 8, 3, 0, L.PRO_NONE, False),
    "L0_skip_cat_raw": (4, 32, 688, 512, 256, 256, 1, 1, 0, L.PRO_NONE, False),
    "L3_res1_raw": (4, 4, 86, 2048, 0, 1024, 8, 3, 0, L.PRO_NONE, True),
    "L3_dec_res0_raw": (4, 4, 86, 1280, 1024, 2048, 8, 3, 0, L.PRO_NONE, False),
    "L3_v_raw": (4, 4, 86, 1024, 0, 1024, 1, 1, 0, L.PRO_NONE, False),
    "L3_skip_cat_raw": (4, 4, 86, 1280, 1024, 1024, 1, 1, 0, L.PRO_NONE, False),
    "L3_proj_raw": (4, 4, 86, 1024, 0, 1024, 1, 1, 0, L.PRO_NONE, True),
    "L1_skip_cat_raw": (4, 16, 344, 768, 512, 512, 1, 1, 0, L.PRO_NONE, False),
    "L2_skip_cat_raw": (4, 8, 172, 1024, 768, 768, 1, 1, 0, L.PRO_NONE, False),
    "L0_skip_512_raw": (4, 32, 688, 512, 0, 512, 1, 1, 0, L.PRO_NONE, False),
    "L0_skip_512": (4, 32, 688, 512, 0, 512, 1, 1, 1, L.PRO_NONE, False),
    "L0_skip_cat": (4, 32, 688, 512, 256, 256, 1, 1, 0, L.PRO_NONE, False),
    "L1_res0_dec": (4, 16, 344, 768, 512, 1024, 8, 3, 0, L.PRO_SILU, False),
    "L1_res1": (4, 16, 344, 1024, 0, 512, 8, 3, 0, L.PRO_SCALE_SILU, True),
    "L2_res0": (4, 8, 172, 768, 0, 1536, 8, 3, 0, L.PRO_SILU, False),
    "L3_res0": (4, 4, 86, 1024, 0, 2048, 8, 3, 0, L.PRO_SILU, False),
    "L4_res0": (4, 2, 43, 1280, 0, 2560, 8, 3, 0, L.PRO_SILU, False),
    "L4_qk": (4, 2, 43, 1280, 0, 2560, 1, 1, 0, L.PRO_SCALE, False),
    "L4_proj": (4, 2, 43, 1280, 0, 1280, 1, 1, 0, L.PRO_SCALE_SILU, True),
    "L3_qk": (4, 4, 86, 1024, 0, 2048, 1, 1, 0, L.PRO_SCALE, False),
    "L0_skip_256_raw": (4, 32, 688, 256, 0, 256, 1, 1, 0, L.PRO_NONE, False),
    "L3_qkv": (4, 4, 86, 1024, 0, 3072, 1, 1, 0, L.PRO_SCALE, False),
    "L4_qkv": (4, 2, 43, 1280, 0, 3840, 1, 1, 0, L.PRO_SCALE, False),
    "L3_skip_dec": (4, 4, 86, 1280, 1024, 1024, 1, 1, 0, L.PRO_NONE, False),
    "L4_skip_dec": (4, 2, 43, 1280, 1280, 1280, 1, 1, 0, L.PRO_NONE, False),
    "L4_proj_raw": (4, 2, 43, 1280, 0, 1280, 1, 1, 0, L.PRO_NONE, True),
    "L4_v_raw": (4, 2, 43, 1280, 0, 1280, 1, 1, 0, L.PRO_NONE, False),
    "L4_res0_raw": (4, 2, 43, 1280, 0, 2560, 8, 3, 0, L.PRO_NONE, False),
    "L4_res1_raw": (4, 2, 43, 2560, 0, 1280, 8, 3, 0, L.PRO_NONE, True),
    "L4_qkv_raw": (4, 2, 43, 1280, 0, 3840, 1, 1, 0, L.PRO_NONE, False),
    "L3_qkv_raw": (4, 4, 86, 1024, 0, 3072, 1, 1, 0, L.PRO_NONE, False),
    "L4_dec_res0_raw": (4, 2, 43, 1280, 1280, 2560, 8, 3, 0, L.PRO_NONE, False),
    "L4_dec_skip_raw": (4, 2, 43, 1280, 1280, 1280, 1, 1, 0, L.PRO_NONE, False),
    "L3_up_res0_raw": (4, 4, 86, 1280, 0, 2560, 8, 3, 1, L.PRO_NONE, False),
    "L3_up_res1_raw": (4, 4, 86, 2560, 0, 1280, 8, 3, 0, L.PRO_NONE, True),
    "L3_up_skip_raw": (4, 4, 86, 1280, 0, 1280, 1, 1, 1, L.PRO_NONE, False),
    "L3_dec_res1_raw": (4, 4, 86, 2048, 0, 1024, 8, 3, 0, L.PRO_NONE, True),
    "L4_down_res0_raw": (4, 2, 43, 1024, 0, 2048, 8, 3, 0, L.PRO_NONE, False),
    "L4_down_skip_raw": (4, 2, 43, 1024, 0, 1024, 1, 1, 0, L.PRO_NONE, False),
    # remaining 3x3 / 1x1 shapes of the default UNet at B=4 (round 3: per-layer table of the whole LDS-DMA family)
    "L0_dec1_res0_raw": (4, 32, 688, 256, 256, 512, 8, 3, 0, L.PRO_NONE, False),
    "L1_down_res0_raw": (4, 16, 344, 256, 0, 512, 8, 3, 0, L.PRO_NONE, False),
    "L1_down_res1_raw": (4, 16, 344, 512, 0, 256, 8, 3, 0, L.PRO_NONE, True),
    "L1_enc_res0_raw": (4, 16, 344, 512, 0, 1024, 8, 3, 0, L.PRO_NONE, False),
    "L1_up_res0_raw": (4, 16, 344, 768, 0, 1536, 8, 3, 1, L.PRO_NONE, False),
    "L1_up_res1_raw": (4, 16, 344, 1536, 0, 768, 8, 3, 0, L.PRO_NONE, True),
    "L1_dec1_res0_raw": (4, 16, 344, 512, 512, 1024, 8, 3, 0, L.PRO_NONE, False),
    "L1_dec2_res0_raw": (4, 16, 344, 512, 256, 1024, 8, 3, 0, L.PRO_NONE, False),
    "L2_down_res0_raw": (4, 8, 172, 512, 0, 1024, 8, 3, 0, L.PRO_NONE, False),
    "L2_down_res1_raw": (4, 8, 172, 1024, 0, 512, 8, 3, 0, L.PRO_NONE, True),
    "L2_up_res0_raw": (4, 8, 172, 1024, 0, 2048, 8, 3, 1, L.PRO_NONE, False),
    "L2_up_res1_raw": (4, 8, 172, 2048, 0, 1024, 8, 3, 0, L.PRO_NONE, True),
    "L2_dec_res0_raw": (4, 8, 172, 1024, 768, 1536, 8, 3, 0, L.PRO_NONE, False),
    "L2_dec1_res0_raw": (4, 8, 172, 768, 768, 1536, 8, 3, 0, L.PRO_NONE, False),
    "L2_dec2_res0_raw": (4, 8, 172, 768, 512, 1536, 8, 3, 0, L.PRO_NONE, False),
    "L0_skip_cat2_raw": (4, 32, 688, 256, 256, 256, 1, 1, 0, L.PRO_NONE, False),
    "L0_up_skip_raw": (4, 32, 688, 512, 0, 512, 1, 1, 1, L.PRO_NONE, False),
    "L1_skip_256_raw": (4, 16, 344, 256, 0, 512, 1, 1, 0, L.PRO_NONE, False),
    "L1_skip_512_raw": (4, 16, 344, 512, 0, 512, 1, 1, 0, L.PRO_NONE, False),
    "L1_up_skip_raw": (4, 16, 344, 768, 0, 768, 1, 1, 1, L.PRO_NONE, False),
    "L1_skip_cat1_raw": (4, 16, 344, 512, 512, 512, 1, 1, 0, L.PRO_NONE, False),
    "L1_skip_cat2_raw": (4, 16, 344, 512, 256, 512, 1, 1, 0, L.PRO_NONE, False),
    "L2_skip_512_raw": (4, 8, 172, 512, 0, 768, 1, 1, 0, L.PRO_NONE, False),
    "L2_skip_768_raw": (4, 8, 172, 768, 0, 768, 1, 1, 0, L.PRO_NONE, False),
    "L2_up_skip_raw": (4, 8, 172, 1024, 0, 1024, 1, 1, 1, L.PRO_NONE, False),
    "L2_skip_cat1_raw": (4, 8, 172, 768, 768, 768, 1, 1, 0, L.PRO_NONE, False),
    "L2_skip_cat2_raw": (4, 8, 172, 768, 512, 768, 1, 1, 0, L.PRO_NONE, False),
    # 1x1 shapes whose unit count divides the 256 CUs exactly (256 x 256 tiles): what the kernel does without tile quantisation
    "Q0_skip_256_raw": (4, 32, 512, 256, 0, 256, 1, 1, 0, L.PRO_NONE, False),
    "Q0_skip_cat_raw": (4, 32, 512, 512, 256, 256, 1, 1, 0, L.PRO_NONE, False),
    "Q1_skip_512_raw": (4, 16, 512, 512, 0, 512, 1, 1, 0, L.PRO_NONE, False),
    "Q1_skip_cat_raw": (4, 16, 512, 768, 512, 512, 1, 1, 0, L.PRO_NONE, False),
}
# the 3x3 layers of the default UNet that run on the LDS-DMA kernel (levels 0-2) and its 1x1 layers of levels 0-2, in network order
DMA3 = ["L0_res0_enc_raw", "L0_res1_enc_raw", "L1_down_res0_raw", "L1_down_res1_raw", "L1_enc_res0_raw", "L1_res1_raw", "L2_down_res0_raw",
        "L2_down_res1_raw", "L2_res0_raw", "L2_res1_raw", "L2_up_res0_raw", "L2_up_res1_raw", "L2_dec_res0_raw", "L2_dec1_res0_raw",
        "L2_dec2_res0_raw", "L1_up_res0_raw", "L1_up_res1_raw", "L1_dec_res0_raw", "L1_dec1_res0_raw", "L1_dec2_res0_raw", "L0_up_res0_raw",
        "L0_up_res1_raw", "L0_dec_res0_raw", "L0_dec1_res0_raw"]
ONE = ["L0_skip_256_raw", "L0_up_skip_raw", "L0_skip_cat_raw", "L0_skip_cat2_raw", "L1_skip_256_raw", "L1_skip_512_raw", "L1_up_skip_raw",
       "L1_skip_cat_raw", "L1_skip_cat1_raw", "L1_skip_cat2_raw", "L2_skip_512_raw", "L2_skip_768_raw", "L2_up_skip_raw", "L2_skip_cat_raw",
       "L2_skip_cat1_raw", "L2_skip_cat2_raw", "L3_qkv_raw", "L3_v_raw", "L3_proj_raw", "L3_skip_cat_raw", "L3_up_skip_raw"]
SMALL_M = ["L3_res0_raw", "L3_res1_raw", "L3_dec_res0_raw", "L3_up_res0_raw", "L3_up_res1_raw", "L3_v_raw", "L3_skip_cat_raw", "L3_proj_raw",
           "L3_qkv_raw", "L3_up_skip_raw", "L4_res0_raw", "L4_res1_raw", "L4_dec_res0_raw", "L4_down_res0_raw", "L4_qkv_raw", "L4_proj_raw",
           "L4_v_raw", "L4_dec_skip_raw", "L4_down_skip_raw"]


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--cases", default=",".join(CASES))
    ap.add_argument("--iters", type=int, default=30)
    ap.add_argument("--dtype", default="bf16")
    ap.add_argument("--path", default="auto", help="auto | mfma | dma | sm | gemm | dma16 (LDS-DMA kernel on channel-blocked tensors) | both; a+b runs several")
    ap.add_argument("--cold", type=int, default=0, help="N > 0: cycle through N distinct prepared-weight buffers inside the timed graph (weights stream from HBM as in the network, instead of staying cache-resident)")
    ap.add_argument("--cold-act", type=int, default=0, help="N > 0: cycle through N distinct input / output tensor sets inside the timed graph (operands stream from HBM as in the network)")
    ap.add_argument("--batch", type=int, default=0, help="override the case's batch size")
    ap.add_argument("--epi", default="plain", help="plain | real (conv_res0: activated output with channel scales; conv_res1: residual + activated twin)")
    a = ap.parse_args()
    dt = torch.bfloat16 if a.dtype == "bf16" else torch.float32
    dev = "cuda"
    for name in ({"small": SMALL_M, "dma3": DMA3, "one": ONE}.get(a.cases) or a.cases.split(",")):
        B, H, W, C0, C1, Cout, G, ks, rs, pro, has_res = CASES[name]
        B = a.batch or B
        sh, sw = (H // 2, W // 2) if rs == 1 else ((H * 2, W * 2) if rs == 2 else (H, W))
        a0 = torch.randn(B, sh, sw, C0, device=dev).to(dt)
        a1 = torch.randn(B, sh, sw, C1, device=dev).to(dt) if C1 else None
        w = torch.randn(Cout, (C0 + C1) // G, ks, ks, device=dev)
        cs = torch.rand(B, C0 + C1, device=dev) + 0.5
        res = torch.randn(B, H, W, Cout, device=dev).to(dt) if has_res else None
        out = torch.empty(B, H, W, Cout, device=dev, dtype=dt)
        pw_std = ops.wprep(w, G, dt, npix=B * H * W)
        pw_sm = ops.wprep(w, G, dt, CK=16) if (dt == torch.bfloat16 and ((C0 + C1) // G) % 16 == 0) else None
        raw = name.endswith('_raw')
        kw = dict(out_hw=(H, W), src1=a1, scale0=1.0 if raw else 0.8, scale1=1.0 if raw else 1.1, resample=rs, prologue=pro,
                  chan_scale=cs if pro & L.PRO_SCALE else None, residual=res, res_t=0.3, clip=256.0, out=out)
        if a.epi == "real" and raw:
            if has_res:
                kw.update(out2=torch.empty_like(out), out2_scale=0.8)
            else:
                kw.update(out_act=True, out_scale=torch.rand(B, Cout, device=dev) + 0.5, clip=0.0)
        for path in (["mfma", "dma"] if a.path == "both" else a.path.split("+")):
            kw["path"] = "dma" if path in ("dma16", "dmaw", "dma16w") else path      # ..w: weights chunked by the kernel's 16-channel stage
            pw = pw_sm if path in ("sm", "dmaw", "dma16w") else pw_std
            blk = path in ("dma16", "dma16w")
            a0 = ops.mark_c16(a0, blk)
            if a1 is not None:
                ops.mark_c16(a1, blk)
            ops.mark_c16(out, blk and not has_res)
            if kw.get("out2") is not None:
                ops.mark_c16(kw["out2"], blk)
            try:
                for _ in range(3):
                    ops.conv2d(a0, pw, **kw)
            except (RuntimeError, AttributeError) as e:
                print(f"{name:16s} {path:5s} unsupported ({e})")
                continue
            torch.cuda.synchronize()
            # the host cannot enqueue a 10 us kernel every 10 us through ctypes: time a hipGraph of `iters` back-to-back launches
            pws = [pw]
            if a.cold > 0:
                pws = [ops.wprep(torch.randn_like(w), G, dt, CK=pw.CK) for _ in range(a.cold)]
            sets = [(a0, kw)]
            for _ in range(max(a.cold_act - 1, 0)):
                kwc = dict(kw)
                kwc["out"] = torch.empty_like(out)
                if kw.get("src1") is not None:
                    kwc["src1"] = torch.randn_like(kw["src1"].float()).to(dt)
                if kw.get("residual") is not None:
                    kwc["residual"] = torch.randn_like(kw["residual"].float()).to(dt)
                if kw.get("out2") is not None:
                    kwc["out2"] = torch.empty_like(kw["out2"])
                sets.append((torch.randn_like(a0.float()).to(dt), kwc))
            plan = L.Plan()
            with plan.record():
                for it in range(a.iters):
                    ai, kwi = sets[it % len(sets)]
                    ops.conv2d(ai, pws[it % len(pws)], **kwi)
            cap = torch.cuda.Stream()
            plan.graph_build(cap.cuda_stream)
            cap.synchronize()
            plan.graph_launch()
            torch.cuda.synchronize()
            best = 1e30
            for _ in range(3):
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                plan.graph_launch()
                e1.record()
                torch.cuda.synchronize()
                best = min(best, e0.elapsed_time(e1) / a.iters * 1e3)
            us = best
            fl = 2.0 * B * H * W * Cout * ((C0 + C1) // G) * ks * ks
            by = (a0.numel() + (a1.numel() if C1 else 0) + out.numel() * (2 if has_res else 1)) * a0.element_size()
            print(f"{name:16s} {path:5s} {us:9.1f} us  {fl / 1e9:8.2f} GFLOP  {fl / us / 1e6:8.1f} TFLOP/s  {by / us / 1e3:8.1f} GB/s (algorithmic)")


if __name__ == "__main__":
    main()
