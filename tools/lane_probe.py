"""Do two half-batch chains of small-M conv launches on two plan lanes (parallel hipGraph branches) beat one full-batch chain?

    python tools/lane_probe.py [--level 4] [--path sm|auto]

Chain = the conv layers of one attention block (conv_res0, conv_res1, merged qkv, proj) repeated `--blocks` times, dependent launches.
Prints us per chain for: B on one lane, B/2 on one lane, 2 x B/2 on two lanes, (4 x B/4 on ... not available: two lanes only).
"""
import argparse
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd import _lib as L  # noqa: E402
from dualdiffusion_amd import ops  # noqa: E402


def make_chain(B, H, W, C, path, blocks, dev="cuda"):
    dt = torch.bfloat16
    ck = dict(CK=16) if path == "sm" else dict(npix=B * H * W)
    w0 = ops.wprep(torch.randn(2 * C, C // 8, 3, 3, device=dev), 8, dt, **ck)
    w1 = ops.wprep(torch.randn(C, 2 * C // 8, 3, 3, device=dev), 8, dt, **ck)
    wq = ops.wprep(torch.randn(3 * C, C, 1, 1, device=dev), 1, dt, **ck)
    wp = ops.wprep(torch.randn(C, C, 1, 1, device=dev), 1, dt, **ck)
    x = torch.randn(B, H, W, C, device=dev).to(dt)
    y0 = torch.empty(B, H, W, 2 * C, device=dev, dtype=dt)
    x1 = torch.empty(B, H, W, C, device=dev, dtype=dt)
    tw = torch.empty(B, H, W, C, device=dev, dtype=dt)
    qkv = torch.empty(B, H, W, 3 * C, device=dev, dtype=dt)
    x2 = torch.empty(B, H, W, C, device=dev, dtype=dt)
    cs = torch.rand(B, 2 * C, device=dev) + 0.5
    keep = [w0, w1, wq, wp, x, y0, x1, tw, qkv, x2, cs]
    p = None if path == "auto" else path

    def run():
        for _ in range(blocks):
            ops.conv2d(x, w0, out_act=True, out_scale=cs, out=y0, path=p or "auto")
            ops.conv2d(y0, w1, residual=x, res_t=0.3, clip=256.0, out=x1, out2=tw, out2_scale=0.8, path=p or "auto")
            ops.conv2d(x1, wq, out=qkv, path=p or "auto")
            ops.conv2d(qkv[..., :C].contiguous() if False else x1, wp, residual=x1, res_t=0.3, clip=256.0, out=x2, out2=tw, out2_scale=0.8, path=p or "auto")
    return run, keep


def time_plan(plan, reps=5):
    cap = torch.cuda.Stream()
    plan.graph_build(cap.cuda_stream)
    cap.synchronize()
    plan.graph_launch()
    torch.cuda.synchronize()
    best = 1e30
    for _ in range(reps):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        plan.graph_launch()
        e1.record()
        torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) * 1e3)
    return best


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--level", type=int, default=4)
    ap.add_argument("--path", default="auto")
    ap.add_argument("--blocks", type=int, default=8)
    ap.add_argument("--batch", type=int, default=4)
    a = ap.parse_args()
    H, W, C = {4: (2, 43, 1280), 3: (4, 86, 1024)}[a.level]
    B = a.batch
    lib = L.lib()
    full, k0 = make_chain(B, H, W, C, a.path, a.blocks)
    halfa, k1 = make_chain(B // 2, H, W, C, a.path, a.blocks)
    halfb, k2 = make_chain(B // 2, H, W, C, a.path, a.blocks)
    for f in (full, halfa, halfb):
        f()
    torch.cuda.synchronize()
    n = 4 * a.blocks
    res = {}
    plan = L.Plan()
    with plan.record():
        full()
    res[f"B={B} one lane"] = time_plan(plan)
    plan = L.Plan()
    with plan.record():
        halfa()
    res[f"B={B // 2} one lane"] = time_plan(plan)
    plan = L.Plan()
    with plan.record():
        halfa()
        halfb()
    res[f"2 x B={B // 2} one lane (serial)"] = time_plan(plan)
    plan = L.Plan()
    with plan.record():
        L.check(lib.ddx_plan_fork(), "fork")
        halfb()
        L.check(lib.ddx_plan_main(), "main")
        halfa()
        L.check(lib.ddx_plan_join(), "join")
    res[f"2 x B={B // 2} two lanes"] = time_plan(plan)
    for k, v in res.items():
        print(f"level {a.level} path {a.path:5s} {k:32s} {v:9.1f} us per chain of {n} launches = {v / n:6.2f} us per launch")


if __name__ == "__main__":
    main()
