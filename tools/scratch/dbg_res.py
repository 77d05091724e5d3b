import os, sys, torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__)))))
from dualdiffusion_amd import ops
g = torch.Generator(device="cuda").manual_seed(3)
for (B, H, W, C0, C1, Cout, G, res) in [(2, 40, 200, 128, 0, 128, 2, True), (2, 40, 200, 64, 64, 64, 2, False), (4, 32, 344, 256, 0, 512, 8, False),
                                         (4, 32, 688, 256, 0, 512, 8, False), (4, 32, 700, 512, 0, 256, 8, True)]:
    a0 = torch.randn(B, H, W, C0, device="cuda", generator=g).bfloat16()
    a1 = torch.randn(B, H, W, C1, device="cuda", generator=g).bfloat16() if C1 else None
    w = torch.randn(Cout, (C0 + C1) // G, 3, 3, device="cuda", generator=g)
    r = torch.randn(B, H, W, Cout, device="cuda", generator=g).bfloat16() if res else None
    cs = torch.rand(B, Cout, device="cuda", generator=g) + 0.5
    pw = ops.wprep(w, G, torch.bfloat16, npix=B * H * W)
    kw = dict(src1=a1, residual=r, res_t=0.3, clip=256.0) if res else dict(src1=a1, out_act=True, out_scale=cs)
    tw_d, tw_m = torch.zeros(B, H, W, Cout, device="cuda", dtype=torch.bfloat16), torch.zeros(B, H, W, Cout, device="cuda", dtype=torch.bfloat16)
    y_d = ops.conv2d(a0, pw, path="dma", out2=tw_d if res else None, **kw)
    y_m = ops.conv2d(a0, pw, path="mfma", out2=tw_m if res else None, **kw)
    torch.cuda.synchronize()
    d = (y_d.float() - y_m.float())
    e = float(d.norm() / y_m.float().norm())
    # where are the errors: per image / per tile row / per channel block
    per_b = [float(d[b].norm() / y_m[b].float().norm()) for b in range(B)]
    per_g = [float(d[..., k * (Cout // G):(k + 1) * (Cout // G)].norm() / y_m[..., k * (Cout // G):(k + 1) * (Cout // G)].float().norm()) for k in range(G)]
    print((B, H, W, C0, C1, Cout, G, res), "err", e, "per image", per_b, "per group", per_g, flush=True)
    if e > 1e-2:
        bad = (d.abs() > 0.1 * y_m.float().abs().mean()).any(dim=3)   # [B][H][W]
        print("  bad rows (h):", sorted(set(bad.nonzero()[:, 1].tolist()))[:40])
        print("  bad cols (w) count:", len(set(bad.nonzero()[:, 2].tolist())), sorted(set(bad.nonzero()[:, 2].tolist()))[:40])
    if res:
        print("  twin err", float((tw_d.float() - tw_m.float()).norm() / tw_m.float().norm()))
