"""What do the library GEMMs (hipBLASLt / rocBLAS through torch.mm) take on the UNet's mid-size 1x1 layers?  (GPU box; reference point
for the hand-written kernels: out[M, N] = x[M, K] @ w[N, K]^T in bf16, graph of back-to-back launches.)"""
import torch

SHAPES = {"L4_v": (344, 1280, 1280), "L4_qkv": (344, 3840, 1280), "L4_dec_skip": (344, 1280, 2560), "L3_v": (1376, 1024, 1024),
          "L3_qkv": (1376, 3072, 1024), "L3_skip_cat": (1376, 1024, 2304), "L2_skip_cat": (5504, 768, 1792), "L1_skip_cat": (22016, 512, 1280),
          "L0_skip_256": (88064, 256, 256)}
for name, (M, N, K) in SHAPES.items():
    x = torch.randn(M, K, device="cuda").bfloat16()
    ws = [torch.randn(N, K, device="cuda").bfloat16() for _ in range(8)]
    out = torch.empty(M, N, device="cuda", dtype=torch.bfloat16)
    for w in ws:
        torch.mm(x, w.t(), out=out)
    torch.cuda.synchronize()
    g = torch.cuda.CUDAGraph()
    s = torch.cuda.Stream()
    with torch.cuda.stream(s):
        with torch.cuda.graph(g):
            for i in range(48):
                torch.mm(x, ws[i % 8].t(), out=out)
    torch.cuda.synchronize()
    best = 1e9
    for _ in range(3):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); g.replay(); e1.record(); torch.cuda.synchronize()
        best = min(best, e0.elapsed_time(e1) / 48 * 1e3)
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}: {best:7.1f} us  {2.0 * M * N * K / best / 1e6:7.1f} TFLOP/s")
