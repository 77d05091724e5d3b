import torch
SHAPES = {"L0_256": (88064, 256, 256), "L0_cat2": (88064, 256, 512), "L0_cat": (88064, 256, 768), "L0_up(L1)": (22016, 512, 512),
          "L1_cat": (22016, 512, 1280), "L1_cat1": (22016, 512, 1024), "L1_cat2": (22016, 512, 768), "L1_512": (22016, 512, 512), "L1_256":(22016,512,256)}
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
    by = 2.0*(M*K+M*N)
    print(f"{name:12s} M={M:6d} N={N:5d} K={K:5d}: {best:7.1f} us  {2.0 * M * N * K / best / 1e6:7.1f} TFLOP/s {by/best/1e6:7.2f} TB/s")
