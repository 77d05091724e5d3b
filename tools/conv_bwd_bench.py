"""Micro-benchmark of the conv backward kernels (data gradient through the forward kernels on transposed weights, weight
gradient kernel) at layer shapes of the default UNet (GPU box only)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd import _lib as L  # noqa: E402
from dualdiffusion_amd import ops  # noqa: E402

# name: (B, H, W, Cin, Cout, groups, ksize)
CASES = {
    "L0_res0": (4, 32, 688, 256, 512, 8, 3),
    "L0_res1": (4, 32, 688, 512, 256, 8, 3),
    "L0_up_res0": (4, 32, 688, 512, 1024, 8, 3),
    "L1_res0_dec": (4, 16, 344, 1280, 1024, 8, 3),
    "L2_res0": (4, 8, 172, 768, 1536, 8, 3),
    "L4_res0": (4, 2, 43, 1280, 2560, 8, 3),
    "L0_skip": (4, 32, 688, 768, 256, 1, 1),
    "L3_qk": (4, 4, 86, 1024, 2048, 1, 1),
    "L1_skip": (4, 16, 344, 1280, 512, 1, 1),
    "L2_skip": (4, 8, 172, 1792, 768, 1, 1),
    "L0_skip512": (4, 32, 688, 512, 512, 1, 1),
    "L0_skip_b8": (8, 32, 688, 768, 256, 1, 1),
}


def timeit(fn, iters=20):
    for _ in range(3):
        fn()
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(iters):
        fn()
    e1.record()
    torch.cuda.synchronize()
    return e0.elapsed_time(e1) / iters * 1e3


def main():
    dt = torch.bfloat16
    for name, (B, H, W, Cin, Cout, G, ks) in CASES.items():
        x = torch.randn(B, H, W, Cin, device="cuda").to(dt)
        dy = torch.randn(B, H, W, Cout, device="cuda").to(dt)
        w = torch.randn(Cout, Cin // G, ks, ks, device="cuda")
        pw = ops.wprep(w, G, dt, npix=B * H * W)
        pw_t = ops.wprep(w, G, dt, npix=B * H * W, transpose=True)
        out = torch.empty(B, H, W, Cout, device="cuda", dtype=dt)
        dx = torch.empty(B, H, W, Cin, device="cuda", dtype=dt)
        dw = torch.empty(Cout, Cin // G, ks, ks, device="cuda")
        fl = 2.0 * B * H * W * Cout * (Cin // G) * ks * ks
        t_f = timeit(lambda: ops.conv2d(x, pw, out=out))
        t_d = timeit(lambda: ops.conv2d(dy, pw_t, out=dx))
        t_w = timeit(lambda: ops.conv2d_wgrad(dy, x, G, ks, out=dw))
        print(f"{name:12s} {fl / 1e9:7.2f} GFLOP   fwd {t_f:7.1f} us {fl / t_f / 1e6:6.0f} TF   dgrad {t_d:7.1f} us {fl / t_d / 1e6:6.0f} TF   "
              f"wgrad {t_w:7.1f} us {fl / t_w / 1e6:6.0f} TF")


if __name__ == "__main__":
    main()
