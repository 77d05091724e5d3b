"""How long does a hipGraph of N trivial kernels take per kernel?  (launch-gap floor of the plan replay)"""
import os, sys, time
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd import _lib as L, ops
a = torch.zeros(256, device="cuda"); o = torch.zeros(256, device="cuda")
for N in (50, 200, 800):
    plan = L.Plan()
    with plan.record():
        for _ in range(N):
            ops.lincomb3(o, a, 1.0)
    plan.run(); torch.cuda.synchronize()
    cap = torch.cuda.Stream(); plan.graph_build(cap.cuda_stream); cap.synchronize()
    for _ in range(3): plan.graph_launch()
    torch.cuda.synchronize(); t0 = time.perf_counter()
    reps = 20
    for _ in range(reps): plan.graph_launch()
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / reps
    t0 = time.perf_counter()
    for _ in range(reps): plan.run()
    torch.cuda.synchronize(); de = (time.perf_counter() - t0) / reps
    print(f"N={N}: graph {dt * 1e6:.0f} us = {dt / N * 1e6:.2f} us per kernel; eager replay {de / N * 1e6:.2f} us per kernel")
