"""Timing of one UNet training step (train batch + optimizer) of the default UNet on the HIP kernels (GPU box only).

    python tools/train_bench.py [B] [steps]
Eager host orchestration (dualdiffusion_amd/training): an upper bound, the launch-plan / hipGraph form is future work.
"""
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale  # noqa: E402
from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig  # noqa: E402
sys.path.insert(0, os.path.dirname(os.path.abspath(__file__)))
from default_configs import DEFAULT_UNET  # noqa: E402  (unet.json, not the dataclass defaults)
from dualdiffusion_amd.training.optimizer import LRScheduleConfig, OptimizerConfig  # noqa: E402
from dualdiffusion_amd.training.train_step import UNetTrainStep  # noqa: E402


class Fmt:
    ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)


def main():
    B = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 3
    torch.manual_seed(0)
    unet = UNet(UNetConfig(**DEFAULT_UNET)).requires_grad_(False).to(device="cuda", dtype=torch.float32).train(True)
    unet.normalize_weights()
    for n, p in unet.named_parameters():
        if p.ndim == 0:
            p.data.fill_(0.7)
    use_graph = os.environ.get("DDX_TRAIN_GRAPH", "0") != "0"
    if os.environ.get("DDX_DDP_BUCKETS", "0") == "1":    # single-GPU check of the two-bucket RCCL exchange: world_size-1 group
        import torch.distributed as dist
        dist.init_process_group("nccl", init_method="tcp://127.0.0.1:29531", rank=0, world_size=1)
    ts = UNetTrainStep(unet, Fmt(), OptimizerConfig(), LRScheduleConfig(), use_graph=use_graph)
    ts.global_step = 100
    H, W = 32, 688
    g = torch.Generator(device="cuda").manual_seed(1)
    samples = torch.randn(B, 4, H, W, device="cuda", generator=g)
    noise = torch.randn(B, 4, H, W, device="cuda", generator=g)
    sigma = torch.exp(torch.randn(B, device="cuda", generator=g) * 1.2 - 0.4)
    clap = torch.randn(B, 512, device="cuda", generator=g)
    mask = torch.ones(B, dtype=torch.bool, device="cuda")
    out = ts.step(samples, clap, sigma, noise, mask)       # warm-up (kernel attributes, allocator)
    torch.cuda.synchronize()
    times = []
    for _ in range(steps):          # every step ends with a host read of the gradient norm: time them one by one, report the MEDIAN
        t0 = time.perf_counter()    # (the GPU boxes are shared: host-side jitter of a loaded machine shows up as outlier steps)
        out = ts.step(samples, clap, sigma, noise, mask)
        torch.cuda.synchronize()
        times.append(time.perf_counter() - t0)
    dt = sorted(times)[len(times) // 2]
    # phases
    tr = ts.trainer
    torch.cuda.synchronize(); t1 = time.perf_counter()
    loss, grads = tr.train_batch(samples, clap, sigma, noise, mask, Fmt())
    t_host = time.perf_counter() - t1                   # host enqueue time of the batch (no device wait in between)
    torch.cuda.synchronize(); t2 = time.perf_counter()
    ts.opt.step({k: grads[k] for k in ts.params}, 1e-4, 250.0)
    unet.normalize_weights()
    torch.cuda.synchronize(); t3 = time.perf_counter()
    fl = 3 * 489.3e9 * B
    print(("[hipGraph replay] " if use_graph else "") + ("[bucketed rccl exchange, world 1] " if os.environ.get("DDX_DDP_BUCKETS", "0") == "1" else "") + f"UNet train step B={B} (4,{H},{W}) bf16 compute / fp32 master: {dt * 1e3:.1f} ms/step (median of {steps}; min {min(times) * 1e3:.1f}, max {max(times) * 1e3:.1f}) = {B / dt:.1f} samples/s, "
          f"{fl / dt / 1e12:.0f} TFLOP/s (3 x forward FLOPs); train batch {1e3 * (t2 - t1):.1f} ms (host enqueue {1e3 * t_host:.1f} ms), optimizer + weight norm {1e3 * (t3 - t2):.1f} ms; "
          f"loss {float(out['loss'].mean()):.4f} grad_norm {out['grad_norm']:.2f}; peak memory {torch.cuda.max_memory_allocated() / 2**30:.1f} GiB")


if __name__ == "__main__":
    main()
