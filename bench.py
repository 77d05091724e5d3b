"""bench.py -- UNet denoise steps/sec on synthetic 45 s stereo mel latents (BASELINE.json metric).

    python bench.py --gpus 1 --steps 20 --warmup 5
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P bench.py --gpus N ...

One step = one `UNet.forward` (reference src/modules/unets/unet_edm2_b4.py:250-296) of the default EDM2 UNet
(config/models/default/unet.json shape: 293 M parameters, 489.3 GFLOP per sample) on a batch of B=4 latents
(4, 32, 688), bf16 storage / fp32 accumulate, CLAP-conditioned, through the HIP launch plan (hipGraph).
N > 1: independent replicas, one process per GPU, no data-path collective ("replicas only");
value = all ranks' steps / max-over-ranks time.

Extra objects on the JSON line:
  roofline     -- the dominant kernel family (3x3 grouped implicit-GEMM conv on MFMA): algorithmic FLOPs per launch
                  / mean launch duration, measured here with hipEvents on the launch stream (ddx_plan_profile)
  cpu_baseline -- the CPU oracle (fp32 restatement of the reference, oracle/edm2_oracle.py) timed on this box's host
                  cores on a bounded sample (rank 0, N=1 only)
"""
from __future__ import annotations

import argparse
import json
import os
import statistics
import sys
import time

import torch

ROOT = os.path.dirname(os.path.abspath(__file__))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

FLOP_PER_SAMPLE = 489.3e9  # SURVEY.md 8d (FlopCounterMode on the reference, default config, latent 4x32x688)
PEAK_BF16_TFLOPS = 2500.0  # MI355X dense bf16 MFMA peak (MI355X_MICROARCH.md)

DEFAULT_UNET = dict(in_channels=4, out_channels=4, in_channels_emb=512, dropout=0.0, sigma_max=200.0, sigma_min=0.03,
                    sigma_data=1.0, model_channels=256, logvar_channels=128, channel_mult=[1, 2, 3, 4, 5], channel_mult_noise=1,
                    channel_mult_emb=3, channels_per_head=64, num_layers_per_block=2, label_balance=0.5, concat_balance=0.5,
                    res_balance=0.3, attn_balance=0.3, attn_levels=[3, 4], mlp_multiplier=2, mlp_groups=8)


def csrc_hash() -> str:
    """sha256 over the kernel sources and the C ABI header: ties a committed PMC traffic number to the code that produced it."""
    import glob
    import hashlib
    h = hashlib.sha256()
    for f in sorted(glob.glob(os.path.join(ROOT, "dualdiffusion_amd", "csrc", "*.h*")) + [os.path.join(ROOT, "include", "ddx_hip.h")]):
        with open(f, "rb") as fh:
            h.update(os.path.basename(f).encode() + b"\0" + fh.read())
    return h.hexdigest()


def build_model(device, dtype, seed: int):
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    torch.manual_seed(seed)
    unet = UNet(UNetConfig(**DEFAULT_UNET)).requires_grad_(False).train(False)   # weights ~ randn (reference init)
    with torch.no_grad():
        for p in unet.parameters():
            if p.ndim == 0:
                p.fill_(0.7)   # gains are zero-initialised in the reference: a fresh model would skip the body
    unet = unet.to(device=device, dtype=torch.float32)
    unet.normalize_weights()   # forced weight norm (reference trainer does this after every step)
    return unet.to(dtype=dtype)


def cpu_baseline(unet, fmt_range, max_seconds: float = 25.0) -> dict:
    """Time the CPU oracle on the default UNet at B=1 (fp32, all host cores)."""
    from oracle import edm2_oracle as O
    cfg = O.unet_cfg(**{k: (tuple(v) if isinstance(v, list) else v) for k, v in DEFAULT_UNET.items()})
    sd = {k: v.detach().float().cpu() for k, v in unet.state_dict().items()}
    # torch's CPU conv kernels stop scaling (and collapse when oversubscribed) long before 256 hardware threads:
    # use at most 32 threads and say so in `cores`
    cores = min(os.cpu_count() or 1, 32)
    torch.set_num_threads(cores)
    g = torch.Generator().manual_seed(1)
    x = torch.randn(1, 4, 32, 688, generator=g)
    sigma = torch.tensor([1.7])
    emb = O.unet_embeddings(sd, cfg, torch.randn(1, 512, generator=g), torch.tensor([True]))
    times = []
    t_start = time.time()
    with torch.no_grad():
        for i in range(6):
            t0 = time.time()
            O.unet_forward(sd, cfg, x, sigma, emb, freq_range=fmt_range)
            dt = time.time() - t0
            if i >= 1 or dt > 8.0:       # first call is a warm-up unless it alone is already expensive
                times.append(dt)
            if time.time() - t_start > max_seconds:
                break
    t = statistics.median(times)
    host = host_cpu()
    return {"value": 1.0 / (4.0 * t), "unit": "steps/s", "cores": torch.get_num_threads(), "host_cores": host["logical"], "host_cpu": host["model"],
            "kind": "port",
            "sample": f"default UNet fp32 forward at B=1 (489.3 GFLOP) on the CPU oracle, median of {len(times)} timed runs: "
                      f"{t:.3f} s/sample = {FLOP_PER_SAMPLE / t / 1e9:.0f} GFLOP/s; value = 1/(4*t) (B=4 step equivalent); "
                      f"{torch.get_num_threads()} torch threads on a host with {host['logical']} logical CPUs ({host['model']}) -- torch's CPU conv "
                      "kernels stop scaling beyond ~32 threads"}


def host_cpu() -> dict:
    """Logical CPU count and model string of the box the CPU baseline runs on."""
    model = "unknown"
    try:
        with open("/proc/cpuinfo") as fh:
            for ln in fh:
                if ln.lower().startswith("model name"):
                    model = ln.split(":", 1)[1].strip()
                    break
    except OSError:
        pass
    return {"logical": os.cpu_count() or 1, "model": model}


def train_main(a) -> None:
    """BASELINE.json configs[3]: data-parallel UNet training, one process per GPU, global batch = 8 x N (x accumulation steps):
    sigma for the global batch from rank 0, per-rank strided slices, local gradient accumulation, ONE flat-bucket gradient exchange over
    RCCL per optimizer step (two collectives, the decoder's overlapped with the encoder's backward), fused AdamW + EMAs + forced weight
    norm.  One `step` = one optimizer step; value = optimizer steps/s of the whole job (time = max over ranks); weak scaling."""
    from dualdiffusion_amd import distributed as D
    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
    from dualdiffusion_amd.training.optimizer import EMASpec, LRScheduleConfig, OptimizerConfig
    from dualdiffusion_amd.training.sigma_sampler import SigmaSampler, SigmaSamplerConfig
    from dualdiffusion_amd.training.train_step import UNetTrainStep
    rank, world, local_rank = D.world()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    D.init(backend="nccl", device=dev)

    class Fmt:
        ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)

    Bd = a.batch if a.batch != 4 else 8       # configs[3]: 8 per rank (global 64 on 8 GPUs)
    unet = build_model(dev, torch.float32, seed=0).train(True)          # same seed on every rank: identical replicas
    emas = [EMASpec(name="0.9999", tensors={k: p.data.clone() for k, p in unet.named_parameters()}, beta=0.9999),
            EMASpec(name="fb", tensors={k: p.data.clone() for k, p in unet.named_parameters()}, beta=0.99999, feedback_beta=0.9999)]
    step = UNetTrainStep(unet, Fmt(), OptimizerConfig(max_grad_norm=10.0), LRScheduleConfig(learning_rate=1e-2, lr_warmup_steps=4000, lr_reference_steps=20000),
                         use_graph=not a.no_graph, gradient_accumulation_steps=a.accum,
                         sigma_sampler=SigmaSampler(SigmaSamplerConfig(distribution="ln_sech", dist_offset=0.45)), conditioning_dropout=0.1, emas=emas)
    step.global_step = 100
    g = torch.Generator(device=dev).manual_seed(1 + rank)               # synthetic latents / CLAP embeddings, different per rank
    samples = torch.randn(Bd * a.accum, 4, 32, 688, device=dev, generator=g)
    clap = torch.randn(Bd * a.accum, 512, device=dev, generator=g)
    for _ in range(max(a.warmup, 1)):
        out = step.run_batch(samples, clap, generator=g)
    torch.cuda.synchronize(); D.barrier(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(a.steps):
        out = step.run_batch(samples, clap, generator=g)
    torch.cuda.synchronize(); D.barrier(); torch.cuda.synchronize()
    elapsed = time.perf_counter() - t0
    assert torch.isfinite(out["loss"]).all() and out["grad_norm"] == out["grad_norm"], "non-finite training step"
    _units, elapsed = D.replica_throughput(a.steps, elapsed)
    # all ranks must hold identical weights after identical-seed init + summed gradients
    w = torch.stack([p.data.float().sum() for p in unet.parameters()]).sum().reshape(1)
    ws = [torch.empty_like(w) for _ in range(world)] if world > 1 else [w]
    if world > 1:
        import torch.distributed as dist
        dist.all_gather(ws, w)
    if rank == 0:
        gb = Bd * a.accum * world
        sps = gb * a.steps / elapsed
        fl = 3 * FLOP_PER_SAMPLE * gb * a.steps / elapsed
        line = {"metric": "UNet training optimizer steps/sec (data parallel, RCCL gradient all-reduce)", "value": round(a.steps / elapsed, 4), "unit": "steps/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(elapsed / a.steps * 1e3, 2), "higher_is_better": True,
                "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "configs[3]: DDP training of the default EDM2 UNet (293M params), latents (8,4,32,688) per GPU per micro-step, "
                                       "stratified ln_sech sigma over the global batch, conditioning dropout 0.1, AdamW + 2 EMAs (one feedback) + forced weight norm",
                           "global_batch": gb, "per_gpu_batch": Bd, "accumulation_steps": a.accum, "parallelism": f"dp{world}", "graph": not a.no_graph,
                           "gradient_bucket_bytes": int(step.trainer.grad_flat.numel() * 4)},
                "samples_per_s": round(sps, 2), "model_tflops": round(fl / 1e12, 1), "loss_mean": round(float(out["loss"].mean()), 4),
                "grad_norm": round(float(out["grad_norm"]), 3), "replicas_identical": bool(all(torch.equal(x, ws[0]) for x in ws)),
                "roofline": {"bound": "mfma", "achieved": round(fl / 1e12 / world, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                             "frac": round(fl / 1e12 / world / PEAK_BF16_TFLOPS, 4), "traffic": None,
                             "note": "whole-step model FLOPs (3 x forward) per GPU; per-kernel rooflines: profiles/"}}
        print(json.dumps(line))
    if world > 1:
        import torch.distributed as dist
        D.barrier()
        dist.destroy_process_group()


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--gpus", type=int, default=1)
    ap.add_argument("--steps", type=int, default=20)
    ap.add_argument("--warmup", type=int, default=5)
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--no-cpu-baseline", action="store_true")
    ap.add_argument("--no-graph", action="store_true")
    ap.add_argument("--layer-table", action="store_true", help="print the per-op hipEvent profile (rank 0)")
    ap.add_argument("--mode", default="infer", choices=["infer", "train"],
                    help="infer: the BASELINE metric (UNet denoise steps/s, replicas); train: BASELINE configs[3], data-parallel optimizer steps/s")
    ap.add_argument("--accum", type=int, default=1, help="--mode train: gradient-accumulation micro-steps per optimizer step")
    a = ap.parse_args()
    if a.mode == "train":
        return train_main(a)

    from dualdiffusion_amd import distributed as D
    rank, world, local_rank = D.world()
    if world != a.gpus and world > 1:
        raise SystemExit(f"--gpus {a.gpus} but WORLD_SIZE={world}")
    torch.cuda.set_device(local_rank)
    dev = torch.device("cuda", local_rank)
    D.init(backend="nccl", device=dev)   # no-op for a single process; "nccl" is RCCL on ROCm

    from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale

    class Fmt:
        ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)

    B, H, W = a.batch, 32, 688
    unet = build_model(dev, torch.bfloat16, seed=0)
    if not a.no_graph:
        unet.compile()
    g = torch.Generator().manual_seed(1 + rank)
    sigma = torch.exp(torch.empty(B).uniform_(torch.log(torch.tensor(0.03)).item(), torch.log(torch.tensor(200.0)).item(), generator=g))
    x = (torch.randn(B, 4, H, W, generator=g) * torch.sqrt(sigma ** 2 + 1).view(-1, 1, 1, 1)).to(dev)
    sigma = sigma.to(dev)
    clap = torch.randn(B, 512, generator=g)
    fmt = Fmt()
    with torch.no_grad():
        emb = unet.get_embeddings(clap, torch.ones(B, dtype=torch.bool))
        for _ in range(a.warmup):
            out = unet(x, sigma, fmt, emb)
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(a.steps):
            out = unet(x, sigma, fmt, emb)
        torch.cuda.synchronize()
        D.barrier()
        torch.cuda.synchronize()
        elapsed = time.perf_counter() - t0
    assert torch.isfinite(out).all(), "non-finite UNet output"
    total_steps, elapsed = D.replica_throughput(a.steps, elapsed)   # steps summed over ranks, time = max over ranks

    line = None
    if rank == 0:
        ms_per_step = elapsed / a.steps * 1e3
        value = total_steps / elapsed
        # ---- roofline of the dominant kernel family, measured live with hipEvents on the launch stream
        eng = next(iter(unet._engines.values()))
        prof = eng.fplan.profile(reps=3)
        fam = {}
        for tag, fl, by, ms in prof:
            if tag in ("fork", "join"):
                continue                      # lane markers of the plan, not launches
            f = fam.setdefault(tag, [0, 0.0, 0.0, 0.0])
            f[0] += 1; f[1] += fl; f[2] += by; f[3] += ms
        dom = max(fam.items(), key=lambda kv: kv[1][3])
        n, fl, by, ms = dom[1]
        total_ms = sum(v[3] for v in fam.values())
        achieved = fl / (ms * 1e-3) / 1e12
        # HBM-side traffic per launch of the same kernel family: PMC counters cannot be read from inside the process, so
        # the number comes from the committed rocprofv3 --pmc run of this very command (tools/profile_round.sh)
        # (tools/make_traffic.py); it is only reported when the kernel sources still hash to what that run measured
        import glob
        traffic, traffic_src = None, None
        cands = sorted(glob.glob(os.path.join(ROOT, "profiles", "r*_traffic.json")))
        if cands:
            with open(cands[-1]) as fh:
                tj = json.load(fh)
            name = os.path.relpath(cands[-1], ROOT)
            if tj.get("family") != dom[0]:
                print(f"# roofline.traffic: {name} is for {tj.get('family')}, the dominant family is {dom[0]} -> null", file=sys.stderr)
            elif tj.get("csrc_sha256") != csrc_hash():
                print(f"# roofline.traffic: STALE -- {name} was measured on other kernel sources (csrc hash mismatch): re-run tools/profile_round.sh "
                      "-> null", file=sys.stderr)
                traffic_src = f"{name} is stale (kernel sources changed since its PMC run)"
            else:
                traffic, traffic_src = tj["traffic_bytes_per_launch"], f"{name} (rocprofv3 --pmc FETCH_SIZE x2 + WRITE_SIZE, same kernel sources)"
        roofline = {"bound": "mfma", "achieved": round(achieved, 1), "peak": PEAK_BF16_TFLOPS, "unit": "TFLOP/s",
                    "frac": round(achieved / PEAK_BF16_TFLOPS, 4), "traffic": traffic, "traffic_source": traffic_src,
                    "algorithmic_bytes_per_launch": int(by / n), "traffic_ratio": round(traffic / (by / n), 3) if traffic else None,
                    "kernel": dom[0], "launches_per_step": n, "avg_launch_us": round(ms / n * 1e3, 2),
                    "algorithmic_gflop_per_launch": round(fl / n / 1e9, 3), "share_of_step_time": round(ms / total_ms, 3),
                    "step_tflops": round(B * FLOP_PER_SAMPLE / (elapsed / a.steps) / 1e12, 1),
                    # model FLOPs of the reference forward (489.3 GFLOP per sample), not executed MACs: the plan runs the up blocks' skip
                    # convs before the resample (~3 % fewer MACs), so this is throughput in the reference's units, not hardware utilisation
                    "flops": "model",
                    "families_ms": {k: round(v[3], 3) for k, v in sorted(fam.items(), key=lambda kv: -kv[1][3])}}
        if a.layer_table:
            for i, (tag, fl_, by_, ms_) in enumerate(prof):
                print(f"# op {i:3d} {tag:14s} {ms_ * 1e3:9.1f} us  {fl_ / 1e9:9.3f} GFLOP  {fl_ / max(ms_, 1e-9) / 1e9:8.1f} TFLOP/s  "
                      f"{by_ / max(ms_, 1e-9) / 1e6:8.1f} GB/s", file=sys.stderr)
        line = {"metric": "UNet denoise steps/sec (45s stereo mel latent)", "value": round(value, 3), "unit": "steps/s",
                "n_gpus": world, "steps": a.steps, "warmup": a.warmup, "ms_per_step": round(ms_per_step, 3),
                "higher_is_better": True, "scaling": "weak", "vs_baseline": None, "dtype": "bf16", "data": "synthetic",
                "config": {"workload": "configs[1]: full EDM2 UNet (default unet.json, 293M params) bf16 forward, latent (B,4,32,688) = 45 s @ 32 kHz stereo, CLAP-conditioned",
                           "global_batch": B * world, "per_gpu_batch": B, "latent": [4, H, W], "parallelism": f"replicas x{world}",
                           "graph": not a.no_graph, "weights": "random-init (randn, forced weight-norm, gains 0.7)"},
                "roofline": roofline}
        if world == 1 and not a.no_cpu_baseline:
            line["cpu_baseline"] = cpu_baseline(unet, (20.0, 16000.0))
    if world > 1:
        import torch.distributed as dist
        D.barrier()
        dist.destroy_process_group()
    if rank == 0:
        print(json.dumps(line))


if __name__ == "__main__":
    main()
