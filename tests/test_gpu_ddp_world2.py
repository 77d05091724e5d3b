"""BASELINE configs[3] at world size 2 with the REAL differentiation engine: two processes on the one GPU of the test box, each running
UNetTrainStep (HIP forward / backward, two-bucket gradient exchange with the decoder's bucket sent from the backward's hook, fused parameter pass)
on its half of a global batch; the collective is torch.distributed's `gloo` on device tensors (RCCL refuses two ranks on one device -- the RCCL
path itself is exercised at world size 1 in test_gpu_backward.py and by the driver's multi-GPU bench).  The data-parallel semantics under test are
the reference's (src/training/trainer.py:375,1008-1067; unet_trainer.py:197-198,246): every rank ends the step with the same weights, and they are
the weights a single process reaches on the whole global batch."""
import os
import socket

import pytest
import torch
import torch.multiprocessing as mp

from oracle import edm2_oracle as O
from tests.util import rel_l2

pytestmark = pytest.mark.gpu

OVER = dict(model_channels=256, channel_mult=(1, 2), attn_levels=(1,), channels_per_head=64, num_layers_per_block=1, in_channels_emb=64, logvar_channels=32)


class _Fmt:
    def __init__(self):
        from dualdiffusion_amd.modules.formats.frequency_scale import FrequencyScale
        self.ms_freq_scale = FrequencyScale("mel", 20.0, 16000.0, 32000, 3201, 256)


def _batch():
    g = torch.Generator().manual_seed(41)
    B, H, W = 4, 16, 32
    return (torch.randn(B, 4, H, W, generator=g), torch.randn(B, 64, generator=g), torch.tensor([0.3, 1.1, 4.0, 0.7]),
            torch.randn(B, 4, H, W, generator=g), torch.tensor([True, False, True, True]))


def _make_step(mode=None):
    from dualdiffusion_amd.modules.unets.unet_edm2_b4 import UNet, UNetConfig
    from dualdiffusion_amd.training.optimizer import EMASpec, LRScheduleConfig, OptimizerConfig
    from dualdiffusion_amd.training.train_step import UNetTrainStep
    cfg = O.unet_cfg(**OVER)
    sd = O.random_unet_state(cfg, seed=9, gain_value=0.3)
    unet = UNet(UNetConfig(**OVER)).requires_grad_(False)
    unet.load_state_dict(sd, strict=True)
    unet = unet.to(device="cuda", dtype=torch.float32).train(True)
    emas = [EMASpec(name="e", tensors={k: p.data.clone() for k, p in unet.named_parameters()}, beta=0.9)]
    ts = UNetTrainStep(unet, _Fmt(), OptimizerConfig(dynamic_max_grad_norm_z=None), LRScheduleConfig(learning_rate=5e-4, lr_warmup_steps=1, lr_reference_steps=1000),
                       emas=emas, grad_exchange=mode)
    ts.global_step = 1
    return unet, ts


def _worker(rank: int, world: int, port: int, out_dir: str):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(world), LOCAL_RANK="0", MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    import torch.distributed as dist
    dist.init_process_group("gloo", rank=rank, world_size=world)
    torch.cuda.set_device(0)
    unet, ts = _make_step()
    samples, clap, sigma, noise, mask = _batch()
    sl = slice(rank, None, world)            # the reference's per-rank stride of the global batch (unet_trainer.py:246)
    outs = []
    for _ in range(2):
        o = ts.step(samples[sl], clap[sl], sigma[sl], noise[sl], mask[sl])
        outs.append((o["loss"].clone().cpu(), float(o["grad_norm"])))
    w = {k: p.data.clone().cpu() for k, p in unet.named_parameters()}
    ema = {k: v.clone().cpu() for k, v in ts.ema_state()["e"].items()}
    torch.save(dict(outs=outs, w=w, ema=ema), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


def test_ddp_world2_real_engine_matches_the_single_process_global_batch(tmp_path):
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    r0, r1 = (torch.load(tmp_path / f"rank{r}.pt") for r in range(2))
    # identical replicas after the steps (same summed gradients, same parameter pass)
    worst_rep = max(rel_l2(r1["w"][k], r0["w"][k]) for k in r0["w"] if r0["w"][k].ndim > 0)
    assert worst_rep < 1e-6, worst_rep
    assert abs(r0["outs"][1][1] - r1["outs"][1][1]) <= 1e-5 * abs(r0["outs"][1][1])      # one global gradient norm
    # single process on the whole global batch: same loss values per sample, same weights and EMA after two steps
    unet, ts = _make_step()
    samples, clap, sigma, noise, mask = _batch()
    outs = []
    for _ in range(2):
        o = ts.step(samples, clap, sigma, noise, mask)
        outs.append((o["loss"].clone().cpu(), float(o["grad_norm"])))
    for step in range(2):
        both = torch.empty(4)
        both[0::2], both[1::2] = r0["outs"][step][0], r1["outs"][step][0]
        assert rel_l2(both, outs[step][0]) < (1e-4 if step == 0 else 2e-3), (step, both, outs[step][0])
        assert abs(r0["outs"][step][1] - outs[step][1]) <= 5e-3 * abs(outs[step][1]), (r0["outs"][step][1], outs[step][1])
    errs = {k: rel_l2(r0["w"][k], p.data.cpu()) for k, p in unet.named_parameters() if p.ndim > 0}
    worst = max(errs.items(), key=lambda kv: kv[1])
    e_ema = max(rel_l2(r0["ema"][k], v.cpu()) for k, v in ts.ema_state()["e"].items() if v.ndim > 0)
    print(f"world 2 (gloo, real engine) vs one process on the global batch: worst weight {worst[0]} {worst[1]:.2e}, EMA {e_ema:.2e}, replicas {worst_rep:.1e}")
    assert worst[1] < 2e-3 and e_ema < 2e-3
