"""CPU, world_size 2, gloo: the N > 1 helpers used by bench.py (replica aggregation) and by the data-parallel step
(sigma broadcast, strided slices, fused scalar gather)."""
import os
import socket

import torch
import torch.distributed as dist
import torch.multiprocessing as mp


def _free_port() -> int:
    s = socket.socket()
    s.bind(("127.0.0.1", 0))
    port = s.getsockname()[1]
    s.close()
    return port


def _worker(rank: int, ws: int, port: int, out_dir: str):
    os.environ.update(RANK=str(rank), WORLD_SIZE=str(ws), LOCAL_RANK=str(rank), MASTER_ADDR="127.0.0.1", MASTER_PORT=str(port))
    from dualdiffusion_amd import distributed as D
    assert D.init(backend="gloo")
    assert D.world() == (rank, ws, rank)
    # replicas: rank r did 10 steps in (1 + r) seconds -> whole job = 20 steps in max = 2 s
    units, secs = D.replica_throughput(10, 1.0 + rank)
    assert units == 20.0 and secs == 2.0
    # sigma for the global batch comes from rank 0 only
    g = torch.Generator().manual_seed(100 + rank)
    sigma = torch.rand(8, generator=g)
    ref0 = torch.rand(8, generator=torch.Generator().manual_seed(100))
    D.broadcast_from_rank0(sigma)
    assert torch.equal(sigma, ref0)
    # strided slices partition the global batch without overlap, two micro-steps of 2 samples per rank
    mine = torch.cat([D.strided_slice(sigma, rank, ws, a, 2) for a in range(2)])
    assert torch.equal(mine, ref0[rank::ws])
    # one fused gather of two per-sample columns
    loss = torch.arange(2, dtype=torch.float32) + 10 * rank
    sig = torch.arange(2, dtype=torch.float32) + 100 * rank
    gl, gs = D.gather_scalars([loss, sig])
    assert gl.tolist() == [0.0, 1.0, 10.0, 11.0] and gs.tolist() == [0.0, 1.0, 100.0, 101.0]
    # data-parallel gradient exchange: one flat bucket, SUM over ranks; the optimizer then scales by loss_scale / world_size
    from dualdiffusion_amd.training.train_step import allreduce_gradients
    from dualdiffusion_amd.training.optimizer import LRScheduleConfig, lr_multiplier
    grads = {"w": torch.full((3, 2), float(rank + 1)), "gain": torch.tensor(10.0 * (rank + 1)), "b": torch.arange(4.0) * (rank + 1)}
    red = allreduce_gradients(grads)
    assert torch.equal(red["w"], torch.full((3, 2), 3.0)) and float(red["gain"]) == 30.0 and torch.equal(red["b"], torch.arange(4.0) * 3)
    assert red["w"].shape == (3, 2) and red["gain"].shape == ()
    # two-bucket exchange: the early part travels asynchronously while the tail is still being written
    from dualdiffusion_amd.training.train_step import GradientExchange
    flat = torch.zeros(10)
    flat[:6] = float(rank + 1)
    ex = GradientExchange(flat, 6)
    ex.start_early()
    flat[6:] = torch.arange(4.0) * (rank + 1)          # "encoder" gradients arrive after the early bucket left
    ex.finish()
    assert torch.equal(flat[:6], torch.full((6,), 3.0)) and torch.equal(flat[6:], torch.arange(4.0) * 3)
    ex.finish()                                         # no early part this time: one collective over the whole bucket
    assert torch.equal(flat[:6], torch.full((6,), 6.0)) and torch.equal(flat[6:], torch.arange(4.0) * 6)
    for early in (0, 10):                               # degenerate splits
        f2 = torch.full((10,), float(rank + 1))
        e2 = GradientExchange(f2, early)
        e2.start_early()
        e2.finish()
        assert torch.equal(f2, torch.full((10,), 3.0)), early
    # every rank derives the same learning rate from the global step (host schedule, reference trainer.py:653-663)
    c = LRScheduleConfig()
    assert lr_multiplier(c, 2500) == 0.5 and lr_multiplier(c, 70000) == 1.0 and abs(lr_multiplier(c, 280000) - 0.5) < 1e-12
    D.barrier()
    dist.destroy_process_group()
    open(os.path.join(out_dir, f"ok{rank}"), "w").write("ok")


def test_world_size_2_gloo(tmp_path):
    port = _free_port()
    mp.spawn(_worker, args=(2, port, str(tmp_path)), nprocs=2, join=True)
    assert os.path.isfile(tmp_path / "ok0") and os.path.isfile(tmp_path / "ok1")


def test_single_process_fallbacks():
    from dualdiffusion_amd import distributed as D
    assert D.replica_throughput(7, 0.5) == (7.0, 0.5)
    x = torch.arange(4.0)
    assert torch.equal(D.broadcast_from_rank0(x.clone()), x)
    a, b = D.gather_scalars([x, x + 1])
    assert torch.equal(a, x) and torch.equal(b, x + 1)
