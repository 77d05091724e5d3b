"""One-process-per-GPU helpers (torch.distributed; backend "nccl" is RCCL over xGMI on ROCm, "gloo" in CPU tests).

The inference metric shards by sample with no exchange ("replicas only"): the only
collectives on that path are the barrier and the max-over-ranks timing reduction of the benchmark contract.
The data-parallel training step adds: sigma broadcast from rank 0 (replaces the all_gather-row-0 idiom of reference
src/training/module_trainers/unet_trainer.py:197-198), strided per-rank sigma slices (:246) and one fused small
all_gather of per-sample scalars per micro-step (replaces the separate gathers of :284 and src/training/trainer.py:77).
"""
from __future__ import annotations

import os
from typing import Optional, Sequence

import torch
import torch.distributed as dist


def world() -> tuple[int, int, int]:
    """(rank, world_size, local_rank) from the torchrun environment."""
    return int(os.environ.get("RANK", "0")), int(os.environ.get("WORLD_SIZE", "1")), int(os.environ.get("LOCAL_RANK", "0"))


def init(backend: Optional[str] = None, device: Optional[torch.device] = None) -> bool:
    """Initialise the default process group when WORLD_SIZE > 1.  Returns True when distributed."""
    _, ws, _ = world()
    if ws <= 1:
        return False
    os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
    if not dist.is_initialized():
        if backend is None:
            backend = "nccl" if (device is not None and device.type == "cuda") else "gloo"
        kwargs = {"device_id": device} if backend == "nccl" and device is not None else {}
        dist.init_process_group(backend, **kwargs)
    return True


def _comm_device() -> torch.device:
    return torch.device("cuda", torch.cuda.current_device()) if dist.get_backend() == "nccl" else torch.device("cpu")


def barrier() -> None:
    if dist.is_available() and dist.is_initialized():
        dist.barrier()


def replica_throughput(local_units: float, local_elapsed_s: float) -> tuple[float, float]:
    """Whole-job (units, seconds) of independent replicas: units summed over ranks, time = max over ranks."""
    if not (dist.is_available() and dist.is_initialized()):
        return float(local_units), float(local_elapsed_s)
    t = torch.tensor([local_elapsed_s], dtype=torch.float64, device=_comm_device())
    u = torch.tensor([local_units], dtype=torch.float64, device=_comm_device())
    dist.all_reduce(t, op=dist.ReduceOp.MAX)
    dist.all_reduce(u, op=dist.ReduceOp.SUM)
    return float(u.item()), float(t.item())


def broadcast_from_rank0(x: torch.Tensor) -> torch.Tensor:
    """Every rank ends up with rank 0's values (sigma for the whole global batch)."""
    if dist.is_available() and dist.is_initialized():
        dist.broadcast(x, src=0)
    return x


def strided_slice(global_values: torch.Tensor, rank: int, world_size: int, accum_step: int, device_batch: int) -> torch.Tensor:
    """Per-rank, per-micro-step slice of a global-batch vector: global[rank::world][accum*B:(accum+1)*B]
    (reference unet_trainer.py:246) -- keeps every micro-batch stratified across ranks."""
    return global_values[rank::world_size][accum_step * device_batch:(accum_step + 1) * device_batch]


def gather_scalars(columns: Sequence[torch.Tensor]) -> list[torch.Tensor]:
    """One all_gather for several per-sample vectors of equal length: returns each column concatenated over ranks."""
    stacked = torch.stack([c.detach().float().flatten() for c in columns], dim=0)
    if not (dist.is_available() and dist.is_initialized()):
        return [stacked[i] for i in range(stacked.shape[0])]
    ws = dist.get_world_size()
    out = [torch.empty_like(stacked) for _ in range(ws)]
    dist.all_gather(out, stacked)
    full = torch.cat(out, dim=1)
    return [full[i] for i in range(full.shape[0])]
