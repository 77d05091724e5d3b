"""One data-parallel UNet training step on MI355X: train batch (forward, loss, backward) -> gradient all-reduce over RCCL ->
dynamic clipping + fused AdamW (+EMA) -> forced weight normalisation.

Mirrors the per-step body of reference src/training/trainer.py:1001-1067 + :375-381 for `UNetTrainer`
(module_trainers/unet_trainer.py:169-296): sigma is drawn for the GLOBAL batch on rank 0, broadcast, and strided per rank
(dualdiffusion_amd.distributed, SigmaSampler); every rank runs its local batch; gradients are summed across ranks in ONE
flat bucket (1.17 GB for the default UNet: a single large all-reduce suits the point-to-point xGMI links better than many
small ones) and averaged through the optimizer's gradient scale.
"""
from __future__ import annotations

from typing import Optional

import torch

from .optimizer import FusedAdamW, LRScheduleConfig, OptimizerConfig, lr_multiplier
from .unet_grad import UNetTrainer


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def allreduce_gradients(grads: dict, names: Optional[list] = None) -> dict:
    """Sum the gradient dict over all ranks through one flat fp32 bucket; returns views into the bucket (same keys).
    A no-op copy when torch.distributed is not initialised (single process)."""
    names = names or sorted(grads)
    flat = torch.cat([grads[k].reshape(-1).float() for k in names])
    if _world_size() > 1:
        import torch.distributed as dist
        dist.all_reduce(flat, op=dist.ReduceOp.SUM)
    out, off = {}, 0
    for k in names:
        n = grads[k].numel()
        out[k] = flat[off:off + n].view(grads[k].shape)
        off += n
    return out


class UNetTrainStep:

    def __init__(self, unet, format, optimizer: OptimizerConfig = OptimizerConfig(), lr_schedule: LRScheduleConfig = LRScheduleConfig(),
                 ema: Optional[dict] = None, ema_beta: float = 0.0, input_perturbation: float = 0.0) -> None:
        self.unet, self.format, self.lr_cfg = unet, format, lr_schedule
        self.trainer = UNetTrainer(unet)
        self.params = {k: p.data for k, p in unet.named_parameters()}
        self.opt = FusedAdamW(self.params, optimizer, ema, ema_beta)
        self.input_perturbation = input_perturbation
        self.global_step = 0

    def step(self, samples: torch.Tensor, audio_embeddings: torch.Tensor, sigma: torch.Tensor, noise: torch.Tensor,
             conditioning_mask: torch.Tensor, perturbation: Optional[torch.Tensor] = None) -> dict:
        """One optimizer step on this rank's batch (the random draws are inputs: the caller owns the generators)."""
        world = _world_size()
        loss, grads = self.trainer.train_batch(samples, audio_embeddings, sigma, noise, conditioning_mask, self.format, perturbation,
                                               self.input_perturbation)
        missing = [k for k in self.params if k not in grads]
        if missing:
            raise RuntimeError(f"UNetTrainStep: no gradient for {missing[:4]}")
        if world > 1:                       # every gradient already lives in the trainer's flat bucket: one all-reduce, no gather
            import torch.distributed as dist
            dist.all_reduce(self.trainer.grad_flat, op=dist.ReduceOp.SUM)
        lr = self.lr_cfg.learning_rate * lr_multiplier(self.lr_cfg, self.global_step)
        grad_norm = self.opt.step(grads, lr, self.opt.cfg.loss_scale / world)
        # trainer.py:375-381: forced weight normalisation after every optimizer step (one launch over the weight bank)
        if self.trainer.bank is not None:
            self.trainer.bank.normalize()
        else:
            self.unet.normalize_weights()
        self.global_step += 1
        return {"loss": loss, "grad_norm": grad_norm, "lr": lr}
