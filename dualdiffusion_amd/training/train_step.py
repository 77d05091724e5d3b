"""One data-parallel UNet training step on MI355X: train batch (forward, loss, backward) -> gradient all-reduce over RCCL ->
dynamic clipping + fused AdamW (+EMA) -> forced weight normalisation.

Mirrors the per-step body of reference src/training/trainer.py:1001-1067 + :375-381 for `UNetTrainer`
(module_trainers/unet_trainer.py:169-296): sigma is drawn for the GLOBAL batch on rank 0, broadcast, and strided per rank
(dualdiffusion_amd.distributed, SigmaSampler); every rank runs its local batch; gradients are summed across ranks in ONE
flat bucket (1.17 GB for the default UNet: a single large all-reduce suits the point-to-point xGMI links better than many
small ones) and averaged through the optimizer's gradient scale.
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .optimizer import FusedAdamW, LRScheduleConfig, OptimizerConfig, lr_multiplier
from .unet_grad import UNetTrainer


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _dist_ready() -> bool:
    import torch.distributed as dist
    return dist.is_available() and dist.is_initialized()


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


class GradientExchange:
    """Data-parallel gradient SUM over one flat fp32 bucket in two pieces (reference: accelerate's DDP wrapper buckets the
    gradients and all-reduces each bucket as soon as it is complete: src/training/trainer.py:375 accelerator.prepare, :1016
    accelerator.backward).
    `flat[:early_numel]` holds the gradients that are final first (the decoder's, back-propagated before the encoder):
    start_early() sends them asynchronously on RCCL's stream while the rest of the backward pass runs; finish() sends the tail
    and waits for both.  xGMI rings are per-link bound (~150 GB/s): 0.6 GB per bucket keeps each collective bandwidth-bound,
    not latency-bound."""

    def __init__(self, flat: torch.Tensor, early_numel: int) -> None:
        if not 0 <= early_numel <= flat.numel():
            raise ValueError("GradientExchange: early_numel outside the bucket")
        self.flat, self.early_numel, self._pending = flat, early_numel, None

    def start_early(self) -> None:
        import torch.distributed as dist
        if self._pending is not None:
            raise RuntimeError("GradientExchange: start_early twice in one step")
        if self.early_numel > 0:
            self._pending = dist.all_reduce(self.flat[:self.early_numel], op=dist.ReduceOp.SUM, async_op=True)

    def finish(self) -> None:
        import torch.distributed as dist
        if self._pending is None:            # start_early never ran (graph replay, or nothing early): one collective over everything
            dist.all_reduce(self.flat, op=dist.ReduceOp.SUM)
            return
        if self.early_numel < self.flat.numel():
            dist.all_reduce(self.flat[self.early_numel:], op=dist.ReduceOp.SUM)
        self._pending.wait()
        self._pending = None


class UNetTrainStep:

    def __init__(self, unet, format, optimizer: OptimizerConfig = OptimizerConfig(), lr_schedule: LRScheduleConfig = LRScheduleConfig(),
                 ema: Optional[dict] = None, ema_beta: float = 0.0, input_perturbation: float = 0.0, use_graph: bool = False) -> None:
        """use_graph: capture the whole train batch (forward, loss, backward: ~1900 launches) into one hipGraph on first use and
        replay it afterwards (static input buffers).  The eager loop needs ~20 ms of host time per step and every host hiccup of
        a shared machine lands in the step time; the replay needs the host for the input copies, one graph launch, the
        all-reduce and the four optimizer launches."""
        self.unet, self.format, self.lr_cfg = unet, format, lr_schedule
        self.use_graph = use_graph
        self._graph = None
        self._graph_key = None
        self.trainer = UNetTrainer(unet)
        self.params = {k: p.data for k, p in unet.named_parameters()}
        self.opt = FusedAdamW(self.params, optimizer, ema, ema_beta)
        self.input_perturbation = input_perturbation
        self.global_step = 0

    def _train_batch_graph(self, samples, audio_embeddings, sigma, noise, conditioning_mask, perturbation):
        dev = self.unet.device
        ins = [samples.to(dev, torch.float32), audio_embeddings.to(dev, torch.float32), sigma.flatten().to(dev, torch.float32),
               noise.to(dev, torch.float32), conditioning_mask.to(dev), perturbation.to(dev, torch.float32) if perturbation is not None else None]
        key = tuple((tuple(t.shape), t.dtype) if t is not None else None for t in ins)
        if self._graph is None or key != self._graph_key:
            self._static = [t.clone() if t is not None else None for t in ins]
            st = self._static
            run = lambda: self.trainer.train_batch(st[0], st[1], st[2], st[3], st[4], self.format, st[5], self.input_perturbation)   # noqa: E731
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # eager warm-up on a side stream: weight bank, job tables, kernel attributes
                run()
                run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph = torch.cuda.CUDAGraph()
            with torch.cuda.graph(self._graph):
                self._graph_out = run()
            self._graph_key = key
        for d, t in zip(self._static, ins):
            if d is not None:
                d.copy_(t)
        self._graph.replay()
        return self._graph_out

    def step(self, samples: torch.Tensor, audio_embeddings: torch.Tensor, sigma: torch.Tensor, noise: torch.Tensor,
             conditioning_mask: torch.Tensor, perturbation: Optional[torch.Tensor] = None) -> dict:
        """One optimizer step on this rank's batch (the random draws are inputs: the caller owns the generators)."""
        world = _world_size()
        tr = self.trainer
        # DDX_DDP_BUCKETS=1 runs the two-bucket exchange on any initialised process group (world_size 1 included: single-GPU check)
        exchange = world > 1 or (os.environ.get("DDX_DDP_BUCKETS", "0") == "1" and _dist_ready())
        ex = GradientExchange(tr.grad_flat, tr.early_numel) if exchange else None
        # eager: the decoder's bucket travels while the encoder is back-propagated; graph replay: one collective after the replay
        tr.bucket_hook = ex.start_early if (ex is not None and not self.use_graph) else None
        if self.use_graph:
            loss, grads = self._train_batch_graph(samples, audio_embeddings, sigma, noise, conditioning_mask, perturbation)
        else:
            loss, grads = self.trainer.train_batch(samples, audio_embeddings, sigma, noise, conditioning_mask, self.format, perturbation,
                                                   self.input_perturbation)
        missing = [k for k in self.params if k not in grads]
        if missing:
            raise RuntimeError(f"UNetTrainStep: no gradient for {missing[:4]}")
        if ex is not None:                  # every gradient already lives in the trainer's flat bucket: no gather copies
            ex.finish()
        lr = self.lr_cfg.learning_rate * lr_multiplier(self.lr_cfg, self.global_step)
        grad_norm = self.opt.step(grads, lr, self.opt.cfg.loss_scale / world)
        # trainer.py:375-381: forced weight normalisation after every optimizer step (one launch over the weight bank)
        if self.trainer.bank is not None:
            self.trainer.bank.normalize()
        else:
            self.unet.normalize_weights()
        self.global_step += 1
        return {"loss": loss, "grad_norm": grad_norm, "lr": lr}
