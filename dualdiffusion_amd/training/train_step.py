"""One data-parallel UNet optimizer step on MI355X: [gradient-accumulation micro-steps of (train batch: forward, loss, backward)] ->
gradient all-reduce over RCCL -> dynamic clipping + fused AdamW + EMAs (+ feedback) + forced weight normalisation.

Mirrors the per-step body of reference src/training/trainer.py:1001-1108 for `UNetTrainer`
(module_trainers/unet_trainer.py:169-296):
  * `init_batch` (:169-200): sigma is drawn ONCE for the GLOBAL batch (device_batch x accumulation steps x ranks) on rank 0 and
    broadcast (dualdiffusion_amd.distributed.broadcast_from_rank0 replaces the all_gather-row-0 idiom of :197-198);
  * every micro-step takes the strided slice global_sigma[rank::world][accum * B:(accum + 1) * B] (:245-246), draws the
    conditioning mask / noise / input perturbation, runs the train batch on the HIP kernels; gradients ACCUMULATE over the
    micro-steps locally (accelerate's `accumulate` / no_sync, trainer.py:1012-1016) and the per-sample scalars travel in ONE small
    all_gather per micro-step (distributed.gather_scalars replaces the separate gathers of unet_trainer.py:284 / trainer.py:77);
  * only the LAST micro-step exchanges gradients: ONE flat fp32 bucket (1.17 GB for the default UNet) in two collectives, the
    decoder's part overlapped with the encoder's backward (GradientExchange);
  * the optimizer's gradient scale loss_scale / (world x accumulation steps) does the averaging (accelerate divides the loss by
    the accumulation steps, DDP averages over ranks).
`step(...)` is the single-micro-step entry with the random draws given by the caller (tests, benchmarks).
"""
from __future__ import annotations

import os
from typing import Optional

import torch

from .. import distributed as D
from .optimizer import EMASpec, FusedAdamW, LRScheduleConfig, OptimizerConfig, lr_multiplier


def _world_size() -> int:
    import torch.distributed as dist
    return dist.get_world_size() if dist.is_available() and dist.is_initialized() else 1


def _rank() -> int:
    import torch.distributed as dist
    return dist.get_rank() if dist.is_available() and dist.is_initialized() else 0


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


def trainer_options(cfg: dict) -> dict:
    """UNetTrainStep keyword arguments for the reference `module_trainer_config` options (unet_trainer.py:38-72) that change the objective:
    input_perturbation, conditioning_perturbation, conditioning_dropout, normalize_latents, use_dynamic_sigma_data (+ min / max / exp)."""
    check_trainer_config(cfg)
    kw = dict(input_perturbation=float(cfg.get("input_perturbation", 0.0) or 0.0), conditioning_dropout=float(cfg.get("conditioning_dropout", 0.1)),
              conditioning_perturbation=float(cfg.get("conditioning_perturbation", 0.0) or 0.0), normalize_latents=bool(cfg.get("normalize_latents", False)))
    if cfg.get("use_dynamic_sigma_data"):
        kw["dynamic_sigma_data"] = (float(cfg.get("dynamic_sigma_data_min", 0.2)), float(cfg.get("dynamic_sigma_data_max", 5.0)),
                                    float(cfg.get("dynamic_sigma_data_exp", 1.0)))
    return kw


def check_trainer_config(cfg: dict) -> None:
    """Reject reference `module_trainer_config` options that the HIP train batch does not implement, instead of silently training a
    different objective.  Every option of the live trainer (unet_trainer.py:38-72) is implemented (trainer_options); what is left is
    `inpainting_probability`, an option of the retired `module_trainers/old/unet_trainer_b4.py:73-80` (that module no longer imports in
    the reference: its `.module_trainer` sibling is gone) -- the live trainer takes the reference samples from its caller
    (`ref_samples`, unet_trainer.py:223,260), and so does UNetTrainStep.step / run_batch."""
    bad = [k for k in ("inpainting_probability",) if cfg.get(k) not in (None, 0, 0.0, False)]
    if bad:
        raise NotImplementedError(f"UNetTrainStep: trainer options not implemented on the HIP path: {bad}")


class GradientExchange:
    """Data-parallel gradient SUM over one flat fp32 bucket in two pieces (reference: accelerate's DDP wrapper buckets the
    gradients and all-reduces each bucket as soon as it is complete: src/training/trainer.py:375 accelerator.prepare, :1016
    accelerator.backward).
    `flat[:early_numel]` holds the gradients that are final first (the decoder's, back-propagated before the encoder):
    start_early() sends them asynchronously on RCCL's stream while the rest of the backward pass runs; finish() sends the tail
    and waits for both.  xGMI rings are per-link bound (~150 GB/s): 0.6 GB per bucket keeps each collective bandwidth-bound,
    not latency-bound.
    With `accum` (gradient accumulation) the bucket that travels is `accum`, and the last micro-step's gradients `flat` are added
    into it piecewise right before each piece is sent."""

    def __init__(self, flat: torch.Tensor, early_numel: int, accum: Optional[torch.Tensor] = None, mode: Optional[str] = None,
                 comm_dtype: Optional[torch.dtype] = None) -> None:
        """mode: "all_reduce" (default; RCCL picks the algorithm) or "rs_ag": every piece as an explicit reduce_scatter_tensor +
        all_gather_into_tensor pair over equal shards (2 x (N - 1) / N of the bytes cross each GPU's links, spread over all
        peers, instead of whatever ring RCCL builds; the shard between the two collectives is where a sharded optimizer pass would
        run); the < world_size tail elements of a piece that do not divide travel in a small all_reduce.  DDX_GRAD_EXCHANGE=rs_ag
        selects it.  comm_dtype=torch.bfloat16 (DDX_GRAD_COMM=bf16): the bucket travels as bf16 (0.59 GB instead of 1.17 GB for the
        default UNet; the SUM is then rounded to bf16 -- an option, not the default)."""
        if not 0 <= early_numel <= flat.numel():
            raise ValueError("GradientExchange: early_numel outside the bucket")
        if accum is not None and accum.shape != flat.shape:
            raise ValueError("GradientExchange: accumulation bucket must have the gradient bucket's shape")
        self.flat, self.early_numel, self.accum, self._pending = flat, early_numel, accum, None
        self.bucket = accum if accum is not None else flat
        self.mode = mode or os.environ.get("DDX_GRAD_EXCHANGE", "all_reduce")
        if self.mode not in ("all_reduce", "rs_ag"):
            raise ValueError(f"GradientExchange: unknown mode {self.mode!r}")
        if comm_dtype is None and os.environ.get("DDX_GRAD_COMM", "") == "bf16":
            comm_dtype = torch.bfloat16
        self.comm_dtype = comm_dtype

    def _send(self, piece: torch.Tensor, async_op: bool):
        """SUM `piece` (a contiguous slice of the bucket) over the ranks; returns a callable that completes it (waits, second
        collective of the rs_ag pair, cast back)."""
        import torch.distributed as dist
        buf = piece.to(self.comm_dtype) if self.comm_dtype is not None else piece
        done = (lambda: piece.copy_(buf)) if buf is not piece else (lambda: None)
        if self.mode == "all_reduce":
            w = dist.all_reduce(buf, op=dist.ReduceOp.SUM, async_op=async_op)
            return (lambda: (w.wait(), done())) if async_op else done
        world = dist.get_world_size()
        m = buf.numel() // world * world
        shard = torch.empty(m // world, dtype=buf.dtype, device=buf.device)
        w1 = dist.reduce_scatter_tensor(shard, buf[:m], op=dist.ReduceOp.SUM, async_op=async_op) if m else None
        w2 = dist.all_reduce(buf[m:], op=dist.ReduceOp.SUM, async_op=async_op) if m < buf.numel() else None

        def complete():
            if async_op:
                for w in (w1, w2):
                    if w is not None:
                        w.wait()
            if m:
                dist.all_gather_into_tensor(buf[:m], shard)
            done()
        return complete

    def start_early(self) -> None:
        if self._pending is not None:
            raise RuntimeError("GradientExchange: start_early twice in one step")
        e = self.early_numel
        if e > 0:
            if self.accum is not None:
                self.accum[:e].add_(self.flat[:e])
            self._pending = self._send(self.bucket[:e], async_op=True)

    def finish(self) -> None:
        e = self.early_numel
        if self._pending is None:            # start_early never ran (graph replay, or nothing early): one exchange over everything
            if self.accum is not None:
                self.accum.add_(self.flat)
            self._send(self.bucket, async_op=False)()
            return
        if e < self.flat.numel():
            if self.accum is not None:
                self.accum[e:].add_(self.flat[e:])
            self._send(self.bucket[e:], async_op=False)()
        self._pending()
        self._pending = None


class ShardedExchange:
    """GradientExchange's interface over training.sharded.ShardedAdamW: the two bucket segments are reduce-scattered (the decoder's early,
    asynchronously), the parameter pass then runs on this rank's shard and the updated parameters are all-gathered (ShardedAdamW.step)."""

    def __init__(self, opt, flat: torch.Tensor, early_numel: int, accum: Optional[torch.Tensor] = None) -> None:
        self.opt, self.flat, self.early_numel, self.accum = opt, flat, early_numel, accum
        self.bucket = accum if accum is not None else flat
        self._early_sent = False

    def _send(self, i: int, async_op: bool) -> None:
        s, n = (0, self.early_numel) if i == 0 else (self.early_numel, self.flat.numel() - self.early_numel)
        if n == 0:
            return
        if self.accum is not None:
            self.accum[s:s + n].add_(self.flat[s:s + n])
        self.opt.reduce_segment(i, async_op=async_op, bucket=self.bucket)

    def start_early(self) -> None:
        if self._early_sent:
            raise RuntimeError("ShardedExchange: start_early twice in one step")
        self._send(0, True)
        self._early_sent = True

    def finish(self) -> None:
        if not self._early_sent:
            self._send(0, False)
        self._send(1, False)
        self._early_sent = False


class UNetTrainStep:

    def __init__(self, unet, format, optimizer: OptimizerConfig = OptimizerConfig(), lr_schedule: LRScheduleConfig = LRScheduleConfig(),
                 ema: Optional[dict] = None, ema_beta: float = 0.0, input_perturbation: float = 0.0, use_graph: bool = False,
                 gradient_accumulation_steps: int = 1, sigma_sampler=None, conditioning_dropout: float = 0.1,
                 emas: Optional[list] = None, fused_weight_norm: bool = False, trainer=None, optimizer_impl=None,
                 grad_exchange: Optional[str] = None, conditioning_perturbation: float = 0.0, normalize_latents: bool = False,
                 dynamic_sigma_data: Optional[tuple] = None) -> None:
        """use_graph: capture the whole train batch (forward, loss, backward: ~1900 launches) into one hipGraph on first use and
        replay it afterwards (static input buffers).  The eager loop needs ~20 ms of host time per step and every host hiccup of
        a shared machine lands in the step time; the replay needs the host for the input copies, one graph launch, the
        all-reduce and the optimizer launches.
        emas: list of training.optimizer.EMASpec (power-function / classic / feedback EMAs, reference ema.py); with emas or
        fused_weight_norm the parameter pass after the backward is ONE launch (AdamW + EMAs + feedback + forced weight norm).
        trainer / optimizer_impl: differentiation engine and parameter pass (defaults: the module's own UNetTrainer and
        FusedAdamW on the HIP kernels; the world_size-2 CPU test passes stubs to drive this class's control flow over gloo).
        conditioning_perturbation / normalize_latents / dynamic_sigma_data = (min, max, exp): the reference trainer options of the same
        names (unet_trainer.py:65-72; `trainer_options()` maps a config dict); config.dropout > 0 of the module draws one seed per micro-step."""
        self.unet, self.format, self.lr_cfg = unet, format, lr_schedule
        self.use_graph = use_graph
        self._graph = None
        self._graph_key = None
        self.trainer = trainer if trainer is not None else unet._get_trainer()
        self.params = {k: p.data for k, p in unet.named_parameters()}
        wn_rows = None
        if fused_weight_norm or emas:
            wn_rows = {k + ".weight": m.weight.shape[0] for k, m in unet.named_modules()
                       if hasattr(m, "disable_weight_norm") and hasattr(m, "weight") and not m.disable_weight_norm}
        self.fused_weight_norm = wn_rows is not None
        # grad_exchange = "sharded" (DDX_GRAD_EXCHANGE=sharded): ZeRO-1 style -- reduce-scatter, parameter pass on this rank's shard of one
        # flat parameter buffer, all-gather, weight norm (training.sharded); also at world size 1, where it is the same arithmetic
        self.grad_exchange = grad_exchange or os.environ.get("DDX_GRAD_EXCHANGE", "all_reduce")
        if self.grad_exchange not in ("all_reduce", "rs_ag", "sharded"):
            raise ValueError(f"UNetTrainStep: grad_exchange must be all_reduce | rs_ag | sharded, not {self.grad_exchange!r}")
        self.sharded = None
        if self.grad_exchange == "sharded" and optimizer_impl is None:
            from .sharded import ShardedAdamW
            tr = self.trainer
            if ema is not None:
                raise ValueError("UNetTrainStep: the sharded pass takes `emas` (EMASpec list), not the single fixed-beta `ema`")
            total = tr.grad_flat.numel()

            def _normalize():
                if getattr(tr, "bank", None) is not None:
                    tr.bank.normalize()
                else:
                    self.unet.normalize_weights()
            self.sharded = ShardedAdamW(list(unet.named_parameters()), tr.grad_flat, tr.grad_views, [(0, tr.early_numel), (tr.early_numel, total - tr.early_numel)],
                                        optimizer, emas=emas, normalize=_normalize)
            self.params = {k: p.data for k, p in unet.named_parameters()}     # (now views of the flat parameter buffer)
            if getattr(tr, "bank", None) is not None:                          # its job table holds the old storage
                tr.bank, tr._bank_key = None, None
            self.fused_weight_norm = True                                     # (the normalisation runs inside ShardedAdamW.step)
        self.opt = optimizer_impl if optimizer_impl is not None else (self.sharded if self.sharded is not None else
                                                                      FusedAdamW(self.params, optimizer, ema, ema_beta, emas=emas, wn_rows=wn_rows))
        self.emas = list(emas or [])
        self.input_perturbation = input_perturbation
        self.conditioning_perturbation = float(conditioning_perturbation)
        self.normalize_latents, self.dynamic_sigma_data = bool(normalize_latents), dynamic_sigma_data
        self.dropout = float(getattr(getattr(unet, "config", None), "dropout", 0.0) or 0.0)
        if use_graph and (self.dropout > 0):
            raise NotImplementedError("UNetTrainStep(use_graph=True) with config.dropout > 0: the dropout seed is a launch argument, a captured "
                                      "batch would replay one mask (run eagerly)")
        self.accum_steps = int(gradient_accumulation_steps)
        self.sigma_sampler, self.conditioning_dropout = sigma_sampler, conditioning_dropout
        self.global_step = 0
        self.total_samples_processed = 0
        self._accum: Optional[torch.Tensor] = None
        self.last_gathered: list = []        # per micro-step: [loss, sigma] of the GLOBAL micro-batch (one all_gather each)

    def _train_batch_graph(self, samples, audio_embeddings, sigma, noise, conditioning_mask, perturbation, hook=None, split=False):
        """Replay the captured train batch.  With `split` (a gradient exchange exists) the batch is captured as TWO graphs, cut where the
        trainer calls its bucket hook -- [forward, loss, decoder backward] | [encoder backward] -- and `hook` (the asynchronous
        all-reduce of the decoder's gradient bucket; None on the micro-steps of an accumulation that do not communicate) runs between
        the two replays, so the early collective travels on RCCL's stream while the second graph runs (a capture cannot contain the
        collective itself)."""
        dev = self.unet.device
        ins = [samples.to(dev, torch.float32), audio_embeddings.to(dev, torch.float32), sigma.flatten().to(dev, torch.float32),
               noise.to(dev, torch.float32), conditioning_mask.to(dev), perturbation.to(dev, torch.float32) if perturbation is not None else None]
        key = (bool(split),) + tuple((tuple(t.shape), t.dtype) if t is not None else None for t in ins)
        tr = self.trainer
        if self._graph is None or key != self._graph_key:
            self._static = [t.clone() if t is not None else None for t in ins]
            st = self._static
            run = lambda: tr.train_batch(st[0], st[1], st[2], st[3], st[4], self.format, st[5], self.input_perturbation)   # noqa: E731
            tr.bucket_hook = None
            side = torch.cuda.Stream(device=dev)
            side.wait_stream(torch.cuda.current_stream())
            with torch.cuda.stream(side):          # eager warm-up on a side stream: weight bank, job tables, kernel attributes
                run()
                run()
            torch.cuda.current_stream().wait_stream(side)
            torch.cuda.synchronize()
            self._graph, self._graph_tail = torch.cuda.CUDAGraph(), None
            if not split:
                with torch.cuda.graph(self._graph):
                    self._graph_out = run()
            else:
                tail = torch.cuda.CUDAGraph()
                split = {"done": False}

                def split_here():            # called by the trainer once the early part of the gradient bucket is final
                    self._graph.capture_end()
                    tail.capture_begin(pool=self._graph.pool())
                    split["done"] = True

                tr.bucket_hook = split_here
                with torch.cuda.stream(side):
                    self._graph.capture_begin()
                    try:
                        self._graph_out = run()
                    finally:
                        (tail if split["done"] else self._graph).capture_end()
                        tr.bucket_hook = None
                torch.cuda.current_stream().wait_stream(side)
                self._graph_tail = tail if split["done"] else None
            self._graph_key = key
        for d, t in zip(self._static, ins):
            if d is not None:
                d.copy_(t)
        self._graph.replay()
        if self._graph_tail is not None:
            if hook is not None:
                hook()
            self._graph_tail.replay()
        # (no split point -- the trainer never called the hook, e.g. an empty early bucket: GradientExchange.finish sends everything)
        return self._graph_out

    # ------------------------------------------------------------------------------------------------ micro-steps
    def _extras(self) -> bool:
        return self.conditioning_perturbation > 0 or self.normalize_latents or self.dynamic_sigma_data is not None or self.dropout > 0

    def _micro(self, samples, audio_embeddings, sigma, noise, conditioning_mask, perturbation, hook, split=False, cond_perturbation=None,
               dropout_seed=None):
        tr = self.trainer
        tr.bucket_hook = hook
        if self.use_graph:
            if self._extras():
                raise NotImplementedError("UNetTrainStep(use_graph=True): conditioning_perturbation / normalize_latents / dynamic_sigma_data / dropout "
                                          "run on the eager path")
            return self._train_batch_graph(samples, audio_embeddings, sigma, noise, conditioning_mask, perturbation, hook, split or hook is not None)
        if not self._extras():
            return tr.train_batch(samples, audio_embeddings, sigma, noise, conditioning_mask, self.format, perturbation, self.input_perturbation)
        return tr.train_batch(samples, audio_embeddings, sigma, noise, conditioning_mask, self.format, perturbation, self.input_perturbation,
                              conditioning_perturbation=cond_perturbation, conditioning_perturbation_scale=self.conditioning_perturbation,
                              normalize_latents=self.normalize_latents, dynamic_sigma_data=self.dynamic_sigma_data, dropout_seed=dropout_seed)

    def _finish(self, loss, grads, world: int, n_micro: int, ex: Optional[GradientExchange], device_batch: int) -> dict:
        tr = self.trainer
        missing = [k for k in self.params if k not in grads]
        if missing:
            raise RuntimeError(f"UNetTrainStep: no gradient for {missing[:4]}")
        if ex is not None:                  # every gradient already lives in the trainer's flat bucket: no gather copies
            ex.finish()
        elif self._accum is not None and n_micro > 1:
            self._accum.add_(tr.grad_flat)
        if n_micro > 1:                     # the optimizer reads the accumulated bucket (same layout as the trainer's)
            base = tr.grad_flat.data_ptr()
            grads = {k: self._accum[(v.data_ptr() - base) // 4:(v.data_ptr() - base) // 4 + v.numel()].view(v.shape) for k, v in grads.items()}
        lr = self.lr_cfg.learning_rate * lr_multiplier(self.lr_cfg, self.global_step)
        total_batch = device_batch * n_micro * world
        betas = [e.effective_beta(self.global_step, self.total_samples_processed, total_batch) for e in self.emas] if self.emas else None
        if self.sharded is not None:
            grad_norm = self.sharded.step(lr, self.sharded.cfg.loss_scale / (world * n_micro), ema_betas=betas)
        else:
            grad_norm = self.opt.step(grads, lr, self.opt.cfg.loss_scale / (world * n_micro), ema_betas=betas)
        # trainer.py:1105-1108: forced weight normalisation after every optimizer step (inside the fused launch, or one launch over
        # the weight bank)
        if not self.fused_weight_norm:
            if getattr(tr, "bank", None) is not None:
                tr.bank.normalize()
            else:
                self.unet.normalize_weights()
        self.global_step += 1
        self.total_samples_processed += total_batch
        return {"loss": loss, "grad_norm": grad_norm, "lr": lr}

    def ema_state(self) -> dict:
        """{EMA name: {parameter name: shadow tensor}}, complete on every rank -- THE way to read the EMA weights (checkpoint, evaluation,
        reference ema.py:230-321 `EMA_Manager.save`).  With the sharded parameter pass a rank only keeps its own shard of every shadow
        current between steps; this call completes them (one all-gather per EMA and bucket segment: COLLECTIVE, every rank must call it,
        like the reference's checkpoint barrier, trainer.py:1122-1130).  Other modes: the shadows are already whole."""
        if self.sharded is not None:
            if not self.sharded.emas_complete:
                self.sharded.gather_emas()
            return {e.name: self.sharded.ema_tensors(j) for j, e in enumerate(self.emas)}
        return {e.name: e.tensors for e in self.emas}

    def prepare_checkpoint(self) -> dict:
        """What a checkpoint of this rank holds: the model parameters and the complete EMA shadows (collective in sharded mode)."""
        return {"params": {k: p.data for k, p in self.unet.named_parameters()}, "emas": self.ema_state(),
                "global_step": self.global_step, "total_samples_processed": self.total_samples_processed}

    def _exchange(self, world: int, n_micro: int) -> Optional[GradientExchange]:
        tr = self.trainer
        # DDX_DDP_BUCKETS=1 runs the two-bucket exchange on any initialised process group (world_size 1 included: single-GPU check)
        exchange = world > 1 or (os.environ.get("DDX_DDP_BUCKETS", "0") == "1" and _dist_ready())
        if n_micro > 1 and (self._accum is None or self._accum.shape != tr.grad_flat.shape):
            self._accum = torch.zeros_like(tr.grad_flat)
        if self.sharded is not None:
            return ShardedExchange(self.sharded, tr.grad_flat, tr.early_numel, self._accum if n_micro > 1 else None)
        return GradientExchange(tr.grad_flat, tr.early_numel, self._accum if n_micro > 1 else None, mode=self.grad_exchange) if exchange else None

    def step(self, samples: torch.Tensor, audio_embeddings: torch.Tensor, sigma: torch.Tensor, noise: torch.Tensor,
             conditioning_mask: torch.Tensor, perturbation: Optional[torch.Tensor] = None, cond_perturbation: Optional[torch.Tensor] = None,
             dropout_seed: Optional[int] = None) -> dict:
        """One optimizer step on this rank's batch, ONE micro-step (the random draws are inputs: the caller owns the generators)."""
        world = _world_size()
        ex = self._exchange(world, 1)
        # the decoder's bucket travels while the encoder is back-propagated (graph mode: between the two captured halves)
        hook = ex.start_early if ex is not None else None
        loss, grads = self._micro(samples, audio_embeddings, sigma, noise, conditioning_mask, perturbation, hook, cond_perturbation=cond_perturbation,
                                  dropout_seed=dropout_seed)
        return self._finish(loss, grads, world, 1, ex, int(samples.shape[0]))

    def run_batch(self, samples: torch.Tensor, audio_embeddings: torch.Tensor, generator: Optional[torch.Generator] = None,
                  sigma_jitter: Optional[torch.Tensor] = None) -> dict:
        """One optimizer step as the reference's loop body (trainer.py:1001-1108) on this rank's LOCAL batch
        `samples` [device_batch * accumulation steps, C, H, W] / `audio_embeddings`: sigma for the global batch from the sigma
        sampler on rank 0, micro-steps with local gradient accumulation, exchange after the last one, one parameter pass."""
        if self.sigma_sampler is None:
            raise RuntimeError("UNetTrainStep.run_batch needs a sigma_sampler")
        world, rank, A = _world_size(), _rank(), self.accum_steps
        if samples.shape[0] % A:
            raise ValueError(f"local batch of {samples.shape[0]} is not divisible by {A} accumulation steps")
        Bd = samples.shape[0] // A
        dev = self.unet.device
        # init_batch (unet_trainer.py:169-200): whole-batch sigma, identical on every rank
        gsig = self.sigma_sampler.sample(Bd * A * world, jitter=sigma_jitter).float()
        gsig = gsig.to(dev) if _dist_ready() and D.dist.get_backend() == "nccl" else gsig
        gsig = D.broadcast_from_rank0(gsig.contiguous())
        self.global_sigma = gsig
        ex = self._exchange(world, A)
        if self._accum is not None and A > 1:
            self._accum.zero_()
        self.last_gathered = []
        losses = []
        for a in range(A):
            last = a == A - 1
            x = samples[a * Bd:(a + 1) * Bd]
            e = audio_embeddings[a * Bd:(a + 1) * Bd]
            sig = D.strided_slice(gsig, rank, world, a, Bd).to(dev)
            # the reference's draws in the reference's order (unet_trainer.py:234-257): mask, conditioning perturbation, noise, input perturbation
            mask = torch.rand(Bd, generator=generator, device=generator.device if generator is not None else dev) > self.conditioning_dropout
            cpert = (torch.randn((Bd, self.unet.cemb), generator=generator, device=mask.device) if self.conditioning_perturbation > 0 else None)
            noise = torch.randn(x.shape, generator=generator, device=mask.device)
            pert = torch.randn(x.shape, generator=generator, device=mask.device) if self.input_perturbation > 0 else None
            dseed = int(torch.randint(0, 2 ** 62, (1,), generator=generator, device=mask.device).item()) if self.dropout > 0 else None
            hook = ex.start_early if (ex is not None and last) else None
            loss, grads = self._micro(x, e, sig, noise, mask, pert, hook, split=ex is not None, cond_perturbation=cpert, dropout_seed=dseed)
            if not last:                      # no_sync micro-step: gradients stay local
                self._accum.add_(self.trainer.grad_flat)
            self.last_gathered.append(D.gather_scalars([loss, sig]))       # one small collective per micro-step
            # (graph mode returns the captured static loss buffer, which the next replay overwrites: keep a copy per micro-step)
            losses.append(loss.clone() if (self.use_graph and A > 1) else loss)
        return self._finish(torch.cat(losses) if A > 1 else losses[0], grads, world, A, ex, Bd)
