"""Sharded parameter pass between reduce-scatter and all-gather (reference src/training/trainer.py:1016-1067 runs
clip_grad_norm_ + AdamW + the EMA manager over every parameter on every rank after accelerate's all-reduce).

Data-parallel ranks hold identical parameters, so after the gradient SUM each rank only needs to update 1 / N of them:

    reduce_scatter(gradient bucket)  ->  |g|^2 of the shard, all-reduced (one float)  ->  clip * AdamW + EMAs on the shard
    ->  all_gather(updated parameters)  ->  forced weight normalisation (whole rows: after the gather, on every rank)

The optimizer traffic of the 293 M-parameter UNet (AdamW + two EMAs: ~32 B per parameter, 4-5 ms per step on one MI355X) and the
moment memory divide by N; the bytes on the wire are those of the all-reduce (reduce-scatter + all-gather of the same bucket).
Layout: parameters live in ONE flat fp32 buffer with the gradient bucket's layout (the module's parameters are re-pointed to views of
it), so the gather is one collective per bucket segment and needs no packing.  Shards are equal slices of a segment, cut anywhere
(AdamW and the EMA lerps are element-wise); the < N * 64 elements of a segment that do not divide evenly are all-reduced and updated
redundantly by every rank.  EMA shadows are only maintained for the rank's own shard (`gather_emas()` completes them for a checkpoint).
"""
from __future__ import annotations

import ctypes as C
import math
from typing import Callable, Optional

import torch

from .. import _lib as L
from .._lib import check, current_stream, lib, ptr
from .optimizer import OptimizerConfig

ALIGN = 64          # shard boundaries are multiples of 64 elements (16-byte accesses, whole 256-byte runs)
ROW = 2048          # job granularity of the HIP pass: one wave per 2048-element run
SQ_JOB = 1 << 20    # elements per job of the gradient-norm launch


class _Dist:
    """torch.distributed, or a world of one (single-GPU check of the same code path)."""

    def __init__(self, group=None) -> None:
        import torch.distributed as dist
        self.dist, self.group = dist, group
        self.on = dist.is_available() and dist.is_initialized()
        self.world = dist.get_world_size(group) if self.on else 1
        self.rank = dist.get_rank(group) if self.on else 0

    def reduce_scatter(self, out: torch.Tensor, inp: torch.Tensor, async_op: bool = False):
        if not self.on or self.world == 1:
            out.copy_(inp)
            return None
        return self.dist.reduce_scatter_tensor(out, inp, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_reduce(self, t: torch.Tensor, async_op: bool = False):
        if not self.on or self.world == 1 or t.numel() == 0:
            return None
        return self.dist.all_reduce(t, op=self.dist.ReduceOp.SUM, group=self.group, async_op=async_op)

    def all_gather(self, out: torch.Tensor, inp: torch.Tensor) -> None:
        if not self.on or self.world == 1:
            out.copy_(inp)
            return
        self.dist.all_gather_into_tensor(out, inp, group=self.group)


class ShardedAdamW:

    def __init__(self, named_params: list, grad_flat: torch.Tensor, grad_views: dict, segments: list, cfg: OptimizerConfig = OptimizerConfig(),
                 emas: Optional[list] = None, normalize: Optional[Callable[[], None]] = None, group=None, use_hip: Optional[bool] = None) -> None:
        """named_params: [(name, torch.nn.Parameter)] -- every one needs a view in `grad_views` (the trainer's flat gradient bucket);
        segments: [(start, numel)] pieces of the bucket that are exchanged separately (the decoder's gradients travel early);
        emas: training.optimizer.EMASpec list; normalize: forced weight normalisation, run after the gather on every rank."""
        self.cfg, self.emas, self.normalize = cfg, list(emas or []), normalize
        self.d = _Dist(group)
        self.grad_flat = grad_flat
        dev = grad_flat.device
        self.use_hip = (dev.type == "cuda") if use_hip is None else use_hip
        if len(self.emas) > L.MAX_EMAS:
            raise L.DDXError(f"ShardedAdamW: at most {L.MAX_EMAS} EMAs")
        base, total = grad_flat.data_ptr(), grad_flat.numel()
        covered = sum(n for _s, n in segments)
        if covered != total or any(s < 0 or n < 0 for s, n in segments):
            raise ValueError("ShardedAdamW: the segments must tile the gradient bucket")
        # ---- flat parameter mirror in the bucket's layout; the module's parameters become views of it
        self.param_flat = torch.zeros_like(grad_flat)
        self.offsets = {}
        for name, p in named_params:
            g = grad_views[name]
            off = (g.data_ptr() - base) // 4
            if p.dtype != torch.float32 or g.numel() != p.numel() or not 0 <= off <= total - p.numel():
                raise L.DDXError(f"ShardedAdamW: {name} must be float32 with a gradient view of its size inside the bucket")
            view = self.param_flat[off:off + p.numel()].view(p.shape)
            view.copy_(p.data)
            p.data = view
            self.offsets[name] = (off, p.numel())
        self.ema_flat = []
        for e in self.emas:            # EMA shadows in the same layout (so that a checkpoint can gather them with one collective per segment)
            ef = torch.zeros_like(grad_flat)
            for name, (off, n) in self.offsets.items():
                t = e.tensors[name]
                ef[off:off + n].copy_(t.reshape(-1))
                e.tensors[name] = ef[off:off + n].view(t.shape)
            self.ema_flat.append(ef)
        # ---- this rank's shard of every segment, and the replicated tails
        W, r = self.d.world, self.d.rank
        self.shards, self.tails = [], []           # (sharded start, shard length, lo, hi) / [(lo, hi), ...] replicated ranges of the segment
        for s, n in segments:
            # shard boundaries are multiples of ALIGN elements of the BUCKET (not of the segment): a segment that starts off the grid (the
            # decoder's parameter count need not be a multiple of 4) would otherwise push every job of the HIP pass onto its unaligned
            # scalar path.  The head up to the grid and what does not divide by world * ALIGN are all-reduced and updated by every rank.
            a0 = min((s + ALIGN - 1) // ALIGN * ALIGN, s + n)
            S = (s + n - a0) // (W * ALIGN) * ALIGN
            self.shards.append((a0, S, a0 + r * S, a0 + (r + 1) * S))
            self.tails.append([(lo, hi) for lo, hi in ((s, a0), (a0 + W * S, s + n)) if hi > lo])
        self.own = [(lo, hi) for (_s, S, lo, hi) in self.shards if S > 0] + [t for ts in self.tails for t in ts]
        n_own = sum(hi - lo for lo, hi in self.own)
        self.g_own = torch.zeros(n_own, dtype=torch.float32, device=dev)       # reduced gradients of the owned ranges, packed
        self.m = torch.zeros(n_own, dtype=torch.float32, device=dev)
        self.v = torch.zeros(n_own, dtype=torch.float32, device=dev)
        self.own_off, o = [], 0
        for lo, hi in self.own:
            self.own_off.append(o)
            o += hi - lo
        self.steps = 0
        self.grad_norm_logmean = float(math.log(cfg.max_grad_norm))
        self.grad_norm_logvar = self.grad_norm_logmean
        self._ws = torch.zeros(3, dtype=torch.float32, device=dev)
        self._pending = []
        self.emas_complete = True      # False between a step() at world > 1 and the next gather_emas(): only this rank's shard is current
        if self.use_hip:
            self._build_tables()

    # ---- dynamic clip threshold (trainer.py:407-431), as FusedAdamW
    def get_max_grad_norm(self) -> float:
        c = self.cfg
        if c.dynamic_max_grad_norm_z is None:
            return c.max_grad_norm
        return math.exp(self.grad_norm_logmean) + math.exp(self.grad_norm_logvar / 2) * c.dynamic_max_grad_norm_z

    def update_grad_norm_stats(self, grad_norm: float, eps: float = 1e-8) -> None:
        c = self.cfg
        grad_norm = max(grad_norm, eps)
        grad_var = max((grad_norm - math.exp(self.grad_norm_logmean)) ** 2, eps)
        self.grad_norm_logmean = self.grad_norm_logmean * c.grad_norm_mean_ema_beta + (1 - c.grad_norm_mean_ema_beta) * math.log(grad_norm)
        self.grad_norm_logvar = self.grad_norm_logvar * c.grad_norm_std_ema_beta + (1 - c.grad_norm_std_ema_beta) * math.log(grad_var)

    # ---- gradient exchange
    def reduce_segment(self, i: int, async_op: bool = False, bucket: Optional[torch.Tensor] = None) -> None:
        """SUM segment i of `bucket` (default: the gradient bucket; with gradient accumulation the accumulated one) over the ranks into
        this rank's packed shard (+ the replicated tail)."""
        bucket = self.grad_flat if bucket is None else bucket
        s, S, lo, hi = self.shards[i]
        W = self.d.world
        works = []
        if S > 0:
            k = self.own.index((lo, hi))
            works.append(self.d.reduce_scatter(self.g_own[self.own_off[k]:self.own_off[k] + S], bucket[s:s + W * S], async_op))
        for tlo, thi in self.tails[i]:
            works.append(self.d.all_reduce(bucket[tlo:thi], async_op))
            self._pending.append(("tail", tlo, thi, bucket))
        self._pending += [w for w in works if w is not None]

    def _wait(self) -> None:
        for w in self._pending:
            if isinstance(w, tuple):
                continue
            w.wait()
        for w in self._pending:
            if isinstance(w, tuple):
                _t, tlo, thi, bucket = w
                k = self.own.index((tlo, thi))
                self.g_own[self.own_off[k]:self.own_off[k] + thi - tlo].copy_(bucket[tlo:thi])
        self._pending = []

    # ---- the local pass
    def _build_tables(self) -> None:
        jobs, jobs_ex, self._max_n, self._max_rows = [], [], 1, 1
        for k, (lo, hi) in enumerate(self.own):
            o = self.own_off[k]
            n = hi - lo
            pieces = [(0, n // ROW * ROW, ROW), (n // ROW * ROW, n, 0)]
            # |g|^2: ddx_multi_grad_norm runs at most 64 workgroups per job, so a shard of up to 293 M elements is cut into jobs of
            # SQ_JOB elements (16-byte aligned cuts) -- hundreds of jobs, like the per-tensor table of FusedAdamW
            for a in range(0, n, SQ_JOB):
                jobs.append((lo + a, o + a, min(SQ_JOB, n - a)))
            for a, b, fan in pieces:
                if b > a:
                    jobs_ex.append((lo + a, o + a, b - a, (b - a) // fan if fan else 1))
        arr = (L.OptimJob * len(jobs))()
        for i, (lo, o, n) in enumerate(jobs):
            arr[i] = L.OptimJob(p=ptr(self.param_flat) + 4 * lo, g=ptr(self.g_own) + 4 * o, m=ptr(self.m) + 4 * o, v=ptr(self.v) + 4 * o, ema=None, n=n)
            self._max_n = max(self._max_n, n)
        ex = (L.OptimJobEx * len(jobs_ex))()
        for i, (lo, o, n, rows) in enumerate(jobs_ex):
            emap = (C.c_void_p * L.MAX_EMAS)(*[(ptr(self.ema_flat[j]) + 4 * lo) if j < len(self.emas) else None for j in range(L.MAX_EMAS)])
            ex[i] = L.OptimJobEx(p=ptr(self.param_flat) + 4 * lo, g=ptr(self.g_own) + 4 * o, m=ptr(self.m) + 4 * o, v=ptr(self.v) + 4 * o, ema=emap,
                                 n=n, rows=rows, normalize=0, reserved=0)
            self._max_rows = max(self._max_rows, rows)
        dev = self.grad_flat.device
        self._table = torch.frombuffer(bytearray(bytes(arr)), dtype=torch.uint8).to(dev)
        self._table_ex = torch.frombuffer(bytearray(bytes(ex)), dtype=torch.uint8).to(dev)
        self._njobs, self._njobs_ex = len(jobs), len(jobs_ex)

    def _sqnorm_local(self) -> None:
        """self._ws[0] = sum g^2 over the owned ranges; the replicated tails count once (on rank 0)."""
        if self.use_hip:
            check(lib().ddx_multi_grad_norm(ptr(self._table), self._njobs, self._max_n, 1.0, 1.0, ptr(self._ws), current_stream()), "multi_grad_norm")
            sq = self._ws[0:1]
        else:
            sq = self._ws[0:1]
            sq.copy_((self.g_own.double() ** 2).sum().float().reshape(1))
        if self.d.rank != 0:
            for k, (lo, hi) in enumerate(self.own):
                if any((lo, hi) in ts for ts in self.tails):
                    o = self.own_off[k]
                    sq -= (self.g_own[o:o + hi - lo] ** 2).sum()

    def step(self, lr: float, grad_scale: Optional[float] = None, ema_betas: Optional[list] = None) -> float:
        """After reduce_segment() of every segment: finish the exchange, run the pass on the shard, gather the parameters, normalise.
        Returns the (scaled) global gradient norm before clipping."""
        c = self.cfg
        gs = c.loss_scale if grad_scale is None else grad_scale
        self._wait()
        ne = len(self.emas)
        if ne and (ema_betas is None or len(ema_betas) != ne):
            raise L.DDXError("ShardedAdamW.step: ema_betas must give one beta per EMA")
        self._sqnorm_local()
        self.d.all_reduce(self._ws[0:1])
        max_norm = self.get_max_grad_norm()
        self.steps += 1
        bias1, bias2 = 1.0 - c.adam_beta1 ** self.steps, 1.0 - c.adam_beta2 ** self.steps
        if self.use_hip:
            check(lib().ddx_clip_coef(ptr(self._ws), gs, max_norm, current_stream()), "clip_coef")
            betas = (C.c_float * max(ne, 1))(*([float(b) for b in ema_betas] if ne else [1.0]))
            fbs = (C.c_float * max(ne, 1))(*([float(e.feedback_beta) if e.feedback_beta is not None else -1.0 for e in self.emas] if ne else [-1.0]))
            check(lib().ddx_multi_adamw_ema_wn(ptr(self._table_ex), self._njobs_ex, self._max_rows, self._ws.data_ptr() + 4, gs, lr, c.adam_beta1,
                                               c.adam_beta2, c.adam_epsilon, c.adam_weight_decay, self.steps, ne, betas, fbs, 1e-4, current_stream()),
                  "multi_adamw_ema_wn(shard)")
            grad_norm = float(self._ws[2])
        else:
            grad_norm = float(self._ws[0].sqrt()) * gs
            coef = min(1.0, max_norm / (grad_norm + 1e-6))
            if math.isfinite(grad_norm):
                for k, (lo, hi) in enumerate(self.own):
                    o, n = self.own_off[k], hi - lo
                    g = self.g_own[o:o + n] * (gs * coef)
                    p, m, v = self.param_flat[lo:hi], self.m[o:o + n], self.v[o:o + n]
                    m.mul_(c.adam_beta1).add_(g, alpha=1 - c.adam_beta1)
                    v.mul_(c.adam_beta2).addcmul_(g, g, value=1 - c.adam_beta2)
                    p.mul_(1 - lr * c.adam_weight_decay)
                    p.addcdiv_(m, v.sqrt() / math.sqrt(bias2) + c.adam_epsilon, value=-lr / bias1)
                    for j, e in enumerate(self.emas):
                        ef = self.ema_flat[j][lo:hi]
                        ef.lerp_(p, 1 - float(ema_betas[j]))
                        if e.feedback_beta is not None:
                            p.lerp_(ef, 1 - float(e.feedback_beta))
        if not math.isfinite(grad_norm):
            self.steps -= 1
            if math.isnan(grad_norm):
                raise FloatingPointError("gradient norm is NaN: optimizer step skipped (reference trainer aborts here, trainer.py:1055-1060)")
            return grad_norm
        # ---- every rank gets the updated parameters of the other shards (the tails were updated by everybody)
        for (s, S, lo, hi) in self.shards:
            if S > 0:
                self.d.all_gather(self.param_flat[s:s + self.d.world * S], self.param_flat[lo:hi].clone())
        if self.normalize is not None:
            self.normalize()
        L.bump_weights_epoch()
        self.update_grad_norm_stats(grad_norm)
        if self.emas and self.d.world > 1:
            self.emas_complete = False
        return grad_norm

    def gather_emas(self) -> None:
        """Complete every EMA shadow on every rank (checkpoint / evaluation time): the other ranks' shards arrive by all-gather.
        COLLECTIVE: every rank must call it (UNetTrainStep.ema_state() does)."""
        for ef in self.ema_flat:
            for (s, S, lo, hi) in self.shards:
                if S > 0:
                    self.d.all_gather(ef[s:s + self.d.world * S], ef[lo:hi].clone())
        self.emas_complete = True

    def ema_tensors(self, j: int) -> dict:
        """name -> shadow tensor of EMA j, complete on this rank.  Raises when only this rank's shard is current: reading `EMASpec.tensors`
        un-gathered at world > 1 would hand out (N - 1) / N stale values (call gather_emas() on every rank first)."""
        if not self.emas_complete:
            raise L.DDXError("ShardedAdamW: the EMA shadows are only current for this rank's shard -- call gather_emas() (collective) "
                             "or UNetTrainStep.ema_state() before reading them")
        return self.emas[j].tensors
