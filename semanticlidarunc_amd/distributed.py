"""Data-parallel glue for one process per GPU (torch.distributed: backend "nccl" == RCCL over xGMI on ROCm,
"gloo" on CPU for tests).  SURVEY section 8(e): scans are sharded across ranks; the only exchange of a training
step is ONE all-reduce of a flat fp32 gradient buffer (26.8 MB for SalsaNext -- far below the size where
bucketing/overlap would pay on 7 x 153 GB/s links), followed by the identical optimizer step on every rank.

The reference's Trainer is not modified: `attach_to_optimizer` installs an optimizer pre-step hook, so its
plain `loss.backward(); optimizer.step()` (src/models/trainer.py:783-786) becomes a synchronous data-parallel
step.  Semantics that differ from a single process are the ones SURVEY lists: BatchNorm uses local batch
statistics (average the running stats with `average_buffers` before checkpointing) and Lovasz is computed per
shard.
"""
from __future__ import annotations

import os
from typing import Iterable, Iterator, List, Optional

import torch
import torch.distributed as dist


def init_from_env(backend: Optional[str] = None, device: Optional[torch.device] = None) -> tuple:
    """(rank, local_rank, world) from the torchrun environment; initialises the default process group when
    WORLD_SIZE > 1.  Rendezvous address defaults to 127.0.0.1 (single node)."""
    rank = int(os.environ.get("RANK", "0"))
    local_rank = int(os.environ.get("LOCAL_RANK", "0"))
    world = int(os.environ.get("WORLD_SIZE", "1"))
    if torch.cuda.is_available() and (backend or "nccl") == "nccl":
        # one process per GPU: make local_rank's card the current device BEFORE anything allocates -- the reference's Trainer
        # uses the bare torch.device("cuda") (trainer.py:229), which is the current device, and every launch of this package
        # goes to the current device's stream
        torch.cuda.set_device(local_rank)
        if device is None:
            device = torch.device("cuda", local_rank)
    if world > 1 and not dist.is_initialized():
        os.environ.setdefault("MASTER_ADDR", "127.0.0.1")
        os.environ.setdefault("MASTER_PORT", "29500")
        if backend is None:
            backend = "nccl" if torch.cuda.is_available() else "gloo"
        kwargs = {"device_id": device} if (device is not None and backend == "nccl") else {}
        dist.init_process_group(backend, rank=rank, world_size=world, **kwargs)
    return rank, local_rank, world


class FlatGradAllReduce:
    """Average the gradients of `params` over the process group with one collective on a flat buffer.
    Parameters whose .grad is None contribute zeros (and receive the average)."""

    def __init__(self, params: Iterable[torch.nn.Parameter], group=None):
        self.params: List[torch.nn.Parameter] = [p for p in params if p.requires_grad]
        if not self.params:
            raise ValueError("no trainable parameters")
        self.group = group
        dev, dt = self.params[0].device, torch.float32
        self.offsets, n = [], 0
        for p in self.params:
            self.offsets.append(n)
            n += p.numel()
        self.flat = torch.zeros(n, dtype=dt, device=dev)
        self._events = []

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    @torch.no_grad()
    def reduce(self) -> None:
        """ONE collective per step: the gradients are packed into the flat buffer by one multi-tensor copy, summed over the ranks by one
        all-reduce, scaled once, and handed back as VIEWS of the flat buffer (no copy back: `optimizer.step()` reads `p.grad` right after this
        hook; the next `zero_grad()` drops or zeroes the views).  `timed = True` brackets the collective with events (`last_allreduce_ms`)."""
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        segs, grads = [], []
        for p, o in zip(self.params, self.offsets):
            seg = self.flat[o:o + p.numel()]
            if p.grad is None:
                seg.zero_()
            elif p.grad.data_ptr() != seg.data_ptr():          # (already a view of the buffer when zero_grad(set_to_none=False) kept it)
                segs.append(seg)
                grads.append(p.grad.reshape(-1))
        if segs:
            torch._foreach_copy_(segs, grads)
        ev = None
        if self.timed and self.flat.is_cuda:
            ev = (torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True))
            ev[0].record()
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        if ev is not None:
            ev[1].record()
            self._events.append(ev)
        self.flat.mul_(1.0 / world)
        for p, o in zip(self.params, self.offsets):
            p.grad = self.flat[o:o + p.numel()].view_as(p)
        self.calls += 1

    timed = False
    calls = 0

    @property
    def last_allreduce_ms(self):
        """Mean duration of the timed all-reduces so far (synchronises), or None."""
        if not getattr(self, "_events", None):
            return None
        torch.cuda.synchronize()
        return sum(a.elapsed_time(b) for a, b in self._events) / len(self._events)

    def attach_to_optimizer(self, optimizer: torch.optim.Optimizer):
        """optimizer.step() now all-reduces first; returns the hook handle."""
        return optimizer.register_step_pre_hook(lambda *_a, **_k: self.reduce())


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t, src=src, group=group)      # on the tensor itself (not .data): the write bumps its version counter


@torch.no_grad()
def average_buffers(module: torch.nn.Module, group=None) -> None:
    """Average floating-point buffers (BatchNorm running statistics) over the ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    for b in module.buffers():
        if b.is_floating_point():
            dist.all_reduce(b, op=dist.ReduceOp.SUM, group=group)     # in place on the buffer itself: version counter bumped, so the
            b.mul_(1.0 / world)                                      # folded eval-BatchNorm caches (salsanext._tkey) are rebuilt


@torch.no_grad()
def all_reduce_metrics(iou=None, ece=None, group=None, device: Optional[torch.device] = None) -> None:
    """Sum the on-device metric accumulators over the ranks: the [C,C] int64 confusion matrix of
    models.evaluator.IoUEvaluator and the evidence of metrics.ece.ECEAggregator (per-bin sums, or its sample buffers).
    EVERY rank enters every collective: a rank that saw no batch contributes zero accumulators allocated on `device`
    (default: the current GPU, or the CPU without one)."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if device is None:
        device = torch.device("cuda", torch.cuda.current_device()) if torch.cuda.is_available() else torch.device("cpu")
    if iou is not None:
        if iou.confmat is None:
            if hasattr(iou, "_ensure"):
                iou._ensure(device)
            else:
                raise RuntimeError("all_reduce_metrics: the IoU accumulator is unallocated on this rank (the other ranks would wait forever)")
        dist.all_reduce(iou.confmat, op=dist.ReduceOp.SUM, group=group)
    if ece is not None:
        if hasattr(ece, "merge_across_ranks"):
            if not getattr(ece, "_keeps_samples", False):
                ece._ensure(device)
            ece.merge_across_ranks(group)
        else:
            for t in (ece._count, ece._sum_correct, ece._sum_conf):
                dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class ShardedSampler(torch.utils.data.Sampler):
    """Disjoint shards of one seeded permutation (the reference shuffles in a single process, src/train_semantics.py:114).
    drop_last=True (training: equal shard sizes keep the gradient average exact) gives every rank
    len(dataset) // world samples; drop_last=False (evaluation) hands the n % world left-over samples to the LAST ranks, one each,
    so every sample is evaluated exactly once and nothing is padded."""

    def __init__(self, n: int, rank: int, world: int, seed: int = 0, shuffle: bool = True, drop_last: bool = True):
        self.n, self.rank, self.world, self.seed, self.shuffle, self.epoch = n, rank, world, seed, shuffle, 0
        self.drop_last = drop_last

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def _bounds(self):
        per, rem = divmod(self.n, self.world)
        if self.drop_last:
            return self.rank * per, (self.rank + 1) * per
        first_big = self.world - rem                       # ranks >= first_big take per + 1
        lo = self.rank * per + max(0, self.rank - first_big)
        return lo, lo + per + (1 if self.rank >= first_big else 0)

    def __len__(self) -> int:
        lo, hi = self._bounds()
        return hi - lo

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            order = list(range(self.n))
        lo, hi = self._bounds()
        return iter(order[lo:hi])
