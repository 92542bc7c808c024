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

    @property
    def nbytes(self) -> int:
        return self.flat.numel() * 4

    @torch.no_grad()
    def reduce(self) -> None:
        world = dist.get_world_size(self.group) if dist.is_initialized() else 1
        if world == 1:
            return
        for p, o in zip(self.params, self.offsets):
            seg = self.flat[o:o + p.numel()]
            if p.grad is None:
                seg.zero_()
            else:
                seg.copy_(p.grad.reshape(-1))
        dist.all_reduce(self.flat, op=dist.ReduceOp.SUM, group=self.group)
        self.flat.mul_(1.0 / world)
        for p, o in zip(self.params, self.offsets):
            g = self.flat[o:o + p.numel()].view_as(p)
            if p.grad is None:
                p.grad = g.clone()
            else:
                p.grad.copy_(g)

    def attach_to_optimizer(self, optimizer: torch.optim.Optimizer):
        """optimizer.step() now all-reduces first; returns the hook handle."""
        return optimizer.register_step_pre_hook(lambda *_a, **_k: self.reduce())


@torch.no_grad()
def broadcast_parameters(module: torch.nn.Module, src: int = 0, group=None) -> None:
    """Make every rank start from rank `src`'s parameters and buffers."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    for t in list(module.parameters()) + list(module.buffers()):
        dist.broadcast(t.data, src=src, group=group)


@torch.no_grad()
def average_buffers(module: torch.nn.Module, group=None) -> None:
    """Average floating-point buffers (BatchNorm running statistics) over the ranks."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    world = dist.get_world_size(group)
    for b in module.buffers():
        if b.is_floating_point():
            dist.all_reduce(b.data, op=dist.ReduceOp.SUM, group=group)
            b.data.mul_(1.0 / world)


@torch.no_grad()
def all_reduce_metrics(iou=None, ece=None, group=None) -> None:
    """Sum the on-device metric accumulators over the ranks: the [C,C] int64 confusion matrix of
    models.evaluator.IoUEvaluator and the per-bin (count, sum_correct, sum_conf) of metrics.ece.ECEAggregator."""
    if not dist.is_initialized() or dist.get_world_size(group) == 1:
        return
    if iou is not None and iou.confmat is not None:
        dist.all_reduce(iou.confmat, op=dist.ReduceOp.SUM, group=group)
    if ece is not None and ece._count is not None:
        for t in (ece._count, ece._sum_correct, ece._sum_conf):
            dist.all_reduce(t, op=dist.ReduceOp.SUM, group=group)


class ShardedSampler(torch.utils.data.Sampler):
    """Disjoint shards of one seeded permutation (the reference shuffles in a single process,
    src/train_semantics.py:114); every rank sees len(dataset) // world samples per epoch."""

    def __init__(self, n: int, rank: int, world: int, seed: int = 0, shuffle: bool = True):
        self.n, self.rank, self.world, self.seed, self.shuffle, self.epoch = n, rank, world, seed, shuffle, 0

    def set_epoch(self, epoch: int) -> None:
        self.epoch = epoch

    def __len__(self) -> int:
        return self.n // self.world

    def __iter__(self) -> Iterator[int]:
        if self.shuffle:
            order = torch.randperm(self.n, generator=torch.Generator().manual_seed(self.seed + self.epoch)).tolist()
        else:
            order = list(range(self.n))
        per = self.n // self.world
        return iter(order[self.rank * per:(self.rank + 1) * per])
