"""A training step replayed as one HIP graph.

The fp32 training path launches ~1 000 small kernels per step (per-layer autograd nodes: conv, BN statistics, affine, their
backward counterparts, layout passes); on MI355X the gaps between them cost ~20 % of the step.  Nothing on the
path synchronises with the host, so the whole step -- forward, loss, backward and (single GPU) the optimizer update -- can be
captured once into a ``torch.cuda.CUDAGraph`` (a hipGraph on ROCm) and replayed on static input buffers.

Data-parallel runs capture forward + loss + backward; the flat RCCL gradient all-reduce (``distributed.FlatGradAllReduce``) and the
optimizer step run after the replay, outside the graph.

    step = GraphedTrainStep(model, optimizer, lambda out, y: salsanext_loss(out, y, 1.0, 1.0, 0)[0], example_x, example_y)
    for x, y in loader:
        loss = step(x, y)            # device scalar; same values as the eager step (dropout masks follow the captured RNG)

Constraints (checked where possible): fixed input shapes; an optimizer created with ``capturable=True`` when the update is inside
the graph (single GPU); no data-dependent Python control flow in model / loss.
"""
from __future__ import annotations

from typing import Callable, Optional, Sequence, Union

import torch

from semanticlidarunc_amd.distributed import FlatGradAllReduce

Tensors = Union[torch.Tensor, Sequence[torch.Tensor]]


def _as_list(x: Tensors):
    return [x] if isinstance(x, torch.Tensor) else list(x)


class GraphedTrainStep:
    def __init__(self, model: torch.nn.Module, optimizer: torch.optim.Optimizer, loss_fn: Callable, example_inputs: Tensors,
                 example_target: torch.Tensor, reducer: Optional[FlatGradAllReduce] = None, warmup: int = 2):
        """`reducer`: pass the FlatGradAllReduce of a data-parallel run (it must NOT be attached to the optimizer as a hook);
        the graph then stops after backward and `__call__` all-reduces and steps eagerly."""
        xs = _as_list(example_inputs)
        if not all(t.is_cuda for t in xs) or not example_target.is_cuda:
            raise RuntimeError("GraphedTrainStep: example tensors must be on the GPU")
        self.model, self.optimizer, self.loss_fn, self.reducer = model, optimizer, loss_fn, reducer
        self.in_graph_update = reducer is None
        if self.in_graph_update and not all(g.get("capturable", True) for g in optimizer.param_groups):      # Adam-family: explicit flag
            raise RuntimeError("GraphedTrainStep: create the optimizer with capturable=True (its update is captured into the graph)")
        self._x = [t.clone() for t in xs]
        self._y = example_target.clone()
        self._single = isinstance(example_inputs, torch.Tensor)
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):                      # warm-up on a side stream: lazy state (packed weights, optimizer moments, allocator)
            for _ in range(max(1, warmup)):
                self._eager()
        torch.cuda.current_stream().wait_stream(side)
        self.optimizer.zero_grad(set_to_none=True)
        self.graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self.graph):
            self._loss = self._fwd_bwd()
            if self.in_graph_update:
                self.optimizer.step()

    def _fwd_bwd(self):
        self.optimizer.zero_grad(set_to_none=True)
        out = self.model(self._x[0]) if self._single else self.model(*self._x)
        loss = self.loss_fn(out, self._y)
        loss.backward()
        return loss.detach()

    def _eager(self):
        loss = self._fwd_bwd()
        if self.reducer is not None:
            self.reducer.reduce()
        self.optimizer.step()
        return loss

    def __call__(self, inputs: Tensors, target: torch.Tensor) -> torch.Tensor:
        xs = _as_list(inputs)
        if len(xs) != len(self._x):
            raise RuntimeError("GraphedTrainStep: number of inputs changed")
        for dst, src in zip(self._x + [self._y], xs + [target]):
            if dst.shape != src.shape or dst.dtype != src.dtype:
                raise RuntimeError(f"GraphedTrainStep: captured {tuple(dst.shape)} {dst.dtype}, got {tuple(src.shape)} {src.dtype}")
            dst.copy_(src, non_blocking=True)
        self.graph.replay()
        if self.reducer is not None:
            self.reducer.reduce()
            self.optimizer.step()
        return self._loss
