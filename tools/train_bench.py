#!/usr/bin/env python
"""Training-step timing (BASELINE configs[1] / configs[3] shape): B scans of 64x2048x5 per GPU, fp32,
SalsaNext in train mode (batch-statistics BatchNorm, live Dropout2d) -> fused softmax + NLL + Lovasz loss ->
backward (dgrad / wgrad on the fp32 matrix cores) -> [flat RCCL gradient all-reduce] -> AdamW.

    python tools/train_bench.py [--batch 4 --steps 5 --warmup 2]
    python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 tools/train_bench.py
"""
import argparse
import json
import os
import sys
import time

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd.distributed import FlatGradAllReduce, broadcast_parameters, init_from_env  # noqa: E402
from semanticlidarunc_amd.loss import salsanext_loss  # noqa: E402
from semanticlidarunc_amd.salsanext import SalsaNext  # noqa: E402
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan  # noqa: E402


def measure(dev, batch=4, steps=5, warmup=2, height=64, width=2048, precision="fp32", graph=False, rank=0, world=1):
    """Time `steps` training steps on `dev` (after `warmup`): returns the JSON-able result dict.  Used by main() below and by
    bench.py's `train_step` block (BASELINE configs[1]: batch 4 of 64x2048, fp32, one GPU)."""
    from semanticlidarunc_amd import salsanext as sn
    prev = sn._TRAIN_CONV_PRECISION
    sn.set_train_conv_precision(precision)
    try:
        model = seeded_model(SalsaNext).to(dev).train()
        broadcast_parameters(model)
        opt = torch.optim.AdamW(model.parameters(), lr=1e-3, weight_decay=1e-4, capturable=graph)
        red = FlatGradAllReduce(model.parameters())
        red.timed = world > 1
        if not graph:
            red.attach_to_optimizer(opt)
        x, y = synthetic_scan(batch, height, width, seed=1234 + rank)
        x, y = x.to(dev), y.to(dev)
        torch.manual_seed(7 + rank)

        def step():
            opt.zero_grad(set_to_none=True)
            loss, nll, ls = salsanext_loss(model(x), y, 1.0, 1.0, 0)
            loss.backward()
            opt.step()
            return loss

        def sync():
            if world > 1:
                torch.distributed.barrier()
            torch.cuda.synchronize()

        for _ in range(warmup):
            step()
        sync()
        if graph:
            from semanticlidarunc_amd.graph_step import GraphedTrainStep
            graphed = GraphedTrainStep(model, opt, lambda out, t: salsanext_loss(out, t, 1.0, 1.0, 0)[0], x, y,
                                       reducer=red if world > 1 else None)      # multi-GPU: the all-reduce + update follow the replay
            run = lambda: graphed(x, y)
        else:
            run = step
        sync()
        t0 = time.perf_counter()
        for _ in range(steps):
            loss = run()
        sync()
        dt = time.perf_counter() - t0
        if world > 1:
            t = torch.tensor([dt], dtype=torch.float64, device=dev)
            torch.distributed.all_reduce(t, op=torch.distributed.ReduceOp.MAX)
            dt = float(t.item())
        ms = dt / steps * 1e3
        flops = 3 * 124.60e9 * batch * (height * width) / (64 * 2048)
        tf = flops / (ms * 1e-3) / 1e12
        return {"metric": "training scans/s (fwd + NLL/Lovasz loss + bwd + AdamW)", "value": round(batch * world * steps / dt, 3),
                "unit": "scans/s", "n_gpus": world, "ms_per_step": round(ms, 2), "batch_per_gpu": batch, "shape": f"{height}x{width}x5",
                "steps": steps, "warmup": warmup,
                "dtype": "f32" if precision == "fp32" else "f32 storage + accumulate, f16x3 products in the forward convs; exact f32 dgrad / wgrad",
                "conv_tflops_per_gpu(3x fwd flops)": round(tf, 2), "frac_fp32_mfma_peak": round(tf / 157.3, 4), "loss": round(float(loss), 5),
                "grad_allreduce_mb": round(red.nbytes / 1e6, 1),
                "grad_allreduce_ms": None if red.last_allreduce_ms is None else round(red.last_allreduce_ms, 3),
                "grad_allreduce": "one flat fp32 RCCL all-reduce per step from an optimizer pre-step hook (gradients packed by one multi-tensor copy, "
                                  "handed back as views of the flat buffer)" if world > 1 else "single rank: no collective",
                "hip_graph": bool(graph)}
    finally:
        sn.set_train_conv_precision(prev)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--batch", type=int, default=4)
    ap.add_argument("--steps", type=int, default=5)
    ap.add_argument("--warmup", type=int, default=2)
    ap.add_argument("--height", type=int, default=64)
    ap.add_argument("--width", type=int, default=2048)
    ap.add_argument("--graph", action="store_true",
                    help="replay fwd + loss + bwd (+ AdamW on one GPU) as one HIP graph (semanticlidarunc_amd.graph_step; removes the ~1000 launch gaps)")
    ap.add_argument("--precision", default="fp32", choices=["fp32", "f16x3"],
                    help="products of the forward convs (storage and accumulation stay fp32; dgrad / wgrad are always exact fp32)")
    a = ap.parse_args()
    local = int(os.environ.get("LOCAL_RANK", "0"))
    torch.cuda.set_device(local)
    dev = torch.device("cuda", local)
    rank, _, world = init_from_env(device=dev)
    out = measure(dev, a.batch, a.steps, a.warmup, a.height, a.width, a.precision, a.graph, rank, world)
    if rank == 0:
        print(json.dumps(out))
    if world > 1:
        torch.distributed.destroy_process_group()


if __name__ == "__main__":
    main()
