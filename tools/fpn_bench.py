#!/usr/bin/env python
"""Single-pass FPN inference timing at the reference's own self-benchmark shape (src/models/semanticFCN.py:381-395:
1 x (2 + 6) x 128 x 2048, median of CUDA-event timings).  python tools/fpn_bench.py [--backbone resnet18] [--batch 1]"""
import argparse
import json
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import salsanext as sn  # noqa: E402
from semanticlidarunc_amd.fpn import SemanticNetworkWithFPN  # noqa: E402
from semanticlidarunc_amd.testing import randomize_bn_  # noqa: E402

ap = argparse.ArgumentParser()
ap.add_argument("--backbone", default="resnet18")
ap.add_argument("--batch", type=int, default=1)
ap.add_argument("--iters", type=int, default=100)
a = ap.parse_args()
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = randomize_bn_(SemanticNetworkWithFPN(a.backbone, 2, 6, num_classes=20), 3).eval().to(dev)
x, meta = torch.randn(a.batch, 2, 128, 2048, device=dev), torch.randn(a.batch, 6, 128, 2048, device=dev)
out = {}
for prec in ("fp32", "f16x3"):
    sn.set_conv_precision(prec)
    with torch.no_grad():
        for _ in range(10):
            model(x, meta)
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            model(x, meta)
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    out[prec] = {"median_ms": round(statistics.median(ts), 3), "scans_per_s": round(a.batch * 1e3 / statistics.median(ts), 1)}
    # the same forward replayed from a HIP graph (hipGraph via torch.cuda.CUDAGraph): removes the per-launch host cost
    with torch.no_grad():
        static_x, static_m = x.clone(), meta.clone()
        s_ = torch.cuda.Stream()
        s_.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(s_):
            for _ in range(3):
                model(static_x, static_m)
        torch.cuda.current_stream().wait_stream(s_)
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph):
            y_static = model(static_x, static_m)
        eager = model(static_x, static_m)
        graph.replay()
        torch.cuda.synchronize()
        assert torch.equal(eager, y_static), "graph replay differs from the eager forward"
        ts = []
        for _ in range(a.iters):
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            graph.replay()
            e1.record()
            torch.cuda.synchronize()
            ts.append(e0.elapsed_time(e1))
    out[prec]["hipgraph_median_ms"] = round(statistics.median(ts), 3)
print(json.dumps({"model": f"FPN/{a.backbone}", "shape": [a.batch, 8, 128, 2048], "reference_published_ms": 9.8 if a.backbone == "resnet18" else 13.6, **out}))
