#!/usr/bin/env python
"""Time the fused block tail (conv_tail_h8) on the shapes the model runs it on (development aid; SLU_LIB_PATH=<lib> for A/B)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import _lib  # noqa: E402

if os.environ.get("SLU_LIB_PATH"):
    _lib.LIB_PATH = os.path.abspath(os.environ["SLU_LIB_PATH"])
from semanticlidarunc_amd import h8  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
for c, hh, ww, res in ((64, 64, 2048, True), (64, 32, 1024, False), (32, 64, 2048, False), (128, 32, 1024, True)):
    g = torch.Generator(device=dev).manual_seed(c)
    a1 = torch.randn(n, c // 8, hh, ww, 8, device=dev, generator=g).half()
    a2 = torch.randn(n, c // 8, hh, ww, 8, device=dev, generator=g).half()
    r = torch.randn(n, c // 8, hh, ww, 8, device=dev, generator=g).half() if res else None
    w2 = h8.pack_conv_weight_h8(torch.randn(c, c, 2, 2, device=dev, generator=g) / (4 * c) ** 0.5)
    w1 = h8.pack_conv_weight_h8(torch.randn(c, 3 * c, 1, 1, device=dev, generator=g) / (3 * c) ** 0.5)
    z = torch.zeros(c, device=dev)
    run = lambda: h8.conv_tail_h8(a1, a2, w2, w1, z, 0.01, (z + 1, z), z, 0.01, (z + 1, z), resid=r)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    real = n * hh * ww * 2.0 * c * (3 + (1 if res else 0))
    print(f"tail C={c} {hh}x{ww} N={n} resid={res}: {ms * 1e3:8.1f} us   {real / ms / 1e6:7.1f} GB/s of real traffic (a1, a2[, resid], out)", flush=True)
