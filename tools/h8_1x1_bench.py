#!/usr/bin/env python
"""Time the wide 1x1 layers of the 64x2048 stack (development aid): `SLU_H8_GEMM1X1=0/1 python tools/h8_1x1_bench.py [N]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import h8, ops  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
LAYERS = [([256, 256, 256], 256, 16, 512, True), ([256, 256, 256], 256, 8, 256, True), ([256, 256, 256], 256, 4, 128, True),
          ([128, 128, 128], 128, 32, 1024, True), ([128, 128, 128], 128, 16, 512, True), ([64], 128, 32, 1024, False), ([128], 256, 16, 512, False),
          ([256], 256, 8, 256, False)]
for li, (parts, cout, H, W, res) in enumerate(LAYERS):
    g = torch.Generator(device=dev).manual_seed(li)
    srcs = [h8.H8Source(torch.randn(n, c // 8, H, W, 8, device=dev, generator=g).half()) for c in parts]
    cin = sum(parts)
    w = h8.pack_conv_weight_h8(torch.randn(cout, cin, 1, 1, device=dev, generator=g) / cin ** 0.5)
    bias = torch.zeros(cout, device=dev)
    resid = torch.randn(n, cout // 8, H, W, 8, device=dev, generator=g).half() if res else None
    run = lambda: h8.conv2d_h8(srcs, w, cin, cout, 1, 1, 0, bias=bias, slope=0.01, bn_a=bias + 1, bn_b=bias, resid=resid)
    ops.TIMING, ops.TIMING_TAGS = [], []
    run()
    name = ops.TIMING[0][0]
    ops.TIMING = None
    for _ in range(3):
        run()
    ts = []
    for r in range(5):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            run()
        e1.record()
        torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) / 5)
    ms = sorted(ts)[2]
    by = n * H * W * 2.0 * (cin + cout * (2 if res else 1))
    fl = 2.0 * cin * cout * n * H * W
    print(f"L{li} {parts}->{cout} {H}x{W}: {ms*1e3:8.1f} us  {by/ms/1e6:7.1f} GB/s(incl resid)  {fl/ms/1e9:7.1f} TF/s  {name}", flush=True)
