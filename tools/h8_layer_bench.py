#!/usr/bin/env python
"""Time single h8 conv layers of the 64x2048 stack: `python tools/h8_layer_bench.py [N]` (development aid)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import _lib  # noqa: E402

if os.environ.get("SLU_LIB_PATH"):          # A/B runs of two builds on the same box
    _lib.LIB_PATH = os.path.abspath(os.environ["SLU_LIB_PATH"])
from semanticlidarunc_amd import h8  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 32
dev = torch.device("cuda:0")
LAYERS = [  # (cin parts, cout, k, dil, pad, H, W, resid)
    ([32], 32, 3, 1, 1, 64, 2048, False), ([32], 32, 3, 2, 2, 64, 2048, True), ([64], 64, 3, 2, 2, 64, 2048, False),
    ([64], 64, 2, 2, 1, 64, 2048, False), ([64, 64, 64], 64, 1, 1, 0, 64, 2048, True), ([32], 32, 1, 1, 0, 64, 2048, False),
    ([128], 128, 3, 2, 2, 32, 1024, False), ([256], 256, 3, 2, 2, 16, 512, False), ([128, 128, 128], 128, 1, 1, 0, 32, 1024, True),
    ([32], 32, 2, 2, 1, 64, 2048, False), ([128], 128, 2, 2, 1, 32, 1024, False), ([256], 256, 2, 2, 1, 16, 512, False),
    ([128], 128, 3, 1, 1, 32, 1024, False), ([256], 256, 3, 1, 1, 8, 256, False),
]
sel = os.environ.get("SLU_LAYERS")
for li, (parts, cout, k, dil, pad, H, W, res) in enumerate(LAYERS):
    if sel and str(li) not in sel.split(","):
        continue
    g = torch.Generator(device=dev).manual_seed(li)
    srcs = [h8.H8Source(torch.randn(n, c // 8, H, W, 8, device=dev, generator=g).half()) for c in parts]
    cin = sum(parts)
    w = h8.pack_conv_weight_h8(torch.randn(cout, cin, k, k, device=dev, generator=g) / (cin * k * k) ** 0.5)
    bias = torch.zeros(cout, device=dev)
    resid = torch.randn(n, cout // 8, H, W, 8, device=dev, generator=g).half() if res else None
    run = lambda: h8.conv2d_h8(srcs, w, cin, cout, k, dil, pad, bias=bias, slope=0.01, bn_a=bias + 1, bn_b=bias, resid=resid)
    for _ in range(3):
        run()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(10):
        run()
    e1.record()
    torch.cuda.synchronize()
    ms = e0.elapsed_time(e1) / 10
    by = n * H * W * 2.0 * (cin + cout * (2 if res else 1))
    fl = 2.0 * cin * cout * k * k * n * H * W
    print(f"L{li} {parts}->{cout} k{k}d{dil} {H}x{W}: {ms*1e3:8.1f} us  {by/ms/1e6:7.1f} GB/s(incl resid)  {fl/ms/1e9:7.1f} TF/s", flush=True)
