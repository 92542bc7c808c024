#!/usr/bin/env python
"""Time one ResContextBlock on the h8 path, fused kernel vs the three separate launches: `python tools/h8_ctx_bench.py [N] [cin]`."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import h8  # noqa: E402
from semanticlidarunc_amd import salsanext as sn  # noqa: E402
from semanticlidarunc_amd.testing import randomize_bn_  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 64
dev = torch.device("cuda:0")
for cin in ([int(sys.argv[2])] if len(sys.argv) > 2 else [5, 32]):
    torch.manual_seed(0)
    blk = randomize_bn_(sn.ResContextBlock(cin, 32), 1).eval().to(dev)
    xh = h8.to_h8(torch.randn(n, cin, 64, 2048, device=dev))
    for fuse in (True, False):
        sn._FUSE_CTX = fuse
        with torch.no_grad():
            for _ in range(3):
                y = blk(xh)
            torch.cuda.synchronize()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(20):
                y = blk(xh)
            e1.record()
            torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 20
        gb = n * 64 * 2048 * 2 * (8 * ((cin + 7) // 8) + 32) / 1e9
        print(f"cin={cin:2d} N={n} fused={fuse!s:5}  {ms * 1e3:8.1f} us   {gb / ms:7.1f} GB/s of (x in + out)   {2 * (cin * 32 + 18 * 1024) * n * 64 * 2048 / ms / 1e9:7.1f} TFLOP/s")
