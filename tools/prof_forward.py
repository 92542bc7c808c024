#!/usr/bin/env python
"""One stacked forward (N samples of 64x2048) for rocprofv3 runs: `rocprofv3 ... -- python3 tools/prof_forward.py [N] [iters] [mc]`.
With a third argument `mc`: N/8 scans x T=8 MC-dropout passes through `mc_predict` (live Dropout2d, fused head + MC reduction: exactly the
launches one default bench.py step times)."""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd.salsanext import SalsaNext  # noqa: E402
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan  # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 8
iters = int(sys.argv[2]) if len(sys.argv) > 2 else 2
dev = torch.device("cuda:0")
model = seeded_model(SalsaNext).to(dev)
x, _ = synthetic_scan(n, 64, 2048)
x = x.to(dev)
mc = len(sys.argv) > 3 and sys.argv[3] == "mc"
if mc:
    from semanticlidarunc_amd.utils.mc_dropout import mc_predict  # noqa: E402
    x = x[: n // 8]
    torch.manual_seed(100)
with torch.no_grad():
    for _ in range(iters):
        y = mc_predict(model, [x], T=8)[0] if mc else model(x)
torch.cuda.synchronize()
print("done", tuple(y.shape))
