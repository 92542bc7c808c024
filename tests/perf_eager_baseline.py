#!/usr/bin/env python
"""Not a test (pytest does not collect it): the SECOND baseline SURVEY.md section 8(d) asks for -- stock PyTorch-ROCm
eager (MIOpen / rocBLAS kernels) running the fp32 oracle's SalsaNext on the same MI355X, same synthetic scans, same
T = 8 stacked MC passes + the torch restatement of the uncertainty reduction.  Lives under tests/ because it executes the
oracle (test infrastructure); the product path never does.

    python tests/perf_eager_baseline.py [scans] [steps]      -> one JSON line
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import salsanext as osalsa, uncertainty as ounc  # noqa: E402
from semanticlidarunc_amd.salsanext import SalsaNext  # noqa: E402
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan  # noqa: E402

T = 8


def main():
    scans = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    sd = {k: v.to(dev) for k, v in seeded_model(SalsaNext).state_dict().items()}
    x, _ = synthetic_scan(scans, 64, 2048)
    x = x.to(dev)
    g = torch.Generator().manual_seed(0)
    out = {}
    for name, dtype in (("fp32", torch.float32), ("fp16_autocast", torch.float16)):
        def step():
            scales = {k: v.to(dev) for k, v in osalsa.draw_dropout_scales(T * scans, 0.2, g).items()}
            with torch.no_grad(), torch.autocast("cuda", dtype=dtype, enabled=dtype != torch.float32):
                logits = osalsa.salsanext_forward(sd, x.repeat(T, 1, 1, 1), scales)
            return ounc.mc_reduce(logits.float().view(T, scans, *logits.shape[1:]))
        for _ in range(2):
            step()
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            step()
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / steps
        out[name] = {"scans_per_s": round(scans / dt, 2), "ms_per_step": round(dt * 1e3, 2)}
    print(json.dumps({"baseline": "stock PyTorch-ROCm eager (MIOpen) running the oracle, T=8 stacked passes + torch MC reduction",
                      "torch": torch.__version__, "scans_per_step": scans, **out}))


if __name__ == "__main__":
    main()
