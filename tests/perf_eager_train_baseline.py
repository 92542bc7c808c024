#!/usr/bin/env python
"""Not a test: stock PyTorch-ROCm eager training step (fp32, train-mode BatchNorm, Dropout2d live, NLL + Lovasz from the
oracle, autograd backward, AdamW) on the same MI355X and batch as tools/train_bench.py (BASELINE configs[1]).
    python tests/perf_eager_train_baseline.py [batch] [steps]
"""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import losses as olosses, salsanext as osalsa  # noqa: E402
from semanticlidarunc_amd.salsanext import SalsaNext  # noqa: E402
from semanticlidarunc_amd.testing import seeded_model, synthetic_scan  # noqa: E402


def main():
    batch = int(sys.argv[1]) if len(sys.argv) > 1 else 4
    steps = int(sys.argv[2]) if len(sys.argv) > 2 else 5
    dev = torch.device("cuda:0")
    sd = {k: v.to(dev).clone() for k, v in seeded_model(SalsaNext).state_dict().items()}
    params = [v.requires_grad_(True) for k, v in sd.items() if v.is_floating_point() and "running_" not in k]
    opt = torch.optim.AdamW(params, lr=1e-3, weight_decay=1e-4)
    x, y = synthetic_scan(batch, 64, 2048)
    x, y = x.to(dev), y.to(dev)
    g = torch.Generator().manual_seed(0)

    def step():
        opt.zero_grad(set_to_none=True)
        scales = {k: v.to(dev) for k, v in osalsa.draw_dropout_scales(batch, 0.2, g).items()}
        logits = osalsa.salsanext_forward(sd, x, scales, bn_train=True)
        loss = olosses.salsanext_loss(logits, y)[0] if hasattr(olosses, "salsanext_loss") else None
        if loss is None:
            p = logits.softmax(1)
            loss = torch.nn.functional.nll_loss(torch.log(p.clamp(min=1e-8)), y) + olosses.lovasz_softmax(p, y, ignore_index=0)
        loss.backward()
        opt.step()

    for _ in range(2):
        step()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    print(json.dumps({"baseline": "stock PyTorch-ROCm eager training step (oracle network + loss, autograd, AdamW), fp32", "batch": batch,
                      "ms_per_step": round(dt * 1e3, 2), "train_scans_per_s": round(batch / dt, 2)}))


if __name__ == "__main__":
    main()
