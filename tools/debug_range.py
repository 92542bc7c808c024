#!/usr/bin/env python
"""Development aid: errors of the fp16-storage and split-fp16 paths at KITTI-like magnitudes (tests/test_gpu_h8_range.py's cases)."""
import os, sys
import torch
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, os.path.join(os.path.dirname(os.path.dirname(os.path.abspath(__file__))), "tests"))
from test_gpu_h8_range import adapt_bn_, kitti_like_scan, oracle_with_stored_maxima
from semanticlidarunc_amd import salsanext as sn
from semanticlidarunc_amd.salsanext import SalsaNext
from semanticlidarunc_amd.testing import seeded_model
cuda = torch.device("cuda:0")
for case in ("adapted_bn", "random_bn_x4", "random_bn"):
    model = seeded_model(SalsaNext).to(cuda)
    x = kitti_like_scan(2, 64, 512, seed=21)
    if case == "adapted_bn":
        adapt_bn_(model, kitti_like_scan(2, 64, 512, seed=22).to(cuda))
    elif case == "random_bn_x4":
        x = x * 4.0
    sd = {k: v.detach().cpu() for k, v in model.state_dict().items()}
    want, peak = oracle_with_stored_maxima(sd, x)
    bn_a = max(float((v / torch.sqrt(sd[k[:-6] + "running_var"] + 1e-5)).abs().max()) for k, v in sd.items() if k.endswith("bn1.weight") or k.endswith("bn2.weight") or k.endswith("bn3.weight") or k.endswith("bn4.weight"))
    for prec in ("f16", "f16x3", "fp32"):
        sn.set_conv_precision(prec)
        with torch.no_grad():
            got = model(x.to(cuda)).cpu()
        sn.set_conv_precision("fp32")
        pg, pw = torch.softmax(got, 1), torch.softmax(want, 1)
        ent = lambda p: -(p * torch.log(p.clamp_min(1e-8))).sum(1) / torch.log(torch.tensor(20.0))
        d = (got - want).abs()
        print(f"{case:14s} {prec:6s} peak {peak:8.1f} max BN gain {bn_a:8.1f} logit scale {float(want.abs().max()):6.2f} |dlogit| max {float(d.max()):.2e} p99.9 {float(d.flatten().kthvalue(int(d.numel()*0.999)).values):.2e} "
              f"median {float(d.median()):.2e}  |dp| {float((pg-pw).abs().max()):.2e}  |dH| {float((ent(pg)-ent(pw)).abs().max()):.2e}  flips {float((got.argmax(1)!=want.argmax(1)).float().mean()):.2e}")
