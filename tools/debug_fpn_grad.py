#!/usr/bin/env python
"""Development aid: gradients of the HIP FPN training path against torch CPU autograd of the functional oracle (oracle/fpn.py), in eval-mode
and in train-mode BatchNorm, in float64 on the CPU side.  python tools/debug_fpn_grad.py"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fpn as ofpn  # noqa: E402
from semanticlidarunc_amd.fpn import SemanticNetworkWithFPN  # noqa: E402
from semanticlidarunc_amd.testing import randomize_bn_  # noqa: E402

dev = torch.device("cuda:0")
kw = dict(backbone="resnet18", input_channels=2, meta_channel_dim=3, num_classes=20)
att = "--noatt" not in sys.argv
kw["attention"] = att
torch.manual_seed(0)
model = randomize_bn_(SemanticNetworkWithFPN(**kw), 3)
g = torch.Generator().manual_seed(61)
x = torch.randn(2, 2, 32, 128, generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
meta = torch.randn(2, 3, 32, 128, generator=g) * 5.0
R = torch.randn(2, 20, 32, 128, generator=g) / (32 * 128)
for train in (False, True):
    sd = {k: v.detach().clone().double().requires_grad_(v.is_floating_point() and 'running_' not in k) for k, v in model.state_dict().items()}
    xc, mc = x.double().requires_grad_(True), meta.double().requires_grad_(True)
    ofpn.BN_TRAIN = train
    out_c = ofpn.fpn_forward(sd, xc, mc, "resnet18", att, True)
    (out_c * R.double()).sum().backward()
    m = model.to(dev).train(train)
    for p in m.parameters():
        p.grad = None
    xg, mg = x.to(dev).requires_grad_(True), meta.to(dev).requires_grad_(True)
    out_g = m(xg, mg)
    (out_g * R.to(dev)).sum().backward()
    rel = lambda a, b: float((a.cpu().double() - b).norm() / max(float(b.norm()), 1e-30))
    print(f"train={train}: out {rel(out_g.detach(), out_c.detach()):.2e}  dx {rel(xg.grad, xc.grad):.2e}  dmeta {rel(mg.grad, mc.grad):.2e}")
    worst = []
    for n, p in m.named_parameters():
        key = n
        if p.grad is None or sd[key].grad is None:
            continue
        worst.append((rel(p.grad, sd[key].grad), n, float(sd[key].grad.norm())))
    worst.sort(reverse=True)
    for r, n, nn_ in worst[:14]:
        print(f"    {r:.2e}  |g| {nn_:.2e}  {n}")
    model = model.cpu()
