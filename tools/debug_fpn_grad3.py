#!/usr/bin/env python
"""Development aid: eval-mode gradients of the HIP FPN path with attention at ONE level at a time, vs fp64 and vs fp32 CPU autograd."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fpn as ofpn
from semanticlidarunc_amd import fpn as myfpn
from semanticlidarunc_amd.fpn import SemanticNetworkWithFPN
from semanticlidarunc_amd.testing import randomize_bn_
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = randomize_bn_(SemanticNetworkWithFPN("resnet18", 2, 3, num_classes=20), 3).eval()
g = torch.Generator().manual_seed(61)
x = torch.randn(2, 2, 32, 128, generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
meta = torch.randn(2, 3, 32, 128, generator=g) * 5.0
R = torch.randn(2, 20, 32, 128, generator=g) / (32 * 128)
LEVELS = set()
orig_att = ofpn._attention
def patched(xx, sd, p):
    return orig_att(xx, sd, p) if int(p[-1]) in LEVELS else xx
ofpn._attention = patched
orig_t = SemanticNetworkWithFPN._t_attend
def t_patched(self, name, att, xx):
    return orig_t(self, name, att, xx) if int(name[-1]) in LEVELS else xx
SemanticNetworkWithFPN._t_attend = t_patched
rel = lambda a, b: float((a.cpu().double() - b.double()).norm() / max(float(b.norm()), 1e-30))
def cpu_run(dtype):
    sd = {k: v.detach().clone().to(dtype) if v.is_floating_point() else v for k, v in model.state_dict().items()}
    xc, mc = x.to(dtype).clone().requires_grad_(True), meta.to(dtype).clone().requires_grad_(True)
    for k in sd:
        if sd[k].is_floating_point() and "running_" not in k: sd[k].requires_grad_(True)
    out = ofpn.fpn_forward(sd, xc, mc, "resnet18", True, True)
    (out * R.to(dtype)).sum().backward()
    return out.detach(), xc.grad, sd
for lv in (set(), {1}, {2}, {3}, {4}, {1, 2, 3, 4}):
    LEVELS = lv
    o64, dx64, sd64 = cpu_run(torch.float64)
    o32, dx32, sd32 = cpu_run(torch.float32)
    m = model.to(dev)
    for p in m.parameters(): p.grad = None
    xg, mg = x.to(dev).requires_grad_(True), meta.to(dev).requires_grad_(True)
    og = m(xg, mg)
    (og * R.to(dev)).sum().backward()
    key = "backbone.layer1.0.conv1.weight"
    print(f"levels {sorted(lv)}: GPU dx {rel(xg.grad, dx64):.1e} dW {rel(dict(m.named_parameters())[key].grad, sd64[key].grad):.1e} | CPU fp32 dx {rel(dx32, dx64):.1e} dW {rel(sd32[key].grad, sd64[key].grad):.1e}")
    model = model.cpu()
