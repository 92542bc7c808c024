#!/usr/bin/env python
"""Development aid: AttentionModule backward on the model's real feature maps (all four levels) vs float64 CPU autograd."""
import os, sys
import torch
import torch.nn.functional as F
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from oracle import fpn as ofpn
from semanticlidarunc_amd import fpn_autograd as fa
from semanticlidarunc_amd.fpn import SemanticNetworkWithFPN
from semanticlidarunc_amd.testing import randomize_bn_
dev = torch.device("cuda:0")
torch.manual_seed(0)
model = randomize_bn_(SemanticNetworkWithFPN("resnet18", 2, 3, num_classes=20), 3)
g = torch.Generator().manual_seed(61)
x = torch.randn(2, 2, 32, 128, generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
meta = torch.randn(2, 3, 32, 128, generator=g) * 5.0
sd = {k: v.detach().double() for k, v in model.state_dict().items()}
# feature maps entering the attention modules (float64 oracle pieces)
m = 3
h = torch.cat([x, meta], 1).double()
xs = F.max_pool2d(F.relu(F.conv2d(h, sd["backbone.conv1.weight"], None, padding=1)), 3, 2, 1)
x1 = ofpn._stage(xs, sd, "layer1", 2, 1)
m1, m2, m3 = (F.interpolate(meta.double(), scale_factor=s, mode="nearest") for s in (1 / 2, 1 / 4, 1 / 8))
x2 = ofpn._stage(torch.cat([x1[:, :-m], m1], 1), sd, "layer2", 2, 2)
x3 = ofpn._stage(torch.cat([x2[:, :-m], m2], 1), sd, "layer3", 2, 2)
x4 = ofpn._stage(torch.cat([x3[:, :-m], m3], 1), sd, "layer4", 2, 2)
feats = {1: ofpn._cbr(x1, sd, "fpn_block1"), 2: ofpn._cbr(x2, sd, "fpn_block2"), 3: ofpn._cbr(x3, sd, "fpn_block3"), 4: ofpn._cbr(x4, sd, "fpn_block4")}
rel = lambda a, b: float((a.cpu().double() - b).norm() / max(float(b.norm()), 1e-30))
for lvl, f in feats.items():
    att = getattr(model, f"attention{lvl}")
    ac = type(att)(att.in_channels, att.out_channels).double(); ac.load_state_dict(att.state_dict())
    R = torch.randn(f.shape, generator=g, dtype=torch.float64)
    fc = f.clone().requires_grad_(True)
    q, k, v = ac.query_conv(fc), ac.key_conv(fc), ac.value_conv(fc)
    t = torch.tanh(q + k); s = ac.attention_conv(t); p = torch.softmax(s, -1); oc = v * p
    for tt in (t, s, v): tt.retain_grad()
    (oc * R).sum().backward()
    ag = att.to(dev)
    fg = f.float().to(dev).requires_grad_(True)
    wqk, bqk = ag.query_conv.weight + ag.key_conv.weight, ag.query_conv.bias + ag.key_conv.bias
    pre = fa.conv2d([fg], wqk, bqk, 1, 0, 1, None, None, None, {})
    tg = fa.tanh(pre)
    sg = fa.conv2d([tg], ag.attention_conv.weight, ag.attention_conv.bias, 1, 0, 1, None, None, None, {})
    vg = fa.conv2d([fg], ag.value_conv.weight, ag.value_conv.bias, 1, 0, 1, None, None, None, {})
    og = fa.RowSoftmaxMulFn.apply(sg, vg)
    for tt in (pre, tg, sg, vg): tt.retain_grad()
    (og * R.float().to(dev)).sum().backward()
    print(f"level {lvl} {tuple(f.shape)}: |q+k| max {float((q + k).abs().max()):.1f}  score range {float(s.min()):.1f}..{float(s.max()):.1f}  p max {float(p.max()):.3f}")
    print("   out", f"{rel(og.detach(), oc.detach()):.1e}", "ds", f"{rel(sg.grad, s.grad):.1e}", "dv", f"{rel(vg.grad, v.grad):.1e}", "dt", f"{rel(tg.grad, t.grad):.1e}",
          "dpre", f"{rel(pre.grad, (t.grad * (1 - t * t)).detach()):.1e}", "dx", f"{rel(fg.grad, fc.grad):.1e}",
          "dWq", f"{rel(ag.query_conv.weight.grad, ac.query_conv.weight.grad):.1e}", "dWatt", f"{rel(ag.attention_conv.weight.grad, ac.attention_conv.weight.grad):.1e}")
    # the same chain in fp32 on the CPU
    a32 = type(att)(att.in_channels, att.out_channels); a32.load_state_dict(att.cpu().state_dict())
    f32 = f.float().clone().requires_grad_(True)
    q3, k3, v3 = a32.query_conv(f32), a32.key_conv(f32), a32.value_conv(f32)
    t3 = torch.tanh(q3 + k3); s3 = a32.attention_conv(t3); o3 = v3 * torch.softmax(s3, -1)
    for tt in (t3, s3): tt.retain_grad()
    (o3 * R.float()).sum().backward()
    print("   fp32 CPU: ds", f"{rel(s3.grad, s.grad):.1e}", "dt", f"{rel(t3.grad, t.grad):.1e}", "dx", f"{rel(f32.grad, fc.grad):.1e}", "dWq", f"{rel(a32.query_conv.weight.grad, ac.query_conv.weight.grad):.1e}")
