#!/usr/bin/env python
"""Development aid: intermediate gradients around attention1 in the full eval-mode model, HIP vs float64 CPU autograd."""
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
model = randomize_bn_(SemanticNetworkWithFPN("resnet18", 2, 3, num_classes=20), 3).eval()
g = torch.Generator().manual_seed(61)
x = torch.randn(2, 2, 32, 128, generator=g) * torch.tensor([20.0, 0.3]).view(1, 2, 1, 1)
meta = torch.randn(2, 3, 32, 128, generator=g) * 5.0
R = torch.randn(2, 20, 32, 128, generator=g) / (32 * 128)
stash_c, stash_g = {}, {}
def att_c(xx, sd, p):
    if p != "attention1":
        return xx
    q = F.conv2d(xx, sd[p + ".query_conv.weight"], sd[p + ".query_conv.bias"])
    k = F.conv2d(xx, sd[p + ".key_conv.weight"], sd[p + ".key_conv.bias"])
    v = F.conv2d(xx, sd[p + ".value_conv.weight"], sd[p + ".value_conv.bias"])
    t = torch.tanh(q + k)
    s = F.conv2d(t, sd[p + ".attention_conv.weight"], sd[p + ".attention_conv.bias"])
    o = v * torch.softmax(s, dim=-1)
    for n, tt in (("in", xx), ("v", v), ("t", t), ("s", s), ("o", o)):
        tt.retain_grad(); stash_c[n] = tt
    return o
ofpn._attention = att_c
orig_bn = ofpn._bn
def bn_c(xx, sdd, p):
    y = orig_bn(xx, sdd, p)
    if p.startswith("decoder_semantic"):
        xx.retain_grad(); y.retain_grad(); stash_c["pre:" + p] = xx; stash_c["bn:" + p] = y
    return y
ofpn._bn = bn_c
orig_cbr = SemanticNetworkWithFPN._t_cbr
def cbr_g(self, name, conv, bn, srcs, resid=None, act=True):
    if not name.startswith("dec"):
        return orig_cbr(self, name, conv, bn, srcs, resid, act)
    pre = fa.conv2d(srcs, conv.weight, conv.bias, conv.kernel_size[0], conv.padding[0], conv.dilation[0], None, None, None, {})
    y = fa.batch_norm(bn, pre)
    pre.retain_grad(); y.retain_grad()
    key = {"dec0": "decoder_semantic.1", "dec1": "decoder_semantic.4"}[name]
    stash_g["pre:" + key] = pre; stash_g["bn:" + key] = y
    return fa.relu(y)
SemanticNetworkWithFPN._t_cbr = cbr_g
def att_g(self, name, att, xx):
    if name != "att1":
        return xx
    wqk, bqk = att.query_conv.weight + att.key_conv.weight, att.query_conv.bias + att.key_conv.bias
    t = fa.tanh(fa.conv2d([xx], wqk, bqk, 1, 0, 1, None, None, None, {}))
    s = fa.conv2d([t], att.attention_conv.weight, att.attention_conv.bias, 1, 0, 1, None, None, None, {})
    v = fa.conv2d([xx], att.value_conv.weight, att.value_conv.bias, 1, 0, 1, None, None, None, {})
    o = fa.RowSoftmaxMulFn.apply(s, v)
    for n, tt in (("in", xx), ("v", v), ("t", t), ("s", s), ("o", o)):
        tt.retain_grad(); stash_g[n] = tt
    return o
SemanticNetworkWithFPN._t_attend = att_g
sd = {k: v.detach().clone().double() if v.is_floating_point() else v for k, v in model.state_dict().items()}
for k in sd:
    if sd[k].is_floating_point() and "running_" not in k: sd[k].requires_grad_(True)
xc, mc = x.double().requires_grad_(True), meta.double().requires_grad_(True)
(ofpn.fpn_forward(sd, xc, mc, "resnet18", True, True) * R.double()).sum().backward()
m = model.to(dev)
xg, mg = x.to(dev).requires_grad_(True), meta.to(dev).requires_grad_(True)
(m(xg, mg) * R.to(dev)).sum().backward()
rel = lambda a, b: float((a.cpu().double() - b).norm() / max(float(b.norm()), 1e-30))
for n in ("bn:decoder_semantic.4", "pre:decoder_semantic.4", "bn:decoder_semantic.1", "pre:decoder_semantic.1", "o", "v", "s", "t", "in"):
    a, b = stash_g[n], stash_c[n]
    d = (a.grad.cpu().double() - b.grad)
    print(f"{n}: value {rel(a.detach(), b.detach()):.1e}  grad {rel(a.grad, b.grad):.1e}  |grad| {float(b.grad.norm()):.2e}  worst abs {float(d.abs().max()):.2e} at {tuple(int(i) for i in (d.abs() == d.abs().max()).nonzero()[0])}")

# ---- the tail in isolation: relu -> ConvTranspose(k4 s2 p1) -> ELU + 1, fed with the float64 run's own tensor ----
yb = stash_c["bn:decoder_semantic.4"]
want = yb.grad
ct = m.decoder_semantic[6]
yin = yb.detach().float().to(dev).requires_grad_(True)
r = fa.relu(yin)
w = ct.weight
cin, cout = w.shape[0], w.shape[1]
wf = torch.zeros((cout, 2, 2, cin, 3, 3), dtype=w.dtype, device=w.device)
pairs = {0: ((0, 1), (-1, 3)), 1: ((1, 0), (0, 2))}
for py, ys in pairs.items():
    for px, xs_ in pairs.items():
        for dy, i in ys:
            for dx_, j in xs_:
                wf[:, py, px, :, dy + 1, dx_ + 1] = w[:, :, i, j].t()
b = ct.bias.repeat_interleave(4)
pre = fa.conv2d([r], wf.reshape(cout * 4, cin, 3, 3), b, 3, 1, 1, None, None, None, {})
d2 = fa.depth_to_space(pre, 2)
o = fa.elu_plus_one(d2)
for tt in (r, pre, d2): tt.retain_grad()
(o * R.to(dev)).sum().backward()
# CPU float64 of the same tail
yc = yb.detach().clone().requires_grad_(True)
rc = F.relu(yc)
prec = F.conv_transpose2d(rc, sd["decoder_semantic.6.weight"].detach(), sd["decoder_semantic.6.bias"].detach(), stride=2, padding=1)
oc = F.elu(prec) + 1
rc.retain_grad(); prec.retain_grad()
(oc * R.double()).sum().backward()
print("tail: out", f"{rel(o.detach(), oc.detach()):.1e}", " d(d2s out)", f"{rel(d2.grad, prec.grad):.1e}", " d(relu out)", f"{rel(r.grad, rc.grad):.1e}", " d(in)", f"{rel(yin.grad, yc.grad):.1e}",
      " vs full-model fp64 grad", f"{rel(yin.grad, want):.1e}")
dd = (r.grad.cpu().double() - rc.grad)
print("worst d(relu out)", float(dd.abs().max()), tuple(int(i) for i in (dd.abs() == dd.abs().max()).nonzero()[0]))
# the data-gradient conv on its own: da = d(pre), expected d(relu out) = conv_transpose-adjoint
da = pre.grad
print("pre.grad finite", bool(torch.isfinite(da).all()), "max", float(da.abs().max()))
