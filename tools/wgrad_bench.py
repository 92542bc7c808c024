#!/usr/bin/env python
"""Per-shape timing of the weight-gradient kernel (training path, BASELINE configs[1]: batch 4 of 64x2048): python tools/wgrad_bench.py [B]"""
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import ops  # noqa: E402

b = int(sys.argv[1]) if len(sys.argv) > 1 else 4
dev = torch.device("cuda:0")
# (cin, cout, k, dil, pad, H, W) of SalsaNext's conv layers, one per distinct shape
SHAPES = [(5, 32, 1, 1, 0, 64, 2048), (32, 32, 3, 1, 1, 64, 2048), (32, 32, 3, 2, 2, 64, 2048), (32, 64, 3, 1, 1, 64, 2048), (64, 64, 3, 2, 2, 64, 2048),
          (64, 64, 2, 2, 1, 64, 2048), (192, 64, 1, 1, 0, 64, 2048), (80, 32, 3, 1, 1, 64, 2048), (96, 32, 1, 1, 0, 64, 2048),
          (64, 128, 3, 1, 1, 32, 1024), (128, 128, 3, 2, 2, 32, 1024), (128, 128, 2, 2, 1, 32, 1024), (384, 128, 1, 1, 0, 32, 1024), (160, 64, 3, 1, 1, 32, 1024),
          (128, 256, 3, 1, 1, 16, 512), (256, 256, 3, 2, 2, 16, 512), (768, 256, 1, 1, 0, 16, 512), (288, 128, 3, 1, 1, 16, 512),
          (256, 256, 3, 1, 1, 8, 256), (256, 256, 3, 2, 2, 8, 256), (320, 128, 3, 1, 1, 8, 256), (256, 256, 3, 2, 2, 4, 128), (768, 256, 1, 1, 0, 4, 128)]
tot = 0.0
for cin, cout, k, dil, pad, h, w in SHAPES:
    cop, cip = (cout + 31) // 32 * 32, (cin + 31) // 32 * 32
    da = torch.randn(b, h * w, cop, device=dev)
    xin = torch.randn(b, h * w, cip, device=dev)
    for _ in range(2):
        ops.conv2d_wgrad(da, xin, b, h, w, cout, cin, k, dil, pad)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(5):
        ops.conv2d_wgrad(da, xin, b, h, w, cout, cin, k, dil, pad)
    e1.record()
    torch.cuda.synchronize()
    us = e0.elapsed_time(e1) / 5 * 1e3
    fl = 2.0 * cin * cout * k * k * b * h * w
    tot += us
    print(f"{cin:4d}->{cout:3d} k{k}d{dil} {h:3d}x{w:4d}  {us:8.1f} us  {fl / us / 1e6:7.1f} TFLOP/s ({fl / us / 1e6 / 157.3 * 100:5.1f} % of fp32 MFMA peak)", flush=True)
print(f"sum {tot:.0f} us")
# the 1x1 layers straight from NCHW (slu_conv1x1_wgrad_nchw) against channel-last copies + the kernel above
from semanticlidarunc_amd.ops import ConvSource  # noqa: E402
for chans, cout, h, w in [((5,), 32, 64, 2048), ((64, 64, 64), 64, 64, 2048), ((32, 32, 32), 32, 64, 2048), ((128, 128, 128), 128, 32, 1024),
                          ((256, 256, 256), 256, 16, 512), ((256, 256, 256), 256, 4, 128)]:
    da = torch.randn(b, cout, h, w, device=dev)
    xs = [ConvSource(torch.randn(b, c, h, w, device=dev)) for c in chans]
    cin = sum(chans)

    def old():
        return ops.conv2d_wgrad(ops.nchw_to_nhwc(da), ops.gather_nhwc(xs), b, h, w, cout, cin, 1, 1, 0)

    def new():
        return ops.conv1x1_wgrad_nchw(da, xs)

    res = []
    for fn in (old, new):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    gb = 4.0 * b * h * w * (cin + cout) / 1e3
    print(f"1x1 {cin:4d}->{cout:3d} {h:3d}x{w:4d}  channel-last copies + kernel {res[0]:8.1f} us   from NCHW {res[1]:8.1f} us ({gb / res[1]:6.0f} GB/s of its inputs)", flush=True)

# the 3x3 / 2x2-dilated layers straight from NCHW (slu_conv2d_wgrad_nchw) against channel-last copies + the kernel above
for cin, cout, k, dil, pad, h, w in [(32, 32, 3, 1, 1, 64, 2048), (32, 64, 3, 1, 1, 64, 2048), (64, 64, 3, 2, 2, 64, 2048), (64, 64, 2, 2, 1, 64, 2048),
                                     (128, 128, 3, 2, 2, 32, 1024), (128, 128, 2, 2, 1, 32, 1024), (256, 256, 3, 2, 2, 16, 512), (256, 256, 3, 1, 1, 8, 256)]:
    da = torch.randn(b, cout, h, w, device=dev)
    xs = [ConvSource(torch.randn(b, cin, h, w, device=dev))]

    def old():
        return ops.conv2d_wgrad(ops.nchw_to_nhwc(da), ops.gather_nhwc(xs), b, h, w, cout, cin, k, dil, pad)

    def new():
        return ops.conv2d_wgrad_nchw(da, xs, k, dil, pad)

    res = []
    for fn in (old, new):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * cin * cout * k * k * b * h * w
    print(f"k{k}d{dil} {cin:4d}->{cout:3d} {h:3d}x{w:4d}  channel-last copies + kernel {res[0]:8.1f} us   from NCHW {res[1]:8.1f} us "
          f"({fl / res[1] / 1e6 / 157.3 * 100:5.1f} % of the fp32 MFMA peak)", flush=True)

# UpBlock.conv1: PixelShuffle(x) with a multiplier | skip with a multiplier
for cx, cs, cout, h, w in [(64, 64, 32, 64, 2048), (128, 128, 64, 32, 1024), (128, 256, 128, 16, 512), (256, 256, 128, 8, 256)]:
    cin = cx // 4 + cs
    da = torch.randn(b, cout, h, w, device=dev)
    xs = [ConvSource(torch.randn(b, cx, h // 2, w // 2, device=dev), torch.ones(b, cx, device=dev) * 1.25, True),
          ConvSource(torch.randn(b, cs, h, w, device=dev), torch.ones(b, cs, device=dev) * 1.25)]
    res = []
    for fn in (lambda: ops.conv2d_wgrad(ops.nchw_to_nhwc(da), ops.gather_nhwc(xs), b, h, w, cout, cin, 3, 1, 1), lambda: ops.conv2d_wgrad_nchw(da, xs, 3, 1, 1)):
        for _ in range(2):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(5):
            fn()
        e1.record()
        torch.cuda.synchronize()
        res.append(e0.elapsed_time(e1) / 5 * 1e3)
    fl = 2.0 * cin * cout * 9 * b * h * w
    print(f"k3d1 PixelShuffle({cx}) | {cs} -> {cout} {h:3d}x{w:4d}  channel-last copies + kernel {res[0]:8.1f} us   from NCHW {res[1]:8.1f} us "
          f"({fl / res[1] / 1e6 / 157.3 * 100:5.1f} % of the fp32 MFMA peak)", flush=True)
