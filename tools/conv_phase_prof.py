#!/usr/bin/env python
"""Where the fp32 conv kernel (training forward / data gradient) spends its shader clocks (development aid).  Needs a library whose conv2d.hip was
built with -DSLU_CONV_PROF (exports slu_conv_prof_read):
    python tools/conv_phase_prof.py ab_libs/libslu_convprof.so [B]
Prints per layer the share of wave 0's clocks per phase: barrier | loads issued | loads landed + LDS written | barrier | MFMA | epilogue."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from semanticlidarunc_amd import ops  # noqa: E402
from semanticlidarunc_amd.ops import ConvSource  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 4
lib = _lib.load()
lib.slu_conv_prof_read.restype, lib.slu_conv_prof_read.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
dev = torch.device("cuda:0")
LAYERS = [(32, 32, 3, 1, 1, 64, 2048), (64, 64, 3, 2, 2, 64, 2048), (64, 64, 2, 2, 1, 64, 2048), (192, 64, 1, 1, 0, 64, 2048), (128, 128, 3, 2, 2, 32, 1024),
          (384, 128, 1, 1, 0, 32, 1024), (256, 256, 3, 2, 2, 16, 512), (256, 256, 3, 1, 1, 8, 256)]
buf = (C.c_ulonglong * 8)()
names = ["barrier0", "regs->LDS (waits for loads)", "barrier1", "issue next loads", "MFMA", "epilogue"]
for li, (cin, cout, k, dil, pad, H, W) in enumerate(LAYERS):
    g = torch.Generator(device=dev).manual_seed(li)
    x = torch.randn(n, cin, H, W, device=dev, generator=g)
    w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, device=dev, generator=g) / (cin * k * k) ** 0.5)
    bias = torch.zeros(cout, device=dev)
    run = lambda: ops.conv2d_fused([ConvSource(x)], w, cout, k, dil, pad, bias=bias, slope=0.01)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    lib.slu_conv_prof_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lib.slu_conv_prof_read(buf)
    v = [int(t) for t in buf]
    tot, nwg = max(1, sum(v[:6])), max(1, v[6])
    us = e0.elapsed_time(e1) * 1e3
    fl = 2.0 * n * H * W * cin * cout * k * k
    print(f"{cin:4d}->{cout:3d} k{k}d{dil} {H}x{W}: {us:7.1f} us {fl / us / 1e6:6.1f} TF/s  {nwg} workgroups, {tot / nwg / 100:7.0f} x100 clk each (memtime units);  "
          + "  ".join(f"{nm} {100.0 * t / tot:4.1f}%" for nm, t in zip(names, v[:6])), flush=True)
