#!/usr/bin/env python
"""Where the streaming 1x1 h8 kernel spends its shader clocks (development aid; library built with -DSLU_H8_PROF, see tools/h8_phase_prof.py):
    python tools/h8_1x1_prof.py <libslu_prof.so> [N]
Per layer: mean clocks of wave 0 of a workgroup per phase (input loads issued + first barrier | weight staging | second barrier |
MFMAs incl. waiting for the inputs | epilogue)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from semanticlidarunc_amd import h8  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib = _lib.load()
lib.slu_h8_prof_read.restype, lib.slu_h8_prof_read.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
dev = torch.device("cuda:0")
LAYERS = [([256, 256, 256], 256, 16, 512), ([128, 128, 128], 128, 32, 1024), ([128], 256, 16, 512), ([64], 128, 32, 1024)]
buf = (C.c_ulonglong * 8)()
for li, (parts, cout, H, W) in enumerate(LAYERS):
    g = torch.Generator(device=dev).manual_seed(li)
    srcs = [h8.H8Source(torch.randn(n, c // 8, H, W, 8, device=dev, generator=g).half()) for c in parts]
    cin = sum(parts)
    w = h8.pack_conv_weight_h8(torch.randn(cout, cin, 1, 1, device=dev, generator=g) / cin ** 0.5)
    bias = torch.zeros(cout, device=dev)
    run = lambda: h8.conv2d_h8(srcs, w, cin, cout, 1, 1, 0, bias=bias, slope=0.01, bn_a=bias + 1, bn_b=bias)
    for _ in range(2):
        run()
    torch.cuda.synchronize()
    lib.slu_h8_prof_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    run()
    e1.record()
    torch.cuda.synchronize()
    lib.slu_h8_prof_read(buf)
    v = [int(x) for x in buf]
    nwg = max(1, v[5])
    tot = sum(v[:5])
    us = e0.elapsed_time(e1) * 1e3
    print(f"{cin}->{cout} {H}x{W}: {us:7.1f} us, {nwg} WGs, clocks/WG {tot / nwg:9.0f} (x {nwg / 512:.1f} rounds of 512 resident WGs = {tot / nwg * nwg / 512 / us:6.1f} clk/us): "
          + "  ".join(f"{nm} {100.0 * x / tot:4.1f}% ({x / nwg:7.0f})" for nm, x in zip(("loads+bar1", "wstage", "bar2", "mfma", "epilogue"), v[:5])), flush=True)
