#!/usr/bin/env python
"""Where the tiled h8 kernel spends its shader clocks (development aid).  Needs a library built with -DSLU_H8_PROF:

    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -DSLU_H8_PROF -Iinclude -Isemanticlidarunc_amd/csrc \
          -c semanticlidarunc_amd/csrc/conv2d_h8.hip -o build_prof/conv2d_h8.o     # + link with the other objects
    python tools/h8_phase_prof.py build_prof/libslu_hip.so [N]

Prints, per layer, the mean clocks of wave 0 of a workgroup in each phase (wait for DMA | barrier | DMA issue | LDS+MFMA | epilogue)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from semanticlidarunc_amd import h8  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 32
lib = _lib.load()
lib.slu_h8_prof_read.restype, lib.slu_h8_prof_read.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
dev = torch.device("cuda:0")
LAYERS = [([32], 32, 3, 1, 1, 64, 2048, False), ([32], 32, 3, 2, 2, 64, 2048, True), ([64], 64, 3, 2, 2, 64, 2048, False),
          ([64], 64, 2, 2, 1, 64, 2048, False), ([128], 128, 3, 2, 2, 32, 1024, False), ([256], 256, 3, 2, 2, 16, 512, False)]
buf = (C.c_ulonglong * 8)()
for li, (parts, cout, k, dil, pad, H, W, res) in enumerate(LAYERS):
    g = torch.Generator(device=dev).manual_seed(li)
    srcs = [h8.H8Source(torch.randn(n, c // 8, H, W, 8, device=dev, generator=g).half()) for c in parts]
    cin = sum(parts)
    w = h8.pack_conv_weight_h8(torch.randn(cout, cin, k, k, device=dev, generator=g) / (cin * k * k) ** 0.5)
    bias = torch.zeros(cout, device=dev)
    resid = torch.randn(n, cout // 8, H, W, 8, device=dev, generator=g).half() if res else None
    run = lambda: h8.conv2d_h8(srcs, w, cin, cout, k, dil, pad, bias=bias, slope=0.01, bn_a=bias + 1, bn_b=bias, resid=resid)
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
    if tot == 0:
        print(f"L{li} {parts}->{cout} k{k}d{dil} {H}x{W}: {us:7.1f} us -- not the tiled kernel (ring3_h8_kernel takes the full-resolution 3x3 layers)", flush=True)
        continue
    print(f"L{li} {parts}->{cout} k{k}d{dil} {H}x{W}: {us:7.1f} us, {nwg} WGs, clocks/WG {tot / nwg:9.0f} ({tot / nwg / us:6.1f} clk/us): "
          + "  ".join(f"{nm} {100.0 * x / tot:4.1f}%" for nm, x in zip(("wait", "barrier", "issue", "mfma", "epilogue"), v[:5])), flush=True)
