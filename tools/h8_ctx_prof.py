#!/usr/bin/env python
"""Where the fused ResContextBlock kernel spends its shader clocks (development aid; run on the GPU box):
    hipcc --offload-arch=gfx950 -O3 -std=c++17 -fPIC -shared -DSLU_CTX_PROF -Iinclude -Isemanticlidarunc_amd/csrc \
          semanticlidarunc_amd/csrc/ctx_block_h8.hip -o /tmp/libctx_prof.so
    python tools/h8_ctx_prof.py /tmp/libctx_prof.so [N]
Prints the mean clocks of wave 0 of a workgroup per phase (P1 | barrier | P2 | barrier | P3 | barrier)."""
import ctypes as C
import os
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import _lib, h8  # noqa: E402

lib = C.CDLL(os.path.abspath(sys.argv[1]))
n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
lib.slu_ctx_prof_read.restype, lib.slu_ctx_prof_read.argtypes = C.c_int, [C.POINTER(C.c_ulonglong)]
lib.slu_ctx_block_h8_fwd.restype, lib.slu_ctx_block_h8_fwd.argtypes = C.c_int, [C.POINTER(_lib.CtxBlockH8Desc), C.c_void_p]
dev = torch.device("cuda:0")
buf = (C.c_ulonglong * 8)()
for cin in (5, 32):
    g = torch.Generator(device=dev).manual_seed(cin)
    x = h8.to_h8(torch.randn(n, cin, 64, 2048, device=dev, generator=g))
    w = [h8.pack_conv_weight_h8(torch.randn(32, c, k, k, device=dev, generator=g) / (c * k * k) ** 0.5) for c, k in ((cin, 1), (32, 3), (32, 3))]
    v32 = torch.rand(32, device=dev, generator=g) + 0.5
    out = torch.empty((n, 4, 64, 2048, 8), dtype=torch.float16, device=dev)
    d = _lib.CtxBlockH8Desc()
    d.x, d.N, d.H, d.W, d.Cin, d.C = x.data_ptr(), n, 64, 2048, cin, 32
    d.w1, d.w2, d.w3 = (t.data_ptr() for t in w)
    d.bias1 = d.bias2 = d.bias3 = d.bn1_a = d.bn1_b = d.bn2_a = d.bn2_b = v32.data_ptr()
    d.slope, d.out = 0.01, out.data_ptr()
    st = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        assert lib.slu_ctx_block_h8_fwd(C.byref(d), st) == 0
    torch.cuda.synchronize()
    lib.slu_ctx_prof_read(buf)
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    assert lib.slu_ctx_block_h8_fwd(C.byref(d), st) == 0
    e1.record()
    torch.cuda.synchronize()
    lib.slu_ctx_prof_read(buf)
    v = [int(t) for t in buf]
    nwg, tot, us = max(1, v[6]), sum(v[:6]), e0.elapsed_time(e1) * 1e3
    tiles = n * 8 * 32 / nwg
    print(f"cin={cin} N={n}: {us:7.1f} us, {nwg} WGs x {tiles:.0f} tiles, clocks/tile {tot / nwg / tiles:8.0f} ({tot / nwg / us:6.1f} clk/us): "
          + "  ".join(f"{nm} {x / nwg / tiles:6.0f}" for nm, x in zip(("P1", "bar", "P2", "bar", "P3", "bar"), v[:6])), flush=True)
