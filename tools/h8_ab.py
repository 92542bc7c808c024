#!/usr/bin/env python
"""A/B the scheduling options of the 8-wave 128-channel configuration of conv_h8_kernel in ONE process (development aid).

Needs a library built with -DSLU_H8_AB (exports slu_h8_dev_set_opt):
    tools/h8_resources.sh -DSLU_H8_AB && hipcc -shared ... -o ab_libs/libslu_ab.so
    python tools/h8_ab.py ab_libs/libslu_ab.so [N] [rounds]

Per layer: every variant's output is compared bit for bit with variant (opt 0, kpc2 0), then the variants are timed in interleaved rounds
(one launch sequence of `reps` launches per variant per round); median and minimum microseconds per launch are printed."""
import ctypes as C
import os
import statistics
import sys

import torch

sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
from semanticlidarunc_amd import _lib  # noqa: E402

_lib.LIB_PATH = os.path.abspath(sys.argv[1])
from semanticlidarunc_amd import h8  # noqa: E402

n = int(sys.argv[2]) if len(sys.argv) > 2 else 64
rounds = int(sys.argv[3]) if len(sys.argv) > 3 else 7
reps = 5
lib = _lib.load()
lib.slu_h8_dev_set_opt.restype, lib.slu_h8_dev_set_opt.argtypes = None, [C.c_int, C.c_int]
dev = torch.device("cuda:0")
LAYERS = [  # (cin, cout, k, dil, pad, H, W)
    (128, 128, 3, 2, 2, 32, 1024), (256, 256, 3, 2, 2, 16, 512), (128, 128, 3, 1, 1, 32, 1024), (256, 256, 3, 1, 1, 8, 256),
    (128, 128, 2, 2, 1, 32, 1024), (256, 256, 2, 2, 1, 16, 512), (256, 256, 2, 2, 1, 8, 256),
]
VARIANTS = [(0, 0), (100, 0), (8, 0), (15, 0), (0, 1), (15, 1)]      # (OPT, two K-steps per barrier for 2x2); 100 = the general multi-source form
sel = os.environ.get("SLU_LAYERS")
for li, (cin, cout, k, dil, pad, H, W) in enumerate(LAYERS):
    if sel and str(li) not in sel.split(","):
        continue
    g = torch.Generator(device=dev).manual_seed(li)
    src = [h8.H8Source(torch.randn(n, cin // 8, H, W, 8, device=dev, generator=g).half())]
    w = h8.pack_conv_weight_h8(torch.randn(cout, cin, k, k, device=dev, generator=g) / (cin * k * k) ** 0.5)
    bias = torch.randn(cout, device=dev, generator=g) * 0.1
    bn_a, bn_b = torch.rand(cout, device=dev, generator=g) + 0.5, torch.randn(cout, device=dev, generator=g) * 0.1
    run = lambda: h8.conv2d_h8(src, w, cin, cout, k, dil, pad, bias=bias, slope=0.01, bn_a=bn_a, bn_b=bn_b)
    variants = [v for v in VARIANTS if k == 2 or v[1] == 0]
    lib.slu_h8_dev_set_opt(0, 0)
    ref = run()
    torch.cuda.synchronize()
    times = {v: [] for v in variants}
    for v in variants:
        lib.slu_h8_dev_set_opt(*v)
        out = run()
        torch.cuda.synchronize()
        if not torch.equal(out, ref):
            d = (out.float() - ref.float()).abs().max().item()
            print(f"L{li} variant {v}: OUTPUT DIFFERS from variant (0, 0): max abs {d}", flush=True)
    for r in range(rounds):
        for v in variants:
            lib.slu_h8_dev_set_opt(*v)
            run()
            e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
            e0.record()
            for _ in range(reps):
                run()
            e1.record()
            torch.cuda.synchronize()
            times[v].append(e0.elapsed_time(e1) / reps * 1e3)
    fl = 2.0 * cin * cout * k * k * n * H * W
    base = statistics.median(times[variants[0]])
    print(f"L{li} {cin}->{cout} k{k}d{dil} {H}x{W} N={n}:", flush=True)
    for v in variants:
        med, mn = statistics.median(times[v]), min(times[v])
        print(f"    opt {v[0]:2d} kpc2 {v[1]}: median {med:8.1f} us  min {mn:8.1f} us  {fl / med / 1e6:7.1f} TF/s  {100.0 * (base / med - 1.0):+5.1f} %", flush=True)
lib.slu_h8_dev_set_opt(15, 1)
