import os, sys, torch
sys.path.insert(0, "/root/repo")
from semanticlidarunc_amd import ops
from semanticlidarunc_amd.ops import ConvSource
dev = torch.device("cuda:0")
n = 4
LAYERS = [(32, 32, 3, 1, 1, 64, 2048), (32, 32, 3, 2, 2, 64, 2048), (32, 32, 1, 1, 0, 64, 2048), (96, 32, 1, 1, 0, 64, 2048), (32, 64, 3, 1, 1, 64, 2048),
          (64, 32, 3, 1, 1, 64, 2048), (64, 64, 3, 2, 2, 64, 2048), (64, 64, 2, 2, 1, 64, 2048), (192, 64, 1, 1, 0, 64, 2048), (64, 128, 3, 1, 1, 32, 1024),
          (128, 128, 3, 2, 2, 32, 1024), (256, 256, 3, 2, 2, 16, 512), (256, 256, 3, 1, 1, 8, 256)]
for li, (cin, cout, k, dil, pad, H, W) in enumerate(LAYERS):
    g = torch.Generator(device=dev).manual_seed(li)
    x = torch.randn(n, cin, H, W, device=dev, generator=g)
    w = ops.pack_conv_weight(torch.randn(cout, cin, k, k, device=dev, generator=g) / (cin * k * k) ** 0.5)
    bias = torch.zeros(cout, device=dev)
    run = lambda: ops.conv2d_fused([ConvSource(x)], w, cout, k, dil, pad, bias=bias, slope=0.01)
    for _ in range(3):
        run()
    torch.cuda.synchronize()
    ts = []
    for _ in range(7):
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record(); run(); e1.record(); torch.cuda.synchronize()
        ts.append(e0.elapsed_time(e1) * 1e3)
    ts.sort()
    fl = 2.0 * n * H * W * cin * cout * k * k
    print(f"{cin:4d}->{cout:3d} k{k}d{dil} {H}x{W}: {ts[3]:7.1f} us {fl / ts[3] / 1e6:6.1f} TF/s", flush=True)
