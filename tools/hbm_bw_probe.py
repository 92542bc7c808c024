import torch, time
dev=torch.device('cuda:0')
def t(f, n=20):
    for _ in range(3): f()
    torch.cuda.synchronize(); e0=torch.cuda.Event(enable_timing=True); e1=torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): f()
    e1.record(); torch.cuda.synchronize(); return e0.elapsed_time(e1)/n
for mb in (268, 537, 1074):
    a=torch.empty(mb*1000*1000//4, dtype=torch.float32, device=dev); b=torch.empty_like(a)
    ms=t(lambda: a.fill_(1.0)); print(f"fill  {mb} MB: {ms*1e3:.1f} us  {mb/ms/1e3:.2f} TB/s")
    ms=t(lambda: b.copy_(a)); print(f"copy  {mb} MB: {ms*1e3:.1f} us  {2*mb/ms/1e3:.2f} TB/s (r+w)")
    ms=t(lambda: a.sum()); print(f"sum   {mb} MB: {ms*1e3:.1f} us  {mb/ms/1e3:.2f} TB/s")
