import torch, time
n = 192*1024*1024
x = torch.empty(n, dtype=torch.uint8, device='cuda')
y = torch.empty(n, dtype=torch.uint8).pin_memory()
for d in ("d2h","h2d"):
    for _ in range(3):
        torch.cuda.synchronize(); t0=time.perf_counter()
        if d=="d2h": y.copy_(x, non_blocking=True)
        else: x.copy_(y, non_blocking=True)
        torch.cuda.synchronize(); dt=time.perf_counter()-t0
    print(d, "%.1f GB/s"%(n/dt/1e9))
