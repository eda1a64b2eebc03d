import torch, time
for mb in (5, 32, 137):
    h = torch.empty(mb*1024*1024, dtype=torch.uint8).pin_memory()
    d = torch.empty_like(h, device='cuda')
    for _ in range(3): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); t=time.perf_counter()
    n=10
    for _ in range(n): d.copy_(h, non_blocking=True)
    torch.cuda.synchronize(); dt=(time.perf_counter()-t)/n
    print(mb, 'MB', round(dt*1e3,3), 'ms', round(mb/1024/dt,1), 'GiB/s')
