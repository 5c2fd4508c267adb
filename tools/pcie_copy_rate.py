import torch, time
for mb in (4.6, 1.0):
    n = int(mb * (1 << 20))
    d = torch.empty(n, dtype=torch.uint8, device="cuda:0"); h = torch.empty(n, dtype=torch.uint8).pin_memory()
    for _ in range(5): h.copy_(d, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): h.copy_(d, non_blocking=True); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 50
    print("D2H", mb, "MiB:", t * 1e6, "us", n / t / 1e9, "GB/s")
    for _ in range(5): d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(50): d.copy_(h, non_blocking=True); torch.cuda.synchronize()
    t = (time.perf_counter() - t0) / 50
    print("H2D", mb, "MiB:", t * 1e6, "us", n / t / 1e9, "GB/s")
