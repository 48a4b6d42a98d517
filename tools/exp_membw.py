"""Plain fill / copy rates of this box (torch kernels), for reading the store-bound kernels' numbers against."""
import torch
dev = "cuda:0"
for mb in (236, 944):
    n = mb * 1024 * 1024 // 4
    a = torch.empty(n, device=dev); b = torch.empty(n, device=dev)
    for name, fn, bytes_ in (("fill", lambda: a.fill_(1.0), n * 4), ("copy", lambda: b.copy_(a), 2 * n * 4), ("read(sum)", lambda: a.sum(), n * 4)):
        for _ in range(3): fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10): fn()
        e1.record(); torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 100
        print(f"{mb} MB {name}: {us:.1f} us  {bytes_ / us / 1e6:.2f} TB/s")
