"""Development: what HBM write rate does a plain streaming fill reach on this box?  (the training kernels' u / h stores are compared with it)"""
import torch

for mb in (84, 336, 1344):
    n = mb * 1024 * 1024 // 2
    x = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    y = torch.empty(n, dtype=torch.bfloat16, device="cuda")
    for name, fn, bytes_ in (("fill", lambda: x.zero_(), 2 * n), ("copy", lambda: y.copy_(x), 4 * n)):
        for _ in range(3):
            fn()
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(20):
            fn()
        e1.record()
        torch.cuda.synchronize()
        us = e0.elapsed_time(e1) * 1000 / 20
        print("%4d MB %s: %7.1f us  %6.2f TB/s" % (mb, name, us, bytes_ / us / 1e6))
