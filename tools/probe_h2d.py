"""Host -> device copy rate of this box: pageable vs page-locked, one big copy vs many small ones."""
import time

import torch

for mb in (4, 64, 512):
    n = mb << 20
    pinned = torch.empty(n, dtype=torch.uint8).pin_memory()
    pageable = torch.empty(n, dtype=torch.uint8)
    dev = torch.empty(n, dtype=torch.uint8, device="cuda")
    for name, src in (("pinned", pinned), ("pageable", pageable)):
        dev.copy_(src, non_blocking=True); torch.cuda.synchronize()
        t0 = time.perf_counter()
        reps = max(1, 1024 // mb)
        for _ in range(reps):
            dev.copy_(src, non_blocking=True)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / reps
        print("H2D %4d MB %-8s %.2f GB/s" % (mb, name, n / dt / 1e9), flush=True)
    host = torch.empty(n, dtype=torch.uint8).pin_memory()
    host.copy_(dev, non_blocking=True); torch.cuda.synchronize()
    t0 = time.perf_counter(); host.copy_(dev, non_blocking=True); torch.cuda.synchronize()
    print("D2H %4d MB pinned   %.2f GB/s" % (mb, n / (time.perf_counter() - t0) / 1e9), flush=True)
