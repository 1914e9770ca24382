"""Achievable streaming HBM rate on this box: device-to-device copy of 8 GiB (read + write counted) and a read-only reduction."""
import time
import torch

dev = torch.device("cuda:0")
n = 8 << 30
a = torch.empty(n, dtype=torch.uint8, device=dev).random_(0, 255)
b = torch.empty_like(a)
for name, fn, bytes_moved in (("copy (read+write)", lambda: b.copy_(a), 2 * n), ("read-only sum", lambda: a.view(torch.int64).sum(), n)):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(10):
        fn()
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 10
    print("%s: %.2f TB/s (%.2f ms for %d GiB)" % (name, bytes_moved / dt / 1e12, dt * 1e3, n >> 30), flush=True)
