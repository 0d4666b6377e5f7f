"""H2D copy rate from pinned memory on a side stream, per size (probe for frame_loader.py)."""
import time
import torch
h = torch.empty((8192 * 50, 3), dtype=torch.float32).pin_memory()
d = torch.empty_like(h, device="cuda")
s = torch.cuda.Stream()
with torch.cuda.stream(s):
    d.copy_(h, non_blocking=True)
torch.cuda.synchronize()
for rep in range(2):
    for n in (1000, 20000, 100000, 200000, 400000):
        torch.cuda.synchronize(); t0 = time.perf_counter()
        with torch.cuda.stream(s):
            for _ in range(10):
                d[:n].copy_(h[:n], non_blocking=True)
        t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
        print("n=%6d (%.2f MB): enqueue %.3f ms, total %.3f ms per copy (%.1f GB/s)" % (n, n * 12 / 1e6, (t1 - t0) * 100, (t2 - t0) * 100, n * 12 / ((t2 - t0) / 10) / 1e9))
# the same while a long kernel occupies another stream
x = torch.randn(8192, 8192, device="cuda")
for n in (100000, 400000):
    torch.cuda.synchronize()
    for _ in range(20):
        y = x @ x
    t0 = time.perf_counter()
    with torch.cuda.stream(s):
        for _ in range(10):
            d[:n].copy_(h[:n], non_blocking=True)
    t1 = time.perf_counter(); s.synchronize(); t2 = time.perf_counter(); torch.cuda.synchronize(); t3 = time.perf_counter()
    print("busy GPU, n=%6d: enqueue %.3f ms per copy, copies done after %.2f ms, matmuls after %.2f ms" % (n, (t1 - t0) * 100, (t2 - t0) * 1e3, (t3 - t0) * 1e3))
