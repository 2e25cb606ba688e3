"""Practical HBM ceiling of the box: device-to-device copy and a read-only reduction through torch (1 GiB operands)."""
import torch
n = 1 << 27          # 1 GiB of fp64
x = torch.ones(n, dtype=torch.float64, device="cuda")
y = torch.empty_like(x)
def timeit(fn, reps=20):
    for _ in range(3): fn()
    torch.cuda.synchronize()
    e0 = torch.cuda.Event(enable_timing=True); e1 = torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(reps): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / reps * 1e-3
t = timeit(lambda: y.copy_(x))
print("copy   : %.1f GB/s (read + write)" % (2 * 8 * n / t / 1e9))
t = timeit(lambda: x.sum())
print("sum    : %.1f GB/s (read only)" % (8 * n / t / 1e9))
t = timeit(lambda: torch.add(x, y, out=y))
print("axpy   : %.1f GB/s (2 reads + 1 write)" % (3 * 8 * n / t / 1e9))
