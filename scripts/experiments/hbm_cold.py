"""What a cold read stream gets on this chip: reductions / copies over buffers far larger than the 256 MB Infinity Cache
(each timed launch touches bytes no earlier launch of the loop left in any cache)."""
import torch
dev = torch.device("cuda:0")
def t(fn, n=5):
    fn(); torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    for _ in range(n): fn()
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1) / n * 1e-3
for gb in (0.1, 0.5, 2, 8):
    n = int(gb * 1e9 / 4)
    x = torch.empty(n, dtype=torch.float32, device=dev).normal_()
    s = t(lambda: x.sum())
    y = torch.empty_like(x)
    c = t(lambda: y.copy_(x))
    m = t(lambda: x.view(torch.int32).max())
    print("%5.1f GB: sum %7.1f us = %5.0f GB/s | max(int32) %7.1f us = %5.0f GB/s | copy %7.1f us = %5.0f GB/s (read + write)" % (gb, s * 1e6, gb * 1e9 / s / 1e9, m * 1e6, gb * 1e9 / m / 1e9, c * 1e6, 2 * gb * 1e9 / c / 1e9), flush=True)
    del x, y
