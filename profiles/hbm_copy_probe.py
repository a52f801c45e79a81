import torch
for mb in (100, 200, 800):
    n = mb * 2**20 // 4
    a = torch.randn(n, device="cuda"); b = torch.empty_like(a)
    for _ in range(3): b.copy_(a)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(20): b.copy_(a)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print("copy %4d MB: %.1f us  %.0f GB/s (read+write)" % (mb, t * 1e6, 2 * n * 4 / t / 1e9))
    s.record()
    for _ in range(20): a.mul_(1.0001)
    e.record(); torch.cuda.synchronize()
    t = s.elapsed_time(e) / 20 * 1e-3
    print("inplace scale %4d MB: %.1f us  %.0f GB/s" % (mb, t * 1e6, 2 * n * 4 / t / 1e9))
