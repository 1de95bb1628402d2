"""Achievable HBM bandwidth on this GPU (context for the 8 TB/s spec peak used in the roofline fractions)."""
import torch
for name, fn, factor in [("copy (read + write)", lambda a, b: b.copy_(a), 2), ("fill (write)", lambda a, b: b.fill_(1.0), 1),
                         ("sum (read)", lambda a, b: a.sum(), 1)]:
    for gb in (1, 4):
        n = gb * (1 << 30) // 4
        a, b = torch.ones(n, device="cuda"), torch.empty(n, device="cuda")
        for _ in range(3):
            fn(a, b)
        torch.cuda.synchronize()
        e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
        e0.record()
        for _ in range(10):
            fn(a, b)
        e1.record()
        torch.cuda.synchronize()
        ms = e0.elapsed_time(e1) / 10
        print("%-22s %d GiB: %7.1f GB/s" % (name, gb, factor * n * 4 / ms * 1e-6))
        del a, b
