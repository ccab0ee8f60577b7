#!/usr/bin/env python3
"""Reference point: device-to-device copy rate on this box (read+write bytes / time)."""
import torch
for gb in (1, 4):
    n = gb * (1 << 30) // 4
    x = torch.empty(n, dtype=torch.float32, device="cuda").normal_()
    y = torch.empty_like(x)
    for _ in range(3):
        y.copy_(x)
    torch.cuda.synchronize()
    s, e = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    s.record()
    for _ in range(10):
        y.copy_(x)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("copy %d GiB: %.3f ms -> %.2f TB/s (read+write)" % (gb, ms, 2 * n * 4 / (ms * 1e-3) / 1e12))
    z = torch.empty_like(x)
    s.record()
    for _ in range(10):
        torch.add(x, y, out=z)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("add  %d GiB: %.3f ms -> %.2f TB/s (2 reads + 1 write)" % (gb, ms, 3 * n * 4 / (ms * 1e-3) / 1e12))
    s.record()
    for _ in range(10):
        y.fill_(1.0)
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("fill %d GiB: %.3f ms -> %.2f TB/s (write only)" % (gb, ms, n * 4 / (ms * 1e-3) / 1e12))
    s.record()
    for _ in range(10):
        t = x.sum()
    e.record()
    torch.cuda.synchronize()
    ms = s.elapsed_time(e) / 10
    print("sum  %d GiB: %.3f ms -> %.2f TB/s (read only)" % (gb, ms, n * 4 / (ms * 1e-3) / 1e12))
