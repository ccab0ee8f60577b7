#!/usr/bin/env python3
"""Is some device memory slower than other?  Allocates `n` blocks of `gib` GiB one after the other (all held) and
measures a copy inside each block (first half -> second half, torch's copy kernel, best of 5, read + write counted).
usage: gpu_memory_map.py [blocks] [GiB per block]"""
import sys
import torch
n = int(sys.argv[1]) if len(sys.argv) > 1 else 24
gib = float(sys.argv[2]) if len(sys.argv) > 2 else 4.0
blocks = []
e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
for i in range(n):
    b = torch.empty(int(gib * (1 << 30)), dtype=torch.uint8, device="cuda")
    blocks.append(b)
    half = b.numel() // 2
    src, dst = b[:half].view(torch.float32), b[half:].view(torch.float32)
    src.zero_(); dst.zero_()
    best = 1e9
    for _ in range(5):
        e0.record(); dst.copy_(src); e1.record(); e1.synchronize()
        best = min(best, e0.elapsed_time(e1))
    print("block %2d  at %#x  copy %.3f ms  %.0f GB/s" % (i, b.data_ptr(), best, 2 * half / best / 1e6), flush=True)
