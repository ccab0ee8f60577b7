#!/usr/bin/env python3
"""Does a pageable H2D copy on one stream overlap kernels running on another stream?"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np, torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes
N, K, B = 64800, 32400, 4096
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=50)
y = 1.0 + 0.95 * torch.randn(B, N, device="cuda")
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
host = torch.randn(B, N)                       # pageable
pinned = torch.randn(B, N).pin_memory()
dst = torch.empty(B, N, device="cuda")
s1, s2 = torch.cuda.Stream(), torch.cuda.Stream()
def run(src, label):
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s1.cuda_stream)
    t_launch = time.perf_counter() - t0
    with torch.cuda.stream(s2):
        dst.copy_(src, non_blocking=True)
    s2.synchronize()
    t_copy = time.perf_counter() - t0
    s1.synchronize()
    t_all = time.perf_counter() - t0
    print("%s: launch returned after %.1f ms, copy done at %.1f ms, decode done at %.1f ms" % (label, t_launch * 1e3, t_copy * 1e3, t_all * 1e3))
for _ in range(2):
    run(host, "pageable")
    run(pinned, "pinned  ")
torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(host); torch.cuda.synchronize(); print("pageable copy alone %.1f ms" % ((time.perf_counter() - t0) * 1e3))
torch.cuda.synchronize(); t0 = time.perf_counter(); dst.copy_(pinned); torch.cuda.synchronize(); print("pinned copy alone %.1f ms" % ((time.perf_counter() - t0) * 1e3))
