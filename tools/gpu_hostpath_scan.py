#!/usr/bin/env python3
"""Host-buffer path (ldpc_decode, the reference's Coder::decode signature) by number of launch groups:
T(groups) for 1..5 groups of 4096 frames of the (64800, 32400) code from pageable memory, 50 sum-product
iterations at full work: the slope is a group's cost inside the pipeline, the intercept what cannot overlap
(first copy in, page locking, last copy out).  Also a pinned (torch) input for comparison.
usage: gpu_hostpath_scan.py [max groups] [staged|lock_pages] [copy threads]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes
N, K = 64800, 32400
B, iters = 4096, 50
maxg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
mode = sys.argv[2] if len(sys.argv) > 2 else "staged"        # ldpc_decoder_config.host_input
threads = int(sys.argv[3]) if len(sys.argv) > 3 else 0       # ldpc_decoder_config.host_copy_threads (0 = 4)
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
rng = np.random.default_rng(1)
y = (1.0 + 0.95 * rng.standard_normal((B * maxg, N), dtype=np.float32)).astype(np.float32)
dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=iters, host_input=mode, host_copy_threads=threads)
print("host_input =", mode, "copy threads =", threads or 4)
dec.decode(y[:B], want_iters=False)                      # warm-up (allocations, code load)
yp = torch.from_numpy(y).pin_memory().numpy()
for name, src in (("pageable", y), ("caller-locked", yp)):
    for groups in range(1, maxg + 1):
        best = 1e9
        for rep in range(3):
            t0 = time.perf_counter()
            dec.decode(src[:groups * B], want_iters=False)
            best = min(best, time.perf_counter() - t0)
        print("%-13s %d x %d frames: %7.1f ms  %7.1f Mbit/s info" % (name, groups, B, best * 1e3, groups * B * K / best / 1e6), flush=True)
