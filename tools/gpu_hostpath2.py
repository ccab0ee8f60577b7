#!/usr/bin/env python3
"""Host-buffer path with early termination at a practical noise level: one multi-group ldpc_decode
call (the reference's Coder::decode) with host polling (tail compaction applies per group) against
the same call without polling.  Min-sum on the (64800, 32400) profile code at 2 dB."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import myldpccppapi_amd as L
from myldpccppapi_amd import codes
N, K = 64800, 32400
B, groups, iters = 4096, 3, 50
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
rng = np.random.default_rng(1)
sd = 10 ** (-2.0 / 20)
y = (1.0 + sd * rng.standard_normal((B * groups, N), dtype=np.float32)).astype(np.float32)
bits = B * groups * K
ref = None
for poll in (0, 2):
    dec = L.Decoder(g, K, max_batch=B, algo="ms", max_iter=iters, poll_interval=poll)
    dec.decode(y[:B], want_iters=False)
    for rep in range(2):
        t0 = time.perf_counter()
        out, it = dec.decode(y)
        t1 = time.perf_counter() - t0
    if ref is None:
        ref = (out, it)
    same = np.array_equal(out, ref[0]) and np.array_equal(it, ref[1])
    print("poll_interval=%d: %.1f ms for %d x %d frames (%.1f Mbit/s info, PCIe-inclusive), avg iters %.2f, same results: %s"
          % (poll, t1 * 1e3, groups, B, bits / t1 / 1e6, it.mean(), same), flush=True)
    dec.close()
