#!/usr/bin/env python3
"""Host-buffer path (the reference's Coder::decode signature): PCIe-inclusive throughput of
ldpc_decode on pageable host memory, one multi-group call (H2D of group k+1 overlaps the decode
of group k) against the same groups decoded by separate calls (no overlap)."""
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
y = (1.0 + 0.95 * rng.standard_normal((B * groups, N), dtype=np.float32)).astype(np.float32)
dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=iters)
dec.decode(y[:B], want_iters=False)                      # warm-up (allocations, code load)
bits = B * groups * K
for rep in range(3):
    t0 = time.perf_counter()
    outs = [dec.decode(y[i * B:(i + 1) * B], want_iters=False)[0] for i in range(groups)]
    t2 = time.perf_counter() - t0
    t0 = time.perf_counter()
    out1, _ = dec.decode(y, want_iters=False)
    t1 = time.perf_counter() - t0
    assert np.array_equal(out1, np.concatenate(outs))
    print("host path, %d x %d frames, %d iterations: one call %.1f ms (%.1f Mbit/s info, PCIe-inclusive); "
          "separate calls %.1f ms (%.1f Mbit/s)" % (groups, B, iters, t1 * 1e3, bits / t1 / 1e6, t2 * 1e3, bits / t2 / 1e6))
