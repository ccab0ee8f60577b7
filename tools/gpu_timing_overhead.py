#!/usr/bin/env python3
"""Cost of the per-launch HIP events bench.py keeps on during its timed region: the headline
step with set_timing(True) vs set_timing(False), alternating on one box."""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
N, K, B = 64800, 32400, 4096
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=50)
y = channel.awgn_device(N, 0, B, 0.95)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
for _ in range(2):
    dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
torch.cuda.synchronize()
for rep in range(3):
    for timing in (True, False):
        dec.set_timing(timing)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(4):
            dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
        torch.cuda.synchronize()
        dt = (time.perf_counter() - t0) / 4
        print("timing=%s: %.3f ms/step, %.1f Mbit/s" % (timing, dt * 1e3, B * K / dt / 1e6), flush=True)
