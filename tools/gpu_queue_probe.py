#!/usr/bin/env python3
"""Does the step time depend on the STREAM (i.e. on the hardware queue behind it)?  One decoder, the headline step
on the default stream and on eight more streams in turn, three passes.  usage: gpu_queue_probe.py"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
N, K, B, ITERS = 64800, 32400, 4096, 50
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=2026, device=0)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=ITERS, llr_scale=8.0, early_term=True)
streams = [None] + [torch.cuda.Stream() for _ in range(8)]
torch.cuda.synchronize()
for p in range(3):
    row = []
    for st in streams:
        s = torch.cuda.current_stream().cuda_stream if st is None else st.cuda_stream
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(2):
            dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
        torch.cuda.synchronize()
        row.append((time.perf_counter() - t0) / 2 * 1e3)
    print("pass %d: default %.1f | streams %s" % (p, row[0], " ".join("%.1f" % t for t in row[1:])), flush=True)
