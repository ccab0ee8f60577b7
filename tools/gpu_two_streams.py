#!/usr/bin/env python3
"""Do launch boundaries cost anything?  The headline batch (4096 frames of the (64800, 32400) code, 50
sum-product iterations at full work) decoded by ONE decoder on one stream against the same frames split
over k decoders of 4096/k frames on k streams (asynchronous calls issued back to back, so the GPU can
fill one decoder's draining kernel with the other's blocks).  usage: gpu_two_streams.py [reps]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
N, K, B, ITERS = 64800, 32400, 4096, 50
reps = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=2026, device=0)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
it = torch.empty(B, dtype=torch.int32, device="cuda")
ref = None
for k in (1, 2, 4, 1, 2, 4):
    b = B // k
    decs = [L.Decoder(g, K, max_batch=b, algo="sp", max_iter=ITERS, llr_scale=8.0, early_term=True, poll_interval=0) for _ in range(k)]
    streams = [torch.cuda.Stream() for _ in range(k)]
    def run():
        for i, (d, s) in enumerate(zip(decs, streams)):
            d.decode_device(y[i * b:(i + 1) * b].data_ptr(), b, out[i * b * K // 8:].data_ptr(), b * K // 8,
                            it[i * b:].data_ptr(), s.cuda_stream)
    run(); torch.cuda.synchronize()
    best = 1e9
    for _ in range(reps):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        run()
        torch.cuda.synchronize()
        best = min(best, time.perf_counter() - t0)
    o = out.cpu().numpy().copy()
    if ref is None:
        ref = o
    print("%d decoder(s) x %4d frames on %d stream(s): %7.2f ms  %7.1f Mbit/s  same bytes: %s" % (k, b, k, best * 1e3, B * K / best / 1e6, np.array_equal(o, ref)), flush=True)
    for d in decs:
        d.close()
    del decs
    torch.cuda.empty_cache()
