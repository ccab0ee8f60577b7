#!/usr/bin/env python3
"""Time of the input transpose (init_kernel) and of the output packing at the headline shape (GPU only)."""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
N, K, B = 64800, 32400, 4096
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=1)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
for algo, dt in (("sp", "f32"), ("ms", "f32"), ("ms", "f16")):
    dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=3, msg_dtype=dt, tune={"place": 1})
    dec.set_timing(True)
    for _ in range(4):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, None)
    torch.cuda.synchronize()
    for k in dec.kernel_times():
        if k["phase"] not in (0, 1, 4, 5, 6):
            print(algo, dt, k["name"], "%.4f ms" % (k["ms_total"] / max(k["launches"], 1)), k["launches"])
    dec.close()
