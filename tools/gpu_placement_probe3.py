#!/usr/bin/env python3
"""How are the speeds distributed over smaller allocations?  `n` decoders of `B` frames each alive at once (no placement
search), the check kernel of each timed.  usage: gpu_placement_probe3.py [n] [B]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes
n = int(sys.argv[1]) if len(sys.argv) > 1 else 16
B = int(sys.argv[2]) if len(sys.argv) > 2 else 1024
N, K = 64800, 32400
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=20260101)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
decs = [L.Decoder(g, K, max_batch=B, algo="sp", max_iter=20, tune={"link_narrow": False, "link_half": False, "place": 1}) for _ in range(n)]
res = []
for r in range(2):
    row = []
    for dec in decs:
        dec.set_timing(True)
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
        torch.cuda.synchronize()
        kt = {k["name"].split("<")[0]: k["ms_total"] / k["launches"] for k in dec.kernel_times() if k["phase"] in (0, 1)}
        row.append(kt["check_link_kernel"])
    res.append(row)
    print("pass %d check ms:" % r, " ".join("%.4f" % v for v in row), flush=True)
