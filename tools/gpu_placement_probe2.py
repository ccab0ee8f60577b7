#!/usr/bin/env python3
"""Several decoders alive at once (distinct physical memory each): does each have its own speed of the column-fused
check kernel, and does it keep it?  usage: gpu_placement_probe2.py [decoders] [rounds]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes

n = int(sys.argv[1]) if len(sys.argv) > 1 else 5
rounds = int(sys.argv[2]) if len(sys.argv) > 2 else 3
N, K, B = 64800, 32400, 4096
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=20260101)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
decs = [L.Decoder(g, K, max_batch=B, algo="sp", max_iter=50, tune={"link_narrow": False, "link_half": False, "place": 1}) for _ in range(n)]
for r in range(rounds):
    for i, dec in enumerate(decs):
        dec.set_timing(True)
        for _ in range(2):
            dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
        torch.cuda.synchronize()
        kt = {k["name"]: k["ms_total"] / k["launches"] for k in dec.kernel_times() if k["phase"] in (0, 1)}
        ad = dec.array_addresses()
        print("round %d decoder %d: step %.2f ms  %s  Q %#x R %#x" % (r, i, dec.stats()["ms_total"],
              {k.split("<")[0]: round(v, 4) for k, v in kt.items()}, ad["Q"], ad["R"]), flush=True)
for d in decs:
    d.close()
