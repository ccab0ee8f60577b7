#!/usr/bin/env python3
"""Does the time of the column-fused check kernel depend on WHERE the decoder's arrays land in device memory?
One process: decoders are created one after the other with dummy allocations of varying size kept alive in
between (so every decoder's Q / R / chan arrays get different physical and virtual places), each runs a few
steps with per-launch timing; the decoder's own placement search is off (place = 1).
usage: gpu_placement_probe.py [trials] [nopads]      (nopads: no dummy allocations -- identical virtual addresses every time)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes

trials = int(sys.argv[1]) if len(sys.argv) > 1 else 10
nopads = len(sys.argv) > 2
N, K, B = 64800, 32400, 4096
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=20260101)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
s = torch.cuda.current_stream().cuda_stream
pads = []
for t in range(trials):
    dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=50, tune={"link_narrow": False, "link_half": False, "place": 1})
    dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
    torch.cuda.synchronize()
    dec.set_timing(True)
    for _ in range(2):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
    torch.cuda.synchronize()
    kt = {k["name"]: k["ms_total"] / k["launches"] for k in dec.kernel_times() if k["phase"] in (0, 1)}
    st = dec.stats()
    ad = dec.array_addresses()
    print("trial %2d  pads held %5.0f MB  step %.2f ms  check %.4f  Q %#x R %#x chan %#x  R-Q %#x  chan-Q %#x" % (
        t, sum(p.numel() for p in pads) / 1e6, st["ms_total"], [v for k, v in kt.items() if "check" in k][0],
        ad["Q"], ad["R"], ad["chan"], ad["R"] - ad["Q"], ad["chan"] - ad["Q"]), flush=True)
    dec.close()
    # perturb the allocator: keep an odd-sized block alive so that the next decoder's arrays land elsewhere
    if nopads:
        continue
    pads.append(torch.empty((37 + 61 * t) * (1 << 20) + 4096 * (t + 1), dtype=torch.uint8, device="cuda"))
