#!/usr/bin/env python3
"""Is the 'slow state' of the column-fused check kernel a property of the box, of the process or of where the
message arrays lie?  One process: decoder after decoder for the headline workload, with a dummy allocation of a
different size in between (so the arrays move), each timed over 3 steps.  usage: gpu_state_probe.py [rounds]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
N, K, B, ITERS = 64800, 32400, 4096, 50
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 8
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=2026, device=0)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
pads = [0, 1 << 20, 3 << 20, 64 << 10, 7 << 20, 0, 513 << 10, 129 << 20, 0, 33 << 20, 2 << 20, 0]
keep = []
for r in range(rounds):
    pad = pads[r % len(pads)]
    if pad:
        keep.append(torch.empty(pad, dtype=torch.uint8, device="cuda"))
    dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=ITERS, llr_scale=8.0, early_term=True)
    s = torch.cuda.current_stream().cuda_stream
    dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
    torch.cuda.synchronize()
    dec.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(3):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 3
    kt = {k["name"]: round(k["ms_total"] / k["launches"], 4) for k in dec.kernel_times() if k["phase"] in (0, 1)}
    print("round %d pad %9d B: %.1f ms/step  %s" % (r, pad, dt * 1e3, kt), flush=True)
    dec.close()
