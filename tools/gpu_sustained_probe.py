#!/usr/bin/env python3
"""Burst and sustained copy rate next to the headline step, alternating, on one box: does the 'slow state'
of the step coincide with a lower sustained copy rate?  usage: gpu_sustained_probe.py [rounds]"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
N, K, B, ITERS = 64800, 32400, 4096, 50
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 6
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=2026, device=0)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=ITERS, llr_scale=8.0, early_term=True)
s = torch.cuda.current_stream().cuda_stream
dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
torch.cuda.synchronize()
for r in range(rounds):
    dec.set_timing(True)
    t0 = time.perf_counter()
    for _ in range(4):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / 4
    kt = {k["name"]: round(k["ms_total"] / k["launches"], 4) for k in dec.kernel_times() if k["phase"] in (0, 1)}
    sus = L.capi.hbm_sustained(0, 1 << 30, 300)
    burst = L.capi.hbm_probe(0, 1 << 30, 5)
    sus2 = L.capi.hbm_sustained(0, 1 << 30, 1500)
    print("round %d: step %.1f ms %s | copy: sustained 0.3 s %.0f GB/s, burst %.0f, sustained 1.5 s %.0f" % (r, dt * 1e3, kt, sus, burst, sus2), flush=True)
    time.sleep(2.0 if r % 2 else 0.0)
