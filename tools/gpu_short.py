#!/usr/bin/env python3
"""Short-code throughput: fused LDS-resident layered decoder vs the streaming layered kernels.
usage: gpu_short.py rate N B sigma"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes
rate, N, B, sigma = int(sys.argv[1]), int(sys.argv[2]), int(sys.argv[3]), float(sys.argv[4])
algo = sys.argv[5] if len(sys.argv) > 5 else "layered"
K, M, z = codes.wimax_dims(rate, N)
rows, cols = codes.wimax_edges(rate, N)
g = L.Graph(rows, cols, M, N)
torch.manual_seed(1)
y = 1.0 + sigma * torch.randn(B, N, device="cuda")
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
it = torch.empty(B, dtype=torch.int32, device="cuda")
for fused in ("1", "ldsp", "0") if algo in ("layered", "ms") else (("1", "ldsp") if algo == "ms_fused" else ("1", "0")):
    os.environ["LDPC_TUNE_FUSED"] = "1" if fused == "ldsp" else fused
    os.environ["LDPC_TUNE_LDSP"] = "1" if fused == "ldsp" else "0"
    dec = L.Decoder(g, K, max_batch=B, algo=algo, layer_rows=z, max_iter=40, poll_interval=0, tune=L.capi.tune_from_env())
    for _ in range(2):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 5
    for _ in range(reps):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = dec.stats()
    print(algo + " rate=%d N=%d z=%d B=%d sigma=%.2f fused=%s: %.3f ms/decode, %.1f Mbit/s info, avg iters %.2f, converged %d/%d" % (
        rate, N, z, B, sigma, fused, dt * 1e3, B * K / dt / 1e6, it.float().mean().item(), st["frames_converged"], B))
    dec.close()
