#!/usr/bin/env python3
"""BG1-profile layered decoding at a given lifting size: layered_ldsp_kernel (posterior in LDS,
check records in cache) against the one-launch-per-layer streaming kernels.
usage: gpu_ldsp.py Z B sigma [iters] [early_term] [modes] [algo]      modes: comma list of ldsp,stream; algo: layered | ms"""
import sys, os, time
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes
Z, B, sigma = int(sys.argv[1]), int(sys.argv[2]), float(sys.argv[3])
iters = int(sys.argv[4]) if len(sys.argv) > 4 else 20
early = int(sys.argv[5]) if len(sys.argv) > 5 else 1
modes = sys.argv[6].split(",") if len(sys.argv) > 6 else ["ldsp", "stream"]
algo = sys.argv[7] if len(sys.argv) > 7 else "layered"
rows, cols = codes.nr_bg1_profile_edges(Z)
N, K, M = 68 * Z, 22 * Z, 46 * Z
g = L.Graph(rows, cols, M, N)
torch.manual_seed(1)
y = 1.0 + sigma * torch.randn(B, N, device="cuda")
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
it = torch.empty(B, dtype=torch.int32, device="cuda")
ref = None
for mode in modes:
    os.environ["LDPC_TUNE_LDSP"] = "1" if mode == "ldsp" else "0"
    os.environ["LDPC_TUNE_FUSED"] = "1" if mode == "ldsp" else "0"
    dec = L.Decoder(g, K, max_batch=B, algo=algo, layer_rows=Z, max_iter=iters, early_term=early, tune=L.capi.tune_from_env())
    for _ in range(2):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    reps = 3
    for _ in range(reps):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / reps
    st = dec.stats()
    avg = it.float().mean().item()
    print(algo + " Z=%d N=%d E=%d B=%d sigma=%.2f %s: %.3f ms, %.1f Mbit/s info, avg iters %.2f, converged %d/%d, %.1f G edge-iterations/s" % (
        Z, N, len(rows), B, sigma, mode, dt * 1e3, B * K / dt / 1e6, avg, st["frames_converged"], B,
        B * avg * len(rows) / dt / 1e9), flush=True)
    res = (out.clone(), it.clone())
    if ref is not None:
        print("  same bytes:", bool((res[0] == ref[0]).all()), " same iteration counts:", bool((res[1] == ref[1]).all()))
    ref = res
    dec.close()
