#!/usr/bin/env python3
"""On-GPU timing sweep of the flooding kernels on the DVB-S2-profile code.
usage: gpu_sweep.py algo V B iters early_term [reps]   (env LDPC_TUNE_RPW / LDPC_TUNE_CPW honoured)"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes

algo, V, B, iters, et = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), int(sys.argv[5])
reps = int(sys.argv[6]) if len(sys.argv) > 6 else 2
N, K = 64800, 32400
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=iters, frames_per_lane=V, early_term=bool(et), tune=L.capi.tune_from_env())
torch.manual_seed(1)
y = (1.0 + 0.95 * torch.randn(B, N, device="cuda", dtype=torch.float32))
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
timing = os.environ.get("SWEEP_TIMING", "1") == "1"
dec.set_timing(timing)
for rep in range(reps):
    dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, None)
    torch.cuda.synchronize()
st = dec.stats()
it = st["iterations_launched"]
bc, bv = 8 * g.E * B, (8 * g.E + 4 * N) * B
if not timing:
    print("%s V=%d B=%d it=%d et=%d: total %.3f ms -> %.1f us per 64-frame tile-round, %.2f TB/s algorithmic, %.1f Mbit/s info" % (
        algo, V, B, it, et, st["ms_total"], st["ms_total"] * 1e3 / it / ((B + 63) // 64),
        (bc + bv) * it / (st["ms_total"] * 1e-3) / 1e12, B * K / (st["ms_total"] * 1e-3) / 1e6))
    sys.exit(0)
print("%s V=%d B=%d it=%d et=%d rpw=%s cpw=%s: total %.2f ms | check %.3f ms/it (%.2f TB/s) | var %.3f ms/it (%.2f TB/s) | other %.2f ms | both %.2f TB/s algorithmic | conv %d" % (
    algo, V, B, it, et, os.environ.get("LDPC_TUNE_RPW", "-"), os.environ.get("LDPC_TUNE_CPW", "-"), st["ms_total"],
    st["ms_check"] / it, bc / (st["ms_check"] / it * 1e-3) / 1e12, st["ms_var"] / it, bv / (st["ms_var"] / it * 1e-3) / 1e12,
    st["ms_other"], (bc + bv) * it / ((st["ms_check"] + st["ms_var"]) * 1e-3) / 1e12, st["frames_converged"]))
