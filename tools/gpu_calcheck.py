import sys, os
sys.path.insert(0, "/root/repo")
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes
N, K = 64800, 32400
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
for B in (256, 512, 1024, 4096):
    for rep in range(3):
        dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=10, frames_per_lane=4)
        y = (1.0 + 0.95 * torch.randn(B, N, device="cuda", dtype=torch.float32))
        out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
        dec.set_timing(True)
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, None)
        torch.cuda.synchronize()
        kt = [k for k in dec.kernel_times() if "link" in k["name"]]
        print(B, rep, [(k["name"], round(k["ms_total"] / k["launches"], 4)) for k in kt], flush=True)
        dec.close()
