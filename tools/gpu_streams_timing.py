import sys, os, time
sys.path.insert(0, "/root/repo")
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
N, K, B, ITERS = 64800, 32400, 4096, 50
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
y = channel.awgn_device(N, 0, B, 0.95, seed=2026, device=0)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
for streams in (0, 2):
    dec = L.Decoder(g, K, max_batch=B, algo="sp", max_iter=ITERS, llr_scale=8.0, early_term=True, streams=streams)
    s = torch.cuda.current_stream().cuda_stream
    for _ in range(2):
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
    torch.cuda.synchronize()
    for every in (0, 5, 1):
        dec.set_timing(every)
        torch.cuda.synchronize()
        ts = []
        for _ in range(10):
            t0 = time.perf_counter()
            dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, s)
            torch.cuda.synchronize()
            ts.append((time.perf_counter() - t0) * 1e3)
        print("streams", streams, "timing every", every, ["%.1f" % t for t in ts], flush=True)
    dec.close()
