#!/usr/bin/env python3
"""Quick on-GPU sanity run: parity vs the oracle on small codes, then a timing
sweep on the DVB-S2-profile code.  Development aid, not a test."""
import sys, time, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
import oracle


def parity(rate, N, sigma, B, algo, V, max_iter=40, tap=0, seed=1):
    K, M, z = codes.wimax_dims(rate, N)
    rows, cols = codes.wimax_edges(rate, N)
    g = L.Graph(rows, cols, M, N)
    og = oracle.Graph(rows, cols, M, N, K)
    y = channel.awgn_frames(N, 0, B, sigma, seed)
    dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=max_iter, frames_per_lane=V, layer_rows=z)
    out, iters = dec.decode(y)
    o = oracle.decode(og, y, algo, max_iter=max_iter, layer_rows=z)
    ok = np.array_equal(out, o["out"]) and np.array_equal(iters, o["iters"])
    st = dec.stats()
    print("parity rate=%d N=%d %s V=%d B=%d sigma=%.2f: %s  iters=%s conv=%d/%d" % (
        rate, N, algo, V, B, sigma, "OK" if ok else "MISMATCH", o["iters"][:8].tolist(), st["frames_converged"], B))
    if not ok:
        print("   out diff bytes:", int(np.count_nonzero(out != o["out"])), "iters gpu", iters[:8].tolist())
    dec.close()
    return ok


def main():
    print("devices:", L.device_count())
    allok = True
    for algo in ("ms", "sp", "layered"):
        for V in (1, 2, 4):
            allok &= parity(0, 648, 0.75, 9, algo, V)
            allok &= parity(4, 576, 0.5, 70, algo, V)
            allok &= parity(0, 2304, 0.9, 5, algo, V)
    print("ALL PARITY OK" if allok else "PARITY FAILURES")
    if "--bench" in sys.argv:
        import torch
        N, K = 64800, 32400
        rows, cols = codes.dvbs2_profile_edges(N, K)
        g = L.Graph(rows, cols, N - K, N)
        for algo in ("sp", "ms"):
            for V in (1, 2, 4):
                B = 1024
                dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=10, frames_per_lane=V, early_term=True)
                y = (1.0 + 0.95 * torch.randn(B, N, device="cuda", dtype=torch.float32))
                out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
                dec.set_timing(True)
                for rep in range(2):
                    dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), None, None)
                    torch.cuda.synchronize()
                st = dec.stats()
                bytes_iter = (16 * g.E + 4 * N) * B
                it = st["iterations_launched"]
                print("bench %s V=%d B=%d: total %.2f ms, check %.2f ms, var %.2f ms, other %.2f; per-iter %.3f ms -> %.2f TB/s algorithmic; conv %d" % (
                    algo, V, B, st["ms_total"], st["ms_check"], st["ms_var"], st["ms_other"],
                    (st["ms_check"] + st["ms_var"]) / it, bytes_iter * it / ((st["ms_check"] + st["ms_var"]) * 1e-3) / 1e12, st["frames_converged"]))
                dec.close()
                del y, out
    return 0 if allok else 1


if __name__ == "__main__":
    sys.exit(main())
