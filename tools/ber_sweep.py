#!/usr/bin/env python3
"""BER / FER versus SNR on the GPU, in the reference's convention (Test.cpp:56-57):
BPSK +-1, sd = 10^(-SNR_dB/20), all-zero codeword (valid for every linear code), channel values
generated in HBM by ldpc_awgn_device (counter-based noise, csrc/ldpc_channel.h: frame f of batch
b is frame b*frames + f of the seed's stream, so a point can be extended or re-run on any rank)
and errors counted by ldpc_count_errors_device; nothing crosses PCIe but the counts.  The
reference counts differing BYTES (Test.cpp:105-110); this prints byte errors too.

    python tools/ber_sweep.py [--code dvbs2_12|dvbs2_910|bg1|wimax:<rate>:<N>] [--algo sp|ms|layered]
                              [--snr=1.0,1.5,...]  (write --snr=-0.5,0 for a list that starts with a minus) [--frames 4096] [--iters 50]
The DVB-S2 / BG1 codes are PROFILE SURROGATES (codes.py): the numbers are not the standards'."""
import argparse, json, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes

ap = argparse.ArgumentParser()
ap.add_argument("--code", default="dvbs2_12")
ap.add_argument("--algo", default="sp")
ap.add_argument("--snr", default="2.5,3.0,3.5,4.0,4.5,5.0")
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--iters", type=int, default=50)
ap.add_argument("--llr-scale", type=float, default=8.0)
ap.add_argument("--batches", type=int, default=1, help="batches of --frames per SNR point")
ap.add_argument("--seed", type=int, default=20260101)
args = ap.parse_args()

layer = 0
if args.code == "dvbs2_12":
    N, K = 64800, 32400
    rows, cols = codes.dvbs2_profile_edges(N, K)
elif args.code == "dvbs2_910":
    N, K = 64800, 58320
    rows, cols = codes.dvbs2_profile_edges(N, K)
elif args.code == "bg1":
    Z = 384
    N, K, layer = 68 * Z, 22 * Z, Z
    rows, cols = codes.nr_bg1_profile_edges(Z)
else:
    _, rate, N = args.code.split(":")
    rate, N = int(rate), int(N)
    K, M, layer = codes.wimax_dims(rate, N)
    rows, cols = codes.wimax_edges(rate, N)
M = N - K
g = L.Graph(rows, cols, M, N)
B = args.frames
dec = L.Decoder(g, K, max_batch=B, algo=args.algo, max_iter=args.iters, llr_scale=args.llr_scale,
                layer_rows=layer, poll_interval=2)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
it = torch.empty(B, dtype=torch.int32, device="cuda")
from myldpccppapi_amd import channel
import time
y = torch.empty((B, N), dtype=torch.float32, device="cuda")
print("code=%s algo=%s frames=%d x %d max_iter=%d (info bits per point: %d)" % (
    args.code, args.algo, B, args.batches, args.iters, B * K * args.batches))
points = [float(x) for x in args.snr.split(",")]
# one untimed batch first: the first launch of every kernel (the library's and torch's) pays one-time costs
channel.awgn_device(N, 0, B, 10.0 ** (-points[-1] / 20.0), seed=args.seed + 1, out=y)
dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
channel.count_errors_device(out, None, B)
float(it.float().sum())
torch.cuda.synchronize()
for snr in points:
    sd = 10.0 ** (-snr / 20.0)
    tot = [0, 0, 0]
    it_sum, conv = 0.0, 0
    t0 = time.perf_counter()
    for b in range(args.batches):
        channel.awgn_device(N, b * B, B, sd, seed=args.seed, out=y)
        dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
        e = channel.count_errors_device(out, None, B)
        tot = [t + x for t, x in zip(tot, e)]
        it_sum += float(it.float().sum())
        conv += dec.stats()["frames_converged"]
    dt = time.perf_counter() - t0
    frames = B * args.batches
    print(json.dumps({"snr_db": snr, "sd": round(sd, 4), "ber": tot[0] / (frames * K), "bit_errors": tot[0],
                      "byte_errors": tot[1], "fer": tot[2] / frames, "avg_iters": round(it_sum / frames, 2),
                      "frames_converged": conv, "frames": frames,
                      "end_to_end_Mbit_s": round(frames * K / dt / 1e6, 1)}), flush=True)
