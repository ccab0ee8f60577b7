#!/usr/bin/env python3
"""Decode time of one batch versus the hand-over threshold and the polling interval, at operating points where
a few frames in a thousand run to the last round (GPU only).  Prints one line per setting:
    python tools/gpu_handover_probe.py [--algo sp|ms] [--snr=4.5,5.0] [--frames 4096]"""
import argparse, os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel

ap = argparse.ArgumentParser()
ap.add_argument("--algo", default="sp")
ap.add_argument("--snr", default="4.5,5.0")
ap.add_argument("--frames", type=int, default=4096)
ap.add_argument("--batches", type=int, default=4)
ap.add_argument("--compact", default="auto,512,128,-1", help="thresholds to try (auto = the library's choice)")
ap.add_argument("--poll", default="1,2,4")
args = ap.parse_args()
N, K = 64800, 32400
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
B = args.frames
y = torch.empty((B, N), dtype=torch.float32, device="cuda")
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
it = torch.empty(B, dtype=torch.int32, device="cuda")
for compact in [None if c == "auto" else int(c) for c in args.compact.split(",")]:
    for poll in [int(p) for p in args.poll.split(",")]:
        tune = {} if compact is None else {"compact": compact}
        dec = L.Decoder(g, K, max_batch=B, algo=args.algo, max_iter=50, llr_scale=8.0, poll_interval=poll, tune=tune)
        for snr in [float(x) for x in args.snr.split(",")]:
            sd = 10.0 ** (-snr / 20.0)
            ms = []
            for b in range(-1, args.batches):
                channel.awgn_device(N, max(b, 0) * B, B, sd, seed=20260101, out=y)
                torch.cuda.synchronize()
                e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
                e0.record()
                dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
                e1.record()
                torch.cuda.synchronize()
                if b >= 0:
                    ms.append(e0.elapsed_time(e1))
            st = dec.stats()
            print("algo=%s compact=%s poll=%d snr=%.1f ms=%s mean=%.2f rounds=%d frame_rounds=%d" % (
                args.algo, compact, poll, snr, ["%.2f" % m for m in ms], sum(ms) / len(ms), st["iterations_launched"],
                st.get("frame_rounds", -1)), flush=True)
        dec.close()
