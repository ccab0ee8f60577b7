#!/usr/bin/env python3
"""Stress of the streams handle (ldpc_decoder_config.streams): device-pointer calls (asynchronous and
polled = one host thread per range), host-buffer calls, creation and destruction, over and over in one
process, against the single-stream decoder's bytes.  usage: gpu_streams_stress.py [rounds]"""
import sys, os
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 20
N = 2304
K, M, z = codes.wimax_dims(codes.RATE_1_2, N)
rows, cols = codes.wimax_edges(codes.RATE_1_2, N)
g = L.Graph(rows, cols, M, N)
B = 2048
y = channel.awgn_device(N, 0, B, 0.8, seed=61, device=0)
yh = y.cpu().numpy()
want = {}
for algo, poll in (("sp", 0), ("ms", 2), ("layered", 0)):
    one = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=20, layer_rows=z, poll_interval=poll,
                    tune={"fused": False, "ldsp": False} if algo != "layered" else None)
    for n in (B, 1500, 1):
        want[algo, n] = one.decode(yh[:n])
    one.close()
bad = 0
for rnd in range(rounds):
    for algo, poll in (("sp", 0), ("ms", 2), ("layered", 0)):
        # as the test does: fresh channel values on the device and their copy to the host, every time
        y = channel.awgn_device(N, 0, B, 0.8, seed=61, device=0)
        yh2 = y.cpu().numpy()
        if not np.array_equal(yh2, yh):
            bad += 1
            print("MISMATCH channel", rnd, flush=True)
        for streams in (2, 3):
            dec = L.Decoder(g, K, max_batch=B, algo=algo, max_iter=20, layer_rows=z, poll_interval=poll, streams=streams,
                            tune={"fused": False, "ldsp": False} if algo != "layered" else None)
            s = torch.cuda.Stream()
            for n in (B, 1500, 1, B):
                out = torch.zeros(L.out_bytes(K, n), dtype=torch.uint8, device="cuda")
                it = torch.zeros(n, dtype=torch.int32, device="cuda")
                with torch.cuda.stream(s):
                    dec.decode_device(y.data_ptr(), n, out.data_ptr(), out.numel(), it.data_ptr(), s.cuda_stream)
                    o, i = out.cpu().numpy(), it.cpu().numpy()
                if not (np.array_equal(o, want[algo, n][0]) and np.array_equal(i, want[algo, n][1])):
                    bad += 1
                    print("MISMATCH device", rnd, algo, streams, n, flush=True)
            oh, ih = dec.decode(yh[:1500])
            if not (np.array_equal(oh, want[algo, 1500][0]) and np.array_equal(ih, want[algo, 1500][1])):
                bad += 1
                print("MISMATCH host", rnd, algo, streams, flush=True)
            dec.close()
        # too small a batch per stream: one plain decoder behind the handle, default kernel choice
        small = L.Decoder(g, K, max_batch=600, algo=algo, max_iter=20, layer_rows=z, streams=2)
        o, i = small.decode(yh[:300])
        if not (np.array_equal(o, want[algo, 1500][0][:300 * K // 8]) and np.array_equal(i, want[algo, 1500][1][:300])):
            bad += 1
            print("MISMATCH small", rnd, algo, flush=True)
        small.close()
    print("round", rnd, "done, mismatches", bad, flush=True)
print("stress finished:", rounds, "rounds,", bad, "mismatches")
