#!/usr/bin/env python3
"""configs[4] (rate 9/10, fp16 messages, early termination): how the iteration counts of a 4096-frame batch
are distributed, and what tile-rounds different compaction schedules would launch.
usage: gpu_iter_hist.py [sigma]"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import torch
import myldpccppapi_amd as L
from myldpccppapi_amd import channel, codes

sigma = float(sys.argv[1]) if len(sys.argv) > 1 else 0.43
N, K, B = 64800, 58320, 4096
rows, cols = codes.dvbs2_profile_edges(N, K)
g = L.Graph(rows, cols, N - K, N)
dec = L.Decoder(g, K, max_batch=B, algo="ms", max_iter=50, msg_dtype="f16", poll_interval=2)
y = channel.awgn_device(N, 0, B, sigma, seed=20260101)
out = torch.empty(L.out_bytes(K, B), dtype=torch.uint8, device="cuda")
it = torch.empty(B, dtype=torch.int32, device="cuda")
dec.decode_device(y.data_ptr(), B, out.data_ptr(), out.numel(), it.data_ptr(), None)
torch.cuda.synchronize()
st = dec.stats()
iters = it.cpu().numpy()
print("avg", iters.mean(), "max", iters.max(), "frame_rounds", st["frame_rounds"], "= tile-rounds of 256:", st["frame_rounds"] / 256)
hist = np.bincount(iters)
print("histogram", {i: int(c) for i, c in enumerate(hist) if c})
F = 256
tiles = iters.reshape(-1, F)
print("tile maxima", tiles.max(1).tolist(), "mean", tiles.max(1).mean())
R = int(iters.max())
running = [int((iters >= r).sum()) for r in range(1, R + 1)]          # frames that take part in round r
print("frames in round r", running)
plain = sum(int((tiles.max(1) >= r).sum()) for r in range(1, R + 1))
ideal = sum(-(-n // F) for n in running)
print("tile-rounds: no compaction %d, ideal packing every round %d, sum iters / F %.1f" % (plain, ideal, iters.sum() / F))
# compaction at the polls (after rounds 2, 4, 6, ...): packed tiles from the next round on
for when in ((4,), (5,), (6,), (4, 6), (3, 5), (2, 4, 6), (3, 4, 5, 6)):
    cur = iters.copy()
    order = np.arange(B)
    total, moved = 0, 0
    slots = order.copy()
    for r in range(1, R + 1):
        t = iters[slots].reshape(-1, F) if len(slots) % F == 0 else None
        act = (iters[slots].reshape(-1, F).max(1) >= r).sum()
        total += int(act)
        if r in when:
            run = slots[iters[slots] > r]                       # still running after round r
            pad = (-len(run)) % F
            moved += len(run)
            done_fill = slots[iters[slots] <= r][:pad]
            slots = np.concatenate([run, done_fill]) if len(run) else slots[:0]
            if len(slots) == 0:
                break
    print("compaction after rounds %s: %d tile-rounds, %d frame moves" % (when, total, moved))
