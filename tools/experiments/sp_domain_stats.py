#!/usr/bin/env python3
"""How often does the sum-product variable node leave the domain in which the IEEE division's scaling /
fix-up instructions do nothing?  Runs the CPU oracle (test infrastructure) on a few frames of the headline
workload, taps r0 / r1 after round i, and classifies every (frame, column) by its partial products.
usage: sp_domain_stats.py [frames] [sigma]   (CPU only; about a minute per frame)"""
import sys
import os
import numpy as np
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import oracle
from myldpccppapi_amd import codes

frames = int(sys.argv[1]) if len(sys.argv) > 1 else 2
sigma = float(sys.argv[2]) if len(sys.argv) > 2 else 0.95
N, K = 64800, 32400
rows, cols = codes.dvbs2_profile_edges(N, K)
g = oracle.Graph(rows, cols, N - K, N, K)
y = oracle.awgn(N, 0, frames, sigma, seed=20260101)
f32 = np.float32
L = f32(2.0 ** -100)
deg = np.diff(g.col_ptr)
print("frames", frames, "sigma", sigma)
for it in (1, 2, 3, 5, 8, 12, 20, 30, 49):
    o = oracle.decode(g, y, "sp", max_iter=50, tap_iter=it)
    r0, r1 = o["taps"]["r0"], o["taps"]["r1"]
    t = np.exp((f32(8.0) * y).astype(f32)).astype(f32)
    with np.errstate(all="ignore"):
        p0 = (t / (f32(1) + t)).astype(f32)
        p1 = (f32(1) / (f32(1) + t)).astype(f32)
    tot = nan = plain = zero_ok = fallback = colfull_ok = 0
    for d in np.unique(deg):
        if d == 0:
            continue
        cs = np.nonzero(deg == d)[0]
        e = np.stack([g.col_edge[g.col_ptr[cs] + k] for k in range(d)], 1)      # [cols, d]
        a0 = r0[:, e]
        a1 = r1[:, e]                                                             # [frames, cols, d]
        col_ok = np.ones((frames, len(cs)), bool)
        col_nan = np.zeros((frames, len(cs)), bool)
        col_zero = np.zeros((frames, len(cs)), bool)
        with np.errstate(all="ignore"):
            for k in range(d):
                t0 = p0[:, cs].copy()
                t1 = p1[:, cs].copy()
                for j in range(d):
                    if j != k:
                        t0 = (t0 * a0[:, :, j]).astype(f32)
                        t1 = (t1 * a1[:, :, j]).astype(f32)
                s = (t0 + t1).astype(f32)
                ok = (s >= L) & ((t0 == 0) | (t0 >= L)) & ((t1 == 0) | (t1 >= L))
                col_ok &= ok
                col_nan |= np.isnan(s) | (s == 0)
                col_zero |= ok & ((t0 == 0) | (t1 == 0))
            f0 = p0[:, cs].copy(); f1 = p1[:, cs].copy()
            for j in range(d):
                f0 = (f0 * a0[:, :, j]).astype(f32); f1 = (f1 * a1[:, :, j]).astype(f32)
            colfull_ok += int(((f0 >= L) & (f1 >= L)).sum())
        tot += col_ok.size
        nan += int((~col_ok & col_nan).sum())
        plain += int((col_ok & ~col_zero).sum())
        zero_ok += int((col_ok & col_zero).sum())
        fallback += int((~col_ok & ~col_nan).sum())
    print("round %2d: all pairs in domain, no zero %.4f | in domain with zero numerators %.4f | NaN or 0/0 %.4f | "
          "other (needs the scaled division) %.6f | full-product test alone would pass %.4f"
          % (it, plain / tot, zero_ok / tot, nan / tot, fallback / tot, colfull_ok / tot), flush=True)
