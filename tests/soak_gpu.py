#!/usr/bin/env python3
"""Randomized soak of the quasi-cyclic decoders against the CPU oracle (test infrastructure:
imports oracle/).  Random base matrices (circulant size, layers, row weights 1..24, single-layer
columns anywhere), random noise levels, a few zeros / huge values / NaNs in the channel values;
layered, min-sum and fused-flooding min-sum, each on the record kernels (default), on the
LDS-resident kernels and on the streaming kernels.  usage: python tests/soak_gpu.py [cases] [seed]  (a script, not collected by pytest)"""
import os, sys
sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import numpy as np
import myldpccppapi_amd as L
from myldpccppapi_amd import codes, channel
import oracle

cases = int(sys.argv[1]) if len(sys.argv) > 1 else 40
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 1)
bad = 0
for case in range(cases):
    z = int(rng.choice([3, 7, 16, 24, 27, 31, 32, 33, 48, 63, 64, 65, 96, 100, 128, 130, 200]))
    mb = int(rng.integers(1, 13))
    nb = int(rng.integers(mb + 1, mb + 30))
    dmax = int(rng.integers(1, min(24, nb) + 1))
    base = -np.ones((mb, nb), np.int64)
    for i in range(mb):
        d = int(rng.integers(1, dmax + 1))
        base[i, rng.choice(nb, d, replace=False)] = rng.integers(0, z, d)
    for j in range(nb):
        if (base[:, j] < 0).all():
            i = int(rng.integers(0, mb))
            if (base[i] >= 0).sum() < 24:
                base[i, j] = rng.integers(0, z)
    base = base[:, [j for j in range(nb) if (base[:, j] >= 0).any()]]
    mb, nb = base.shape
    M, N = mb * z, nb * z
    K = max(8, min(N, (nb - min(mb, nb - 1)) * z - int(rng.integers(0, z))) // 8 * 8)
    rows, cols = codes.qc_edges(base, z)
    g = L.Graph(rows, cols, M, N)
    og = oracle.Graph(rows, cols, M, N, K)
    B = int(rng.integers(1, 40))
    y = channel.awgn_frames(N, 0, B, float(rng.uniform(0.3, 1.2)), seed=1000 + case)
    if B > 2:
        y[1, rng.choice(N, max(1, N // 7), replace=False)] = 0.0
        y[2, rng.choice(N, max(1, N // 50), replace=False)] *= 3000.0
    if B > 4 and rng.random() < 0.3:
        y[4, rng.choice(N, 2, replace=False)] = np.nan
    iters = int(rng.integers(1, 25))
    for algo in ("layered", "ms", "ms_fused"):
        want = oracle.decode(og, y, algo, layer_rows=z, max_iter=iters)
        ok = want["undefined"] == 0 if "undefined" in want else np.ones(B, bool)
        kb = K // 8
        for mode, env in (("record", {"LDPC_TUNE_FUSED": "1", "LDPC_TUNE_LDSP": "1"}),
                          ("lds", {"LDPC_TUNE_FUSED": "1", "LDPC_TUNE_LDSP": "0"}),
                          ("stream", {"LDPC_TUNE_FUSED": "0", "LDPC_TUNE_LDSP": "0"})):
            if algo == "ms_fused" and mode == "stream":
                continue
            os.environ.update(env)
            os.environ["LDPC_TUNE_LDSP_GRID"] = str(int(rng.integers(1, 9)))
            try:
                dec = L.Decoder(g, K, max_batch=B, algo=algo, layer_rows=z, max_iter=iters, tune=L.capi.tune_from_env())
            except L.LdpcError as e:
                if algo == "ms_fused":
                    continue            # not every structure fits the one-launch kernels
                raise
            out, it = dec.decode(y)
            good = np.array_equal(out.reshape(B, kb)[ok], want["out"].reshape(B, kb)[ok]) and np.array_equal(it[ok], want["iters"][ok])
            dec.close()
            if not good:
                bad += 1
                print("MISMATCH case", case, "z", z, "base", base.shape, "K", K, "B", B, "iters", iters, algo, mode, flush=True)
    print("case %d ok: z=%d base=%dx%d K=%d B=%d iters=%d" % (case, z, mb, nb, K, B, iters), flush=True)
# second phase: streaming kernels with early termination, host polling and tail compaction on 802.16e
# codes: mixed noise levels so that a few stragglers remain while most frames have converged
for case in range(max(4, cases // 6)):
    rate = int(rng.integers(0, 6))
    N = int(rng.choice([576, 672, 960, 1152, 2304]))
    K, M, z = codes.wimax_dims(rate, N)
    rows, cols = codes.wimax_edges(rate, N)
    g = L.Graph(rows, cols, M, N)
    og = oracle.Graph(rows, cols, M, N, K)
    B = int(rng.integers(130, 900))
    y = channel.awgn_frames(N, 0, B, float(rng.uniform(0.3, 0.6)) * (1.0 if rate < 3 else 0.7), seed=5000 + case)
    ns = int(rng.integers(1, 60))
    slow = rng.choice(B, ns, replace=False)
    y[slow] = channel.awgn_frames(N, 7000, ns, float(rng.uniform(0.9, 1.4)), seed=6000 + case)
    algo = ("sp", "ms", "ms")[case % 3]
    f16 = algo == "ms" and case % 2 == 1
    iters = int(rng.integers(5, 30))
    want = oracle.decode(og, y, algo, max_iter=iters, msg_f16=f16)
    os.environ["LDPC_TUNE_FUSED"] = "0"
    os.environ["LDPC_TUNE_LDSP"] = "0"
    os.environ["LDPC_TUNE_COMPACT"] = str(int(rng.choice([512, 512, 64, 9])))
    tune = L.capi.tune_from_env()
    tune["merge"] = bool(rng.integers(0, 2))
    devs = None if rng.random() < 0.6 else [0] * int(rng.integers(2, 4))
    dec = L.Decoder(g, K, max_batch=B if devs is None else max(64, B // 3), algo=algo, max_iter=iters,
                    poll_interval=int(rng.integers(0, 4)),
                    frames_per_lane=int(rng.choice([1, 2, 4])), msg_dtype="f16" if f16 else "f32",
                    tune=tune, devices=devs)
    out, it = dec.decode(y)
    good = np.array_equal(out, want["out"]) and np.array_equal(it, want["iters"])
    dec.close()
    if not good:
        bad += 1
        print("MISMATCH streaming case", case, rate, N, B, algo, f16, iters, flush=True)
    print("streaming case %d ok: rate=%d N=%d B=%d %s%s iters=%d stragglers=%d" % (case, rate, N, B, algo, "16" if f16 else "", iters, ns), flush=True)
# third phase: the streaming kernels on IRA (DVB-S2-profile) codes, where the column-fused check kernel
# runs: every form of it (wide / narrow / half / deep), merged and per-class launches, host polling,
# the device-side tail (large asynchronous batches, called twice: idle hint) and device lists
import torch  # noqa: E402
for case in range(max(6, cases // 4)):
    N2, K2 = (12960, 6480) if case % 3 else (16200, 10800)
    rows, cols = codes.dvbs2_profile_edges(N2, K2, profile=None if case % 3 else [(9, 5), (3, 25)])   # column degrees 9, 3, 2
    M2 = N2 - K2
    g = L.Graph(rows, cols, M2, N2)
    og = oracle.Graph(rows, cols, M2, N2, K2)
    big = case % 2 == 0
    B = int(rng.integers(2100, 2600)) if big else int(rng.integers(65, 700))
    algo = ("sp", "ms", "ms")[case % 3]
    f16 = algo == "ms" and case % 2 == 1
    iters = int(rng.integers(6, 22))
    y = channel.awgn_frames(N2, 0, B, 0.62 if algo == "ms" else 0.5, seed=8000 + case)
    ns = int(rng.integers(1, 70))
    slow = rng.choice(B, ns, replace=False)
    y[slow] = channel.awgn_frames(N2, 9000, ns, float(rng.uniform(0.95, 1.3)), seed=8500 + case)
    from concurrent.futures import ThreadPoolExecutor
    chunks = np.array_split(np.arange(B), 16)
    with ThreadPoolExecutor(16) as ex:
        parts = list(ex.map(lambda ix: oracle.decode(og, y[ix], algo, max_iter=iters, msg_f16=f16), chunks))
    want_out = np.concatenate([p_["out"] for p_ in parts])
    want_it = np.concatenate([p_["iters"] for p_ in parts])
    variant = [{}, {"link_narrow": True}, {"link_narrow": False}, {"link_half": True}, {"link_deep": True}][int(rng.integers(0, 5))]
    tune = dict(variant, merge=bool(rng.integers(0, 2)), link_rows=int(rng.choice([0, 3, 8, 16, 21])))
    poll = 0 if big else int(rng.integers(0, 3))
    devs = None if big or rng.random() < 0.5 else [0] * int(rng.integers(2, 4))
    fpl = int(rng.choice([1, 2, 4]))
    dec = L.Decoder(g, K2, max_batch=B if devs is None else max(64, B // 3), algo=algo, max_iter=iters, poll_interval=poll,
                    frames_per_lane=fpl, msg_dtype="f16" if f16 else "f32", tune=tune, devices=devs)
    good = True
    for rep in range(2):
        out, it = dec.decode(y)
        good = good and np.array_equal(out, want_out) and np.array_equal(it, want_it)
    dec.close()
    if not good:
        bad += 1
        print("MISMATCH ira case", case, N2, B, algo, f16, iters, tune, poll, devs, fpl, flush=True)
    print("ira case %d ok: N=%d B=%d %s%s iters=%d stragglers=%d tune=%s poll=%d devices=%s fpl=%d" % (
        case, N2, B, algo, "16" if f16 else "", iters, ns, tune, poll, devs, fpl), flush=True)
print("soak finished:", cases, "cases,", bad, "mismatches")
sys.exit(1 if bad else 0)
