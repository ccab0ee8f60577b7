#!/usr/bin/env python3
"""Generate tests/golden/*.npz from the REFERENCE's own kernel source.

Runs only in the build container (needs /root/reference): the reference's
decodeCL.c is compiled as host C (oracle/Makefile `ref`, oracle/ref_host/) and
driven in the order the reference's host code enqueues its kernels
(oracle/ref_host/ref_driver.c).  Each fixture stores the inputs (code id, channel
values) and what the reference produced: packed output bytes, hard bits, the
batch's `Time=`, per-frame syndrome flags and the messages of one iteration.

A fixture is DATA (inputs + expected outputs); no reference source is stored.
The reference ships no golden vectors of its own (Test.cpp:29 seeds from time(0)).

    python oracle/make_golden.py            # rewrites tests/golden/
"""
import os
import sys

import numpy as np

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import oracle  # noqa: E402
from oracle import refkernels as rk  # noqa: E402
from myldpccppapi_amd.wimax_seeds import SEEDS  # noqa: E402  (data only)

OUT = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", "tests", "golden")

# (name, rate index, N, sigma, frames, seed, tap_iter)
FLOOD_CASES = [
    ("c576_34b", 4, 576, 0.50, 8, 11, 1),     # Test.cpp's hard-coded code (Test.cpp:19-26)
    ("c648_12", 0, 648, 0.75, 8, 12, 2),      # BASELINE config 1 dimensions; K % 8 != 0
    ("c672_23a", 1, 672, 0.62, 6, 13, 2),     # the `% z` shift rule (MyLdpc.cpp:92-93)
    ("c960_23b", 2, 960, 0.62, 5, 14, 2),
    ("c1152_34a", 3, 1152, 0.55, 4, 15, 2),
    ("c576_56", 5, 576, 0.42, 6, 16, 1),      # row weight 20
    ("c2304_12_hard", 0, 2304, 0.90, 4, 17, 3),   # does not converge: SP floods with NaN
    ("c2304_12_easy", 0, 2304, 0.70, 4, 18, 2),
]
# fused layered kernel: floor-scaled shifts only (so not rate 2/3A unless z == 96), z <= 127
LAYERED_CASES = [
    ("l576_12", 0, 576, 0.80, 6, 21),
    ("l576_34a", 3, 576, 0.50, 8, 22),
    ("l960_23b", 2, 960, 0.70, 4, 23),
    ("l1152_56", 5, 1152, 0.45, 4, 24),
    ("l2304_12_hard", 0, 2304, 0.95, 3, 25),
    ("l2304_12_easy", 0, 2304, 0.80, 4, 26),
    ("l2304_34b", 4, 2304, 0.60, 3, 27),
]


# fused flooding kernel decodeOnceMS (DecodeMSCL): times fixed at 120; same shift restrictions
MSCL_CASES = [
    ("m576_12", 0, 576, 0.80, 6, 31),
    ("m576_12_hard", 0, 576, 1.00, 3, 32),
    ("m576_34a", 3, 576, 0.60, 4, 33),
    ("m960_23b", 2, 960, 0.70, 3, 34),
    ("m1152_56", 5, 1152, 0.45, 3, 35),
    ("m2304_12", 0, 2304, 0.85, 2, 36),
]


# host-layered path (DecodeTDMP, MyLdpc.cpp:889-976): only the seeds whose rows all have the same
# weight -- 2/3A (10) and 5/6 (20) -- where the reference's layer sizes are right
# (name, rate index, N, sigma, frames, seed, tap_iter)
TDMP_HOST_CASES = [
    ("t672_23a", 1, 672, 0.62, 6, 41, 2),
    ("t2304_23a", 1, 2304, 0.66, 4, 42, 2),
    ("t2304_23a_hard", 1, 2304, 0.80, 3, 43, 3),
    ("t576_56", 5, 576, 0.42, 8, 44, 1),
    ("t1152_56", 5, 1152, 0.47, 4, 45, 2),
    ("t2304_56_hard", 5, 2304, 0.60, 3, 46, 2),
]


def channel(N, frames, sigma, seed):
    rng = np.random.Generator(np.random.Philox(key=[20260101, seed]))
    return (1.0 + sigma * rng.standard_normal((frames, N))).astype(np.float32)


def main():
    os.makedirs(OUT, exist_ok=True)
    for name, rate, N, sigma, B, seed, tap in FLOOD_CASES:
        rows, cols = oracle.wimax_edges(rate, N)
        mb = len(SEEDS[rate])
        M = mb * (N // 24)
        K = N - M
        rg = rk.RefGraph(rows, cols, M, N, K)
        y = channel(N, B, sigma, seed)
        data = dict(rate=rate, N=N, K=K, M=M, E=len(rows), sigma=sigma, times=40, tap_iter=tap, y=y)
        for algo in ("ms", "sp"):
            r = rk.decode(rg, y, algo, times=40, tap_iter=tap)
            data[algo + "_out"] = r["out"]
            data[algo + "_hard"] = r["hard"]
            data[algo + "_time"] = r["time"]
            data[algo + "_flags"] = r["flags"]
            for k, v in r["taps"].items():
                data["%s_tap_%s" % (algo, k)] = v
        np.savez_compressed(os.path.join(OUT, "flood_%s.npz" % name), **data)
        print("flood", name, "E", len(rows), "ms time", data["ms_time"], "sp time", data["sp_time"])
    for name, rate, N, sigma, B, seed in LAYERED_CASES:
        z = N // 24
        mb = len(SEEDS[rate])
        K = N - mb * z
        y = channel(N, B, sigma, seed)
        out = rk.decode_tdmp_fused(z, np.array(SEEDS[rate], np.int8), y)
        np.savez_compressed(os.path.join(OUT, "layered_%s.npz" % name), rate=rate, N=N, K=K, z=z,
                            sigma=sigma, times=40, y=y, out=out)
        print("layered", name, "z", z)
    for name, rate, N, sigma, B, seed in MSCL_CASES:
        z = N // 24
        mb = len(SEEDS[rate])
        K = N - mb * z
        y = channel(N, B, sigma, seed)
        out = rk.decode_mscl_fused(z, np.array(SEEDS[rate], np.int8), y)
        np.savez_compressed(os.path.join(OUT, "mscl_%s.npz" % name), rate=rate, N=N, K=K, z=z,
                            sigma=sigma, times=120, y=y, out=out)
        print("mscl", name, "z", z)
    for name, rate, N, sigma, B, seed, tap in TDMP_HOST_CASES:
        rows, cols = oracle.wimax_edges(rate, N)
        z = N // 24
        mb = len(SEEDS[rate])
        M = mb * z
        K = N - M
        rg = rk.RefGraph(rows, cols, M, N, K)
        y = channel(N, B, sigma, seed)
        r = rk.decode_tdmp_host(rg, z, y, times=40, tap_iter=tap)
        np.savez_compressed(os.path.join(OUT, "tdmphost_%s.npz" % name), rate=rate, N=N, K=K, M=M, z=z, sigma=sigma,
                            times=40, tap_iter=tap, y=y, out=r["out"], hard=r["hard"], time=r["time"], flags=r["flags"],
                            tap_r=r["taps"]["r"], tap_post=r["taps"]["post"])
        print("tdmp-host", name, "z", z, "time", r["time"], "clean", int((r["flags"] == 0).sum()), "/", B)
    # graph construction facts the survey measured on the reference's own H builder
    # (E printed by the reference's initCheckMatrix + adjacency build)
    np.savez_compressed(os.path.join(OUT, "graph_facts.npz"),
                        cases=np.array([[4, 576, 2112], [0, 648, 2052], [0, 2304, 7296], [0, 64800, 205200]]))


if __name__ == "__main__":
    main()
