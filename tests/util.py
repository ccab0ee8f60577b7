"""Shared helpers for the tests (the oracle is imported here, never in the product)."""
import glob
import os

import numpy as np

import oracle
from myldpccppapi_amd import codes

GOLDEN = os.path.join(os.path.dirname(os.path.abspath(__file__)), "golden")


def golden_files(prefix):
    return sorted(glob.glob(os.path.join(GOLDEN, prefix + "_*.npz")))


def load_golden(path):
    with np.load(path, allow_pickle=False) as z:
        return {k: z[k] for k in z.files}


def wimax_oracle_graph(rate, N):
    K, M, z = codes.wimax_dims(rate, N)
    rows, cols = codes.wimax_edges(rate, N)
    return oracle.Graph(rows, cols, M, N, K), rows, cols, K, M, z


def first_clean_iters(flags_or_iters):
    return np.asarray(flags_or_iters)


def assert_taps_equal(got, want, active, what):
    """Compare message dumps on the frames that were still running at the tapped
    iteration (the reference leaves frozen frames' buffers stale, the oracle leaves
    them untouched)."""
    for f in np.nonzero(active)[0]:
        assert np.array_equal(got[f], want[f], equal_nan=True), "%s differs on frame %d" % (what, f)
