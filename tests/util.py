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


def kernel_choice(mode, grid=0):
    """Tuning dict for L.Decoder(tune=...): "1" = LDS-resident fused kernels, "ldsp" = record
    kernels (posteriors in LDS, check records in cache), "0" = HBM-streaming kernels."""
    t = {"fused": mode != "0", "ldsp": mode == "ldsp"}
    if grid:
        t["ldsp_grid"] = grid
    return t


def converged_frames(rows, cols, M, hard):
    """Per frame: are all parity checks of the final hard bits even?  (The reference's flags == 0,
    decodeCL.c:88-108; a frame can end clean exactly at the last iteration, so `iters < max_iter`
    is not the same thing.)"""
    hard = np.asarray(hard, np.uint8)
    syn = np.zeros((hard.shape[0], M), np.int64)
    np.add.at(syn, (slice(None), np.asarray(rows)), hard[:, np.asarray(cols)].astype(np.int64))
    return ~(syn & 1).any(axis=1)


def first_clean_iters(flags_or_iters):
    return np.asarray(flags_or_iters)


def assert_taps_equal(got, want, active, what):
    """Compare message dumps on the frames that were still running at the tapped
    iteration (the reference leaves frozen frames' buffers stale, the oracle leaves
    them untouched)."""
    for f in np.nonzero(active)[0]:
        assert np.array_equal(got[f], want[f], equal_nan=True), "%s differs on frame %d" % (what, f)


def host_channel_lib(tmp_dir):
    """csrc/ldpc_channel.h (shared by host and device code) compiled for the host: the checker of
    the device channel.  Returns a ctypes library with
      philox(uint32 c[4], k0, k1), normals(seed, frame, groups, double *out),
      awgn(float *llr, frames, N, const uint8 *bits, float sd, seed, first_frame)."""
    import ctypes
    import subprocess
    root = os.path.join(os.path.dirname(os.path.abspath(__file__)), "..")
    src = os.path.join(str(tmp_dir), "ch.c")
    with open(src, "w") as f:
        f.write('#include "%s/myldpccppapi_amd/csrc/ldpc_channel.h"\n' % root + """
void philox(uint32_t *c, uint32_t k0, uint32_t k1) { ldpc_philox4x32_10(c, k0, k1); }
void normals(uint64_t seed, uint64_t frame, long groups, double *out)
{ for (long g = 0; g < groups; ++g) ldpc_ch_normal4(seed, frame, (uint32_t)g, out + 4 * g); }
void awgn(float *llr, long frames, int N, const unsigned char *bits, float sd, uint64_t seed, long first_frame)
{
    for (long f = 0; f < frames; ++f)
        for (int g = 0; g < (N + 3) / 4; ++g) {
            double z[4];
            ldpc_ch_normal4(seed, (uint64_t)(first_frame + f), (uint32_t)g, z);
            for (int i = 0; i < 4 && g * 4 + i < N; ++i)
                llr[f * N + g * 4 + i] = ldpc_ch_sample(bits ? bits[f * N + g * 4 + i] & 1 : 0, sd, z[i]);
        }
}
""")
    so = os.path.join(str(tmp_dir), "ch.so")
    subprocess.check_call(["gcc", "-O2", "-ffp-contract=off", "-shared", "-fPIC", src, "-o", so])
    lib = ctypes.CDLL(so)
    lib.normals.argtypes = [ctypes.c_uint64, ctypes.c_uint64, ctypes.c_long, ctypes.c_void_p]
    lib.awgn.argtypes = [ctypes.c_void_p, ctypes.c_long, ctypes.c_int, ctypes.c_void_p, ctypes.c_float,
                         ctypes.c_uint64, ctypes.c_long]
    return lib
