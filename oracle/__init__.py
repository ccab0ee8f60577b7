"""CPU oracle for the LDPC decode hot path -- TEST INFRASTRUCTURE ONLY.

Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may import
this package, and only as the checker / the timed CPU baseline.  The product
(myldpccppapi_amd/) never imports it.  See oracle/ldpc_oracle.h.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = None

_i32p = ctypes.POINTER(ctypes.c_int32)
_f32p = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


class _Graph(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int32), ("N", ctypes.c_int32), ("K", ctypes.c_int32),
                ("E", ctypes.c_int64), ("rows", _i32p), ("cols", _i32p),
                ("row_ptr", _i32p), ("col_ptr", _i32p), ("col_edge", _i32p)]


class _Taps(ctypes.Structure):
    _fields_ = [("iter", ctypes.c_int), ("r", _f32p), ("q", _f32p), ("post", _f32p),
                ("r0", _f32p), ("r1", _f32p), ("q0", _f32p), ("q1", _f32p)]


def build(force=False):
    """Compile the C restatement (gcc, seconds).  Returns the .so path."""
    so = os.path.join(_HERE, "libldpc_oracle.so")
    src = os.path.join(_HERE, "ldpc_oracle.c")
    if force or not os.path.exists(so) or os.path.getmtime(so) < os.path.getmtime(src):
        subprocess.check_call(["make", "-s", "-C", _HERE, "libldpc_oracle.so"])
    return so


def lib():
    global _LIB
    if _LIB is None:
        L = ctypes.CDLL(build())
        L.oracle_wimax_nnz.restype = ctypes.c_int64
        L.oracle_wimax_nnz.argtypes = [ctypes.c_int]
        L.oracle_wimax_edges.restype = ctypes.c_int64
        L.oracle_wimax_edges.argtypes = [ctypes.c_int, ctypes.c_int32, _i32p, _i32p]
        L.oracle_build_adjacency.restype = ctypes.c_int
        L.oracle_build_adjacency.argtypes = [ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                             _i32p, _i32p, _i32p, _i32p, _i32p]
        common = [ctypes.POINTER(_Graph)]
        tail = [ctypes.c_int, _u8p, ctypes.c_int64, _i32p, _u8p, ctypes.POINTER(_Taps)]
        L.oracle_decode_ms.restype = ctypes.c_int
        L.oracle_decode_ms.argtypes = common + [_f32p, ctypes.c_int64, ctypes.c_int] + tail + [ctypes.c_int]
        L.oracle_decode_sp.restype = ctypes.c_int
        L.oracle_decode_sp.argtypes = common + [_f32p, ctypes.c_int64, ctypes.c_int,
                                                ctypes.c_float] + tail
        L.oracle_decode_layered.restype = ctypes.c_int
        L.oracle_decode_layered.argtypes = common + [ctypes.c_int32, _f32p, ctypes.c_int64,
                                                     ctypes.c_int] + tail + [_u8p]
        L.oracle_decode_layered_host.restype = ctypes.c_int
        L.oracle_decode_layered_host.argtypes = common + [ctypes.c_int32, _f32p, ctypes.c_int64, ctypes.c_int] + tail
        L.oracle_decode_ms_fused.restype = ctypes.c_int
        L.oracle_decode_ms_fused.argtypes = common + [_f32p, ctypes.c_int64, ctypes.c_int] + tail + [_u8p]
        L.oracle_code_size.restype = ctypes.c_int64
        L.oracle_code_size.argtypes = [ctypes.c_int64, ctypes.c_int32]
        L.oracle_test_channel.restype = None
        L.oracle_test_channel.argtypes = [_u8p, _f32p, ctypes.c_int64, ctypes.c_float]
        _LIB = L
    return _LIB


def _p(a, t):
    return a.ctypes.data_as(t)


def wimax_edges(rate, N):
    """Row-major edge list of the reference's Coder(K, N, rate) matrix."""
    L = lib()
    z = N // 24
    cap = int(L.oracle_wimax_nnz(rate)) * z
    rows = np.empty(cap, np.int32)
    cols = np.empty(cap, np.int32)
    E = L.oracle_wimax_edges(rate, N, _p(rows, _i32p), _p(cols, _i32p))
    assert E == cap
    return rows, cols


class Graph:
    """Edge list (row-major) + the CSR/CSC views the oracle decoders walk."""

    def __init__(self, rows, cols, M, N, K):
        self.rows = np.ascontiguousarray(rows, np.int32)
        self.cols = np.ascontiguousarray(cols, np.int32)
        self.M, self.N, self.K, self.E = int(M), int(N), int(K), int(len(rows))
        self.row_ptr = np.empty(M + 1, np.int32)
        self.col_ptr = np.empty(N + 1, np.int32)
        self.col_edge = np.empty(self.E, np.int32)
        rc = lib().oracle_build_adjacency(M, N, self.E, _p(self.rows, _i32p), _p(self.cols, _i32p),
                                          _p(self.row_ptr, _i32p), _p(self.col_ptr, _i32p),
                                          _p(self.col_edge, _i32p))
        if rc:
            raise ValueError("edge list is not a row-major listing of an M x N matrix")
        self._c = _Graph(M, N, K, self.E, _p(self.rows, _i32p), _p(self.cols, _i32p),
                         _p(self.row_ptr, _i32p), _p(self.col_ptr, _i32p),
                         _p(self.col_edge, _i32p))


def out_len(frames, K, pack_mode=0):
    """Bytes a decode of `frames` frames may write."""
    if pack_mode == 0:
        return (frames - 1) * K // 8 + K // 8 if frames else 0
    return (frames * K + 7) // 8


def decode(g, y, algo, max_iter=40, llr_scale=8.0, pack_mode=0, layer_rows=0, tap_iter=0, msg_f16=False):
    """Run one oracle decoder.  algo in {"ms", "sp", "layered", "layered_host", "ms_fused"}.
    "layered_host" = the reference's host-layered DecodeTDMP path (uniform row weight only).

    "ms_fused" = the arithmetic of the reference's fused flooding kernel (DecodeMSCL).
    Returns dict(out=bytes array, iters=int32[frames], hard=uint8[frames,N],
    taps=dict of float arrays when tap_iter > 0)."""
    L = lib()
    y = np.ascontiguousarray(y, np.float32).reshape(-1, g.N)
    frames = y.shape[0]
    out = np.zeros(out_len(frames, g.K, pack_mode), np.uint8)
    iters = np.zeros(frames, np.int32)
    hard = np.zeros((frames, g.N), np.uint8)
    taps = {}
    ctaps = None
    if tap_iter:
        ctaps = _Taps()
        ctaps.iter = tap_iter
        names = ("r0", "r1", "q0", "q1") if algo == "sp" else (("r", "post") if algo in ("layered", "layered_host", "ms_fused") else ("r", "q", "post"))
        for nm in names:
            n = g.N if nm == "post" else g.E
            taps[nm] = np.full((frames, n), np.nan, np.float32)
            setattr(ctaps, nm, _p(taps[nm], _f32p))
    tp = ctypes.byref(ctaps) if ctaps is not None else None
    tail = (pack_mode, _p(out, _u8p), out.size, _p(iters, _i32p), _p(hard, _u8p), tp)
    if algo == "ms":
        rc = L.oracle_decode_ms(ctypes.byref(g._c), _p(y, _f32p), frames, max_iter, *tail, int(msg_f16))
    elif algo == "sp":
        rc = L.oracle_decode_sp(ctypes.byref(g._c), _p(y, _f32p), frames, max_iter,
                                ctypes.c_float(llr_scale), *tail)
    elif algo == "layered":
        undef = np.zeros(frames, np.uint8)
        rc = L.oracle_decode_layered(ctypes.byref(g._c), layer_rows, _p(y, _f32p), frames,
                                     max_iter, *tail, _p(undef, _u8p))
    elif algo == "layered_host":
        rc = L.oracle_decode_layered_host(ctypes.byref(g._c), layer_rows, _p(y, _f32p), frames, max_iter, *tail)
        if rc == -3:
            raise ValueError("layered_host: rows of different weight (the reference's host-layered path is "
                             "only defined for uniform row weight, MyLdpc.cpp:907,958)")
    elif algo == "ms_fused":
        undef = np.zeros(frames, np.uint8)
        rc = L.oracle_decode_ms_fused(ctypes.byref(g._c), _p(y, _f32p), frames, max_iter, *tail,
                                      _p(undef, _u8p))
    else:
        raise ValueError(algo)
    if rc:
        raise RuntimeError("oracle decode failed rc=%d" % rc)
    res = dict(out=out, iters=iters, hard=hard, taps=taps)
    if algo in ("layered", "ms_fused"):
        res["undefined"] = undef
    return res


def test_channel(prior_bytes, sd, seed):
    """Coder::test (MyLdpc.cpp:1061-1078) with libc rand() seeded by `seed`."""
    libc = ctypes.CDLL(None)
    libc.srand(ctypes.c_uint(seed))
    prior = np.ascontiguousarray(prior_bytes, np.uint8)
    post = np.empty(prior.size * 8, np.float32)
    lib().oracle_test_channel(_p(prior, _u8p), _p(post, _f32p), prior.size, ctypes.c_float(sd))
    return post


def awgn(N, first_frame, frames, sd, seed=20260101, codewords=None):
    """float32 [frames, N]: the counter-based channel of csrc/ldpc_channel.h evaluated on the host
    (what ldpc_awgn_device produces on the GPU, float for float)."""
    out = np.empty((frames, N), np.float32)
    bits = None if codewords is None else np.ascontiguousarray(codewords, np.uint8)
    f = lib().oracle_awgn
    f.restype = None
    f.argtypes = [_f32p, ctypes.c_int64, ctypes.c_int32, _u8p, ctypes.c_float, ctypes.c_uint64, ctypes.c_int64]
    f(_p(out, _f32p), frames, N, None if bits is None else _p(bits, _u8p), ctypes.c_float(sd), seed, first_frame)
    return out
