"""ctypes access to oracle/_ref/libldpc_ref.so -- the reference's own kernel file
(/root/reference/decodeCL.c) compiled as host C (oracle/Makefile, target `ref`).

TEST INFRASTRUCTURE ONLY, and only in the build container: the library exists
only where /root/reference exists.  Used by oracle/make_golden.py (to make the
committed fixtures) and by tests that skip when it is absent.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, "_ref", "libldpc_ref.so")
_LIB = None

_ip = ctypes.POINTER(ctypes.c_int)
_fp = ctypes.POINTER(ctypes.c_float)
_u8p = ctypes.POINTER(ctypes.c_uint8)


class _RefGraph(ctypes.Structure):
    _fields_ = [("M", ctypes.c_int), ("N", ctypes.c_int), ("K", ctypes.c_int), ("E", ctypes.c_int),
                ("hRows", _ip), ("hCols", _ip), ("hRowFirstPtr", _ip), ("hRowNextPtr", _ip),
                ("hColFirstPtr", _ip), ("hColNextPtr", _ip)]


class _RefTaps(ctypes.Structure):
    _fields_ = [("iter", ctypes.c_int), ("a", _fp), ("b", _fp), ("c", _fp), ("d", _fp)]


def available():
    return os.path.exists(_SO) or os.path.exists("/root/reference/decodeCL.c")


def lib():
    global _LIB
    if _LIB is None:
        if os.path.exists("/root/reference/decodeCL.c"):
            subprocess.check_call(["make", "-s", "-C", _HERE, "ref"])
        L = ctypes.CDLL(_SO)
        for nm in ("ref_decode_ms", "ref_decode_sp"):
            f = getattr(L, nm)
            f.restype = ctypes.c_int
            f.argtypes = [ctypes.POINTER(_RefGraph), _fp, ctypes.c_int, ctypes.c_int,
                          ctypes.c_char_p, _u8p, _u8p, ctypes.POINTER(_RefTaps)]
        L.ref_decode_tdmp_host.restype = ctypes.c_int
        L.ref_decode_tdmp_host.argtypes = [ctypes.POINTER(_RefGraph), _ip, ctypes.c_int, _fp, ctypes.c_int, ctypes.c_int,
                                           ctypes.c_char_p, _u8p, _u8p, ctypes.POINTER(_RefTaps)]
        L.ref_decode_mscl.restype = ctypes.c_int
        L.ref_decode_mscl.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, _fp, ctypes.c_int,
                                      ctypes.c_char_p]
        L.ref_decode_tdmp.restype = ctypes.c_int
        L.ref_decode_tdmp.argtypes = [ctypes.c_int, ctypes.c_int, ctypes.c_char_p, _fp,
                                      ctypes.c_int, ctypes.c_char_p]
        _LIB = L
    return _LIB


def linked_lists(rows, cols, M, N):
    """The reference's adjacency form (MyLdpc.cpp:171-222): per-row and per-column
    -1-terminated singly linked lists threaded through the edges in edge order."""
    E = len(rows)
    row_first = np.full(M, -1, np.int32)
    row_next = np.full(E, -1, np.int32)
    col_first = np.full(N, -1, np.int32)
    col_next = np.full(E, -1, np.int32)
    last_r = np.full(M, -1, np.int64)
    last_c = np.full(N, -1, np.int64)
    for e in range(E):
        r, c = rows[e], cols[e]
        if last_r[r] < 0:
            row_first[r] = e
        else:
            row_next[last_r[r]] = e
        last_r[r] = e
        if last_c[c] < 0:
            col_first[c] = e
        else:
            col_next[last_c[c]] = e
        last_c[c] = e
    return row_first, row_next, col_first, col_next


class RefGraph:
    def __init__(self, rows, cols, M, N, K):
        self.rows = np.ascontiguousarray(rows, np.int32)
        self.cols = np.ascontiguousarray(cols, np.int32)
        self.M, self.N, self.K, self.E = M, N, K, len(rows)
        self.lists = [np.ascontiguousarray(a) for a in linked_lists(self.rows, self.cols, M, N)]
        p = lambda a: a.ctypes.data_as(_ip)
        self._c = _RefGraph(M, N, K, self.E, p(self.rows), p(self.cols), p(self.lists[0]),
                            p(self.lists[1]), p(self.lists[2]), p(self.lists[3]))


def decode(g, y, algo, times=40, tap_iter=0):
    """Run the reference kernel chain for ONE batch (all frames of y together, as
    decodeOnceMS / decodeOnceSP do).  Returns dict(out, time, hard, flags, taps)."""
    L = lib()
    y = np.ascontiguousarray(y, np.float32).reshape(-1, g.N)
    B = y.shape[0]
    out = np.zeros((B - 1) * g.K // 8 + g.K // 8, np.uint8)
    hard = np.zeros((B, g.N), np.uint8)
    flags = np.zeros(B, np.uint8)
    taps = {}
    ct = None
    if tap_iter:
        ct = _RefTaps()
        ct.iter = tap_iter
        if algo == "ms":
            spec = (("a", "r", g.E), ("b", "post", g.N), ("c", "q", g.E))
        else:
            spec = (("a", "r0", g.E), ("b", "r1", g.E), ("c", "q0", g.E), ("d", "q1", g.E))
        for field, nm, n in spec:
            taps[nm] = np.full((B, n), np.nan, np.float32)
            setattr(ct, field, taps[nm].ctypes.data_as(_fp))
    f = L.ref_decode_ms if algo == "ms" else L.ref_decode_sp
    time = f(ctypes.byref(g._c), y.ctypes.data_as(_fp), B, times,
             out.ctypes.data_as(ctypes.c_char_p), hard.ctypes.data_as(_u8p),
             flags.ctypes.data_as(_u8p), ctypes.byref(ct) if ct is not None else None)
    return dict(out=out, time=time, hard=hard, flags=flags, taps=taps)


def decode_tdmp_fused(z, seed, y):
    """Run the fused layered kernel decodeOnceTDMP (times fixed at 40, floor-scaled
    shifts, z <= 127).  seed: int8 array [mb, 24].  Returns packed bytes."""
    L = lib()
    seed = np.ascontiguousarray(seed, np.int8)
    N = 24 * z
    K = N - seed.shape[0] * z
    y = np.ascontiguousarray(y, np.float32).reshape(-1, N)
    B = y.shape[0]
    out = np.zeros((B - 1) * K // 8 + K // 8, np.uint8)
    rc = L.ref_decode_tdmp(z, seed.shape[0], seed.ctypes.data_as(ctypes.c_char_p),
                           y.ctypes.data_as(_fp), B, out.ctypes.data_as(ctypes.c_char_p))
    if rc:
        raise RuntimeError("ref_decode_tdmp rc=%d" % rc)
    return out


def decode_mscl_fused(z, seed, y):
    """Run the fused flooding kernel decodeOnceMS (times fixed at 120, floor-scaled shifts,
    z <= 127): one host thread per work-item, M = rows of H per frame.  Returns packed bytes."""
    L = lib()
    seed = np.ascontiguousarray(seed, np.int8)
    N = 24 * z
    K = N - seed.shape[0] * z
    y = np.ascontiguousarray(y, np.float32).reshape(-1, N)
    B = y.shape[0]
    out = np.zeros((B - 1) * K // 8 + K // 8, np.uint8)
    rc = L.ref_decode_mscl(z, seed.shape[0], seed.ctypes.data_as(ctypes.c_char_p),
                           y.ctypes.data_as(_fp), B, out.ctypes.data_as(ctypes.c_char_p))
    if rc:
        raise RuntimeError("ref_decode_mscl rc=%d" % rc)
    return out


def decode_tdmp_host(g, z, y, times=40, tap_iter=0):
    """Run the reference's HOST-layered path (Coder::decodeOnceTDMP, MyLdpc.cpp:889-976, over the
    *TDMP kernels decodeCL.c:203-300) for one batch.  Only meaningful when every row of H has the
    same weight (the reference's layer sizes are wrong otherwise); raises if not.
    Returns dict(out, time, hard, flags, taps={r, post})."""
    L = lib()
    y = np.ascontiguousarray(y, np.float32).reshape(-1, g.N)
    B = y.shape[0]
    row_range = np.zeros(g.M + 1, np.int32)
    np.add.at(row_range, g.rows + 1, 1)
    w = row_range[1:]
    if not (w == w[0]).all():
        raise ValueError("rows of different weight: the reference's host-layered path mis-sizes its layers")
    row_range = np.ascontiguousarray(np.cumsum(row_range), np.int32)
    out = np.zeros((B - 1) * g.K // 8 + g.K // 8, np.uint8)
    hard = np.zeros((B, g.N), np.uint8)
    flags = np.zeros(B, np.uint8)
    taps = {}
    ct = None
    if tap_iter:
        ct = _RefTaps()
        ct.iter = tap_iter
        for field, nm, n in (("a", "r", g.E), ("b", "post", g.N)):
            taps[nm] = np.full((B, n), np.nan, np.float32)
            setattr(ct, field, taps[nm].ctypes.data_as(_fp))
    time = L.ref_decode_tdmp_host(ctypes.byref(g._c), row_range.ctypes.data_as(_ip), z, y.ctypes.data_as(_fp), B, times,
                                  out.ctypes.data_as(ctypes.c_char_p), hard.ctypes.data_as(_u8p),
                                  flags.ctypes.data_as(_u8p), ctypes.byref(ct) if ct is not None else None)
    return dict(out=out, time=time, hard=hard, flags=flags, taps=taps)
