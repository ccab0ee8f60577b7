"""Thin Python objects over the C ABI (include/ldpc_hip.h): Graph and Decoder.

Host-side plumbing only -- every decode runs in the HIP kernels of
libldpc_hip.so.  numpy arrays are passed as host pointers, integers (e.g. a
torch tensor's data_ptr()) as device pointers.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import DecoderConfig, DecodeStats, LdpcError  # noqa: F401

ALGO_SP, ALGO_MS, ALGO_LAYERED, ALGO_MS_FUSED = 0, 1, 2, 3
MSG_F32, MSG_F16 = 0, 1
PACK_BYTES, PACK_BITS = 0, 1
ALGOS = {"sp": ALGO_SP, "ms": ALGO_MS, "layered": ALGO_LAYERED, "ms_fused": ALGO_MS_FUSED}


def device_count():
    n = ctypes.c_int(0)
    rc = _lib.load().ldpc_device_count(ctypes.byref(n))
    return n.value if rc == 0 else 0


def out_bytes(K, frames, pack_mode=PACK_BYTES):
    return int(_lib.load().ldpc_out_bytes(K, frames, pack_mode))


class Graph:
    """Parity-check matrix as its nonzeros in row-major order (edge id = rank),
    the form Coder::forDecoder builds at MyLdpc.cpp:171-222."""

    def __init__(self, rows, cols, M, N):
        L = _lib.load()
        self.rows = np.ascontiguousarray(rows, np.int32)
        self.cols = np.ascontiguousarray(cols, np.int32)
        if self.rows.shape != self.cols.shape or self.rows.ndim != 1:
            raise ValueError("rows and cols must be 1-D arrays of equal length")
        self.M, self.N, self.E = int(M), int(N), int(self.rows.size)
        self._h = ctypes.c_void_p()
        i32p = ctypes.POINTER(ctypes.c_int32)
        _lib.check(L.ldpc_graph_create(self.rows.ctypes.data_as(i32p), self.cols.ctypes.data_as(i32p),
                                       self.E, self.M, self.N, ctypes.byref(self._h)))

    def info(self):
        L = _lib.load()
        M, N, rd, cd = (ctypes.c_int32() for _ in range(4))
        E = ctypes.c_int64()
        _lib.check(L.ldpc_graph_info(self._h, ctypes.byref(M), ctypes.byref(N), ctypes.byref(E),
                                     ctypes.byref(rd), ctypes.byref(cd)))
        return dict(M=M.value, N=N.value, E=E.value, max_row_deg=rd.value, max_col_deg=cd.value)

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:   # module may be torn down at exit
            _lib._lib.ldpc_graph_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass


class Decoder:
    """One decoder handle = the device state Coder::forDecoder + addDecodeType set
    up (MyLdpc.cpp:226-552) for one algorithm."""

    def __init__(self, graph, K, max_batch, algo="sp", max_iter=40, llr_scale=8.0, early_term=True,
                 device=0, layer_rows=0, pack_mode=PACK_BYTES, frames_per_lane=0, poll_interval=0,
                 msg_dtype=MSG_F32):
        L = _lib.load()
        cfg = DecoderConfig()
        L.ldpc_decoder_config_init(ctypes.byref(cfg))
        cfg.K, cfg.max_batch = int(K), int(max_batch)
        cfg.algo = ALGOS[algo] if isinstance(algo, str) else int(algo)
        cfg.msg_dtype = {"f32": MSG_F32, "f16": MSG_F16}.get(msg_dtype, msg_dtype)
        cfg.max_iter, cfg.llr_scale = int(max_iter), float(llr_scale)
        cfg.early_term, cfg.device, cfg.layer_rows = int(bool(early_term)), int(device), int(layer_rows)
        cfg.pack_mode, cfg.frames_per_lane, cfg.poll_interval = int(pack_mode), int(frames_per_lane), int(poll_interval)
        self.cfg = cfg
        self.graph = graph
        self.K, self.N, self.E = int(K), graph.N, graph.E
        self._h = ctypes.c_void_p()
        _lib.check(L.ldpc_decoder_create(graph._h, ctypes.byref(cfg), ctypes.byref(self._h)))

    # -- host buffers (the reference's Coder::decode signature) -------------
    def decode(self, llr, want_iters=True):
        """llr: float32 [frames, N] (numpy).  Returns (bytes uint8, iters int32)."""
        L = _lib.load()
        llr = np.ascontiguousarray(llr, np.float32).reshape(-1, self.N)
        frames = llr.shape[0]
        out = np.zeros(out_bytes(self.K, frames, self.cfg.pack_mode), np.uint8)
        iters = np.zeros(frames, np.int32)
        _lib.check(L.ldpc_decode(self._h, llr.ctypes.data, frames, out.ctypes.data, out.size,
                                 iters.ctypes.data if want_iters else None))
        return out, iters

    # -- buffers already in HBM ----------------------------------------------
    def decode_device(self, llr_ptr, frames, out_ptr, out_nbytes, iters_ptr=None, stream=None):
        _lib.check(_lib.load().ldpc_decode_device(self._h, llr_ptr, frames, out_ptr, out_nbytes,
                                                  iters_ptr, stream))

    def set_timing(self, enable=True):
        """True / 1: time every launch of every call; k > 1: of every k-th decode_device call; False: off."""
        _lib.check(_lib.load().ldpc_decoder_set_timing(self._h, int(enable)))

    def stats(self):
        st = DecodeStats()
        _lib.check(_lib.load().ldpc_decoder_stats(self._h, ctypes.byref(st)))
        return {f: getattr(st, f) for f, _ in st._fields_}

    def kernel_times(self):
        """Per-kernel HIP-event times gathered since set_timing(True)."""
        arr = (_lib.KernelTime * 64)()
        n = ctypes.c_int32(0)
        _lib.check(_lib.load().ldpc_decoder_kernel_times(self._h, arr, 64, ctypes.byref(n)))
        return [dict(name=arr[i].name.decode(), phase=arr[i].phase, degree=arr[i].degree,
                     launches=arr[i].launches, ms_total=arr[i].ms_total, bytes_total=arr[i].bytes_total)
                for i in range(n.value)]

    def set_tap(self, it):
        _lib.check(_lib.load().ldpc_decoder_set_tap(self._h, int(it)))

    def dump(self, which, frames):
        per = self.N if which in (2, 3) else self.E
        a = np.empty((frames, per), np.float32)
        _lib.check(_lib.load().ldpc_decoder_dump(self._h, which, a.ctypes.data, a.size))
        return a

    def close(self):
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib._lib.ldpc_decoder_destroy(h)

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
