"""Thin Python objects over the C ABI (include/ldpc_hip.h): Graph and Decoder.

Host-side plumbing only -- every decode runs in the HIP kernels of
libldpc_hip.so.  numpy arrays are passed as host pointers, integers (e.g. a
torch tensor's data_ptr()) as device pointers.
"""
import ctypes

import numpy as np

from . import _lib
from ._lib import DecoderConfig, DecodeStats, LdpcError  # noqa: F401

ALGO_SP, ALGO_MS, ALGO_LAYERED, ALGO_MS_FUSED, ALGO_LAYERED_HOST = 0, 1, 2, 3, 4
MSG_F32, MSG_F16 = 0, 1
PACK_BYTES, PACK_BITS = 0, 1
HOST_INPUT = {"auto": 0, "staged": 1, "lock_pages": 2}     # enum ldpc_host_input
ALGOS = {"sp": ALGO_SP, "ms": ALGO_MS, "layered": ALGO_LAYERED, "ms_fused": ALGO_MS_FUSED,
         "layered_host": ALGO_LAYERED_HOST}

# two-bit fields of ldpc_decoder_config.tune_flags (enum ldpc_tune_field): True forces on, False off
TUNE_FIELDS = {"fused": 0, "ldsp": 2, "ldsp_ext": 4, "ldsp_pack": 6, "link_narrow": 8, "check_wide": 10,
               "syn_xcd": 12, "fused_pack": 14, "fused_loop": 16, "device_tail": 18, "merge": 20, "link_deep": 22, "link_half": 24,
               "link_guided": 26, "tiles_first": 28}
TUNE_INTS = ("rows_per_wave", "cols_per_wave", "link_rows", "compact", "ldsp_grid", "ldsp_per_cu", "ldsp_waves", "place", "q_order")


def apply_tune(cfg, tune):
    """Fill the tuning fields of a DecoderConfig from a dict: tri-state names of TUNE_FIELDS
    (True / False / None = automatic) and integers of TUNE_INTS (link_rows=-1: column-local fusion
    off; compact=-1: tail compaction off).  Kernel selection and launch shapes only."""
    flags, shape = 0, 0
    for k, v in (tune or {}).items():
        if k in TUNE_FIELDS:
            if v is not None:
                flags |= (1 if v else 2) << TUNE_FIELDS[k]
        elif k == "ldsp_per_cu":
            shape |= int(v) & 255
        elif k == "ldsp_waves":
            shape |= (int(v) & 255) << 8
        elif k in TUNE_INTS:
            setattr(cfg, "tune_" + k, int(v))
        else:
            raise KeyError("unknown tuning field %r" % k)
    cfg.tune_flags, cfg.tune_ldsp_shape = flags, shape


def hbm_probe(device=0, nbytes=1 << 30, reps=5, by_policy=False):
    """GB/s of a plain float4 copy on `device` (read + write), best of `reps` launches with the default
    cache policy and `reps` with non-temporal loads and stores; by_policy=True: (best, default, nt)."""
    g = ctypes.c_double(0.0)
    two = (ctypes.c_double * 2)(0.0, 0.0)
    _lib.check(_lib.load().ldpc_hbm_probe_device(int(device), int(nbytes), int(reps), ctypes.byref(g), two))
    return (g.value, two[0], two[1]) if by_policy else g.value


def hbm_sustained(device=0, nbytes=1 << 30, milliseconds=300):
    """GB/s of the non-temporal float4 copy run back to back for `milliseconds` (first third untimed)."""
    g = ctypes.c_double(0.0)
    _lib.check(_lib.load().ldpc_hbm_sustained_device(int(device), int(nbytes), int(milliseconds), ctypes.byref(g)))
    return g.value


def tune_from_env(env=None):
    """For the measurement scripts under tools/: translate LDPC_TUNE_* environment switches into
    a tuning dict for Decoder(tune=...).  The library itself reads no environment variables."""
    import os
    env = os.environ if env is None else env
    t = {}
    for name, key in (("FUSED", "fused"), ("LDSP", "ldsp"), ("LDSP_EXT", "ldsp_ext"), ("LDSP_PACK", "ldsp_pack"),
                      ("LINK_NARROW", "link_narrow"), ("CHECK_WIDE", "check_wide"), ("SYN_XCD", "syn_xcd"),
                      ("FUSED_LOOP", "fused_loop"), ("DEVICE_TAIL", "device_tail"), ("MERGE", "merge"), ("LINK_DEEP", "link_deep"), ("LINK_HALF", "link_half"),
                      ("LINK_GUIDED", "link_guided"), ("TILES_FIRST", "tiles_first")):
        if "LDPC_TUNE_" + name in env:
            t[key] = int(env["LDPC_TUNE_" + name]) != 0
    if "LDPC_TUNE_NO_PACK" in env:
        t["fused_pack"] = False
    for name, key in (("RPW", "rows_per_wave"), ("CPW", "cols_per_wave"), ("LDSP_GRID", "ldsp_grid"),
                      ("LDSP_PER_CU", "ldsp_per_cu"), ("LDSP_WAVES", "ldsp_waves")):
        if "LDPC_TUNE_" + name in env:
            t[key] = int(env["LDPC_TUNE_" + name])
    for name, key in (("LINK_RPW", "link_rows"), ("COMPACT", "compact")):     # 0 meant "off"
        if "LDPC_TUNE_" + name in env:
            v = int(env["LDPC_TUNE_" + name])
            t[key] = v if v > 0 else -1
    return t


def shard_range(frames, part, parts, unit=1):
    """ldpc_shard_range: frames [lo, hi) of part `part` of `parts` (boundaries multiples of `unit`)."""
    lo, hi = ctypes.c_int64(), ctypes.c_int64()
    _lib.check(_lib.load().ldpc_shard_range(int(frames), int(part), int(parts), int(unit),
                                            ctypes.byref(lo), ctypes.byref(hi)))
    return lo.value, hi.value


def host_block_plan(base, frames, N, max_batch, group):
    """ldpc_host_block_plan: (s0, s1, b0, b1, body_end, whole_by_cpu) of launch group `group` in lock mode."""
    out = (ctypes.c_uint64 * 6)()
    _lib.check(_lib.load().ldpc_host_block_plan(int(base), int(frames), int(N), int(max_batch), int(group), out))
    return tuple(int(x) for x in out)


def host_locked_ranges():
    """(live, stale): ranges of caller memory the library holds page-locked now / failed to release."""
    a, b = ctypes.c_int64(-1), ctypes.c_int64(-1)
    _lib.check(_lib.load().ldpc_host_locked_ranges(ctypes.byref(a), ctypes.byref(b)))
    return a.value, b.value


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
                 msg_dtype=MSG_F32, tune=None, devices=None, host_input="auto", host_copy_threads=0):
        L = _lib.load()
        cfg = DecoderConfig()
        L.ldpc_decoder_config_init(ctypes.byref(cfg))
        cfg.K, cfg.max_batch = int(K), int(max_batch)
        cfg.algo = ALGOS[algo] if isinstance(algo, str) else int(algo)
        cfg.msg_dtype = {"f32": MSG_F32, "f16": MSG_F16}.get(msg_dtype, msg_dtype)
        cfg.max_iter, cfg.llr_scale = int(max_iter), float(llr_scale)
        cfg.early_term, cfg.device, cfg.layer_rows = int(bool(early_term)), int(device), int(layer_rows)
        cfg.pack_mode, cfg.frames_per_lane, cfg.poll_interval = int(pack_mode), int(frames_per_lane), int(poll_interval)
        apply_tune(cfg, tune)
        # how ldpc_decode() moves pageable channel values: through the library's pinned ring (default) or by
        # page-locking the caller's pages for the call (include/ldpc_hip.h: enum ldpc_host_input)
        cfg.host_input = HOST_INPUT[host_input] if isinstance(host_input, str) else int(host_input)
        cfg.host_copy_threads = int(host_copy_threads)
        self.cfg = cfg
        self.graph = graph
        self.K, self.N, self.E = int(K), graph.N, graph.E
        self._h = ctypes.c_void_p()
        if devices is None:
            _lib.check(L.ldpc_decoder_create(graph._h, ctypes.byref(cfg), ctypes.byref(self._h)))
        else:       # one handle over several devices (ldpc_decoder_create_multi): host-buffer decode only
            devs = (ctypes.c_int32 * len(devices))(*[int(x) for x in devices])
            _lib.check(L.ldpc_decoder_create_multi(graph._h, ctypes.byref(cfg), devs, len(devices),
                                                   ctypes.byref(self._h)))

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
                     launches=arr[i].launches, ms_total=arr[i].ms_total, bytes_total=arr[i].bytes_total,
                     bytes_moved=arr[i].bytes_moved)
                for i in range(n.value)]

    def link_form(self):
        """Form of the column-fused check kernel (None: the code has none) and the creation-time measurement."""
        form, cal = ctypes.c_int32(-1), ctypes.c_int32(0)
        ms = (ctypes.c_float * 3)()
        _lib.check(_lib.load().ldpc_decoder_link_form(self._h, ctypes.byref(form), ctypes.byref(cal), ms))
        if form.value < 0:
            return None
        return {"form": ("wide", "narrow", "half")[form.value], "chosen_by_measurement": bool(cal.value),
                "ms_per_launch": {"wide": round(ms[0], 4), "narrow": round(ms[1], 4), "half": round(ms[2], 4)}}

    def placement(self):
        """The creation-time placement search: per candidate set of arrays the column-fused check kernel's time per
        launch, and which one was kept (None: no search was made)."""
        n, kept = ctypes.c_int32(0), ctypes.c_int32(0)
        ms = (ctypes.c_float * 16)()
        _lib.check(_lib.load().ldpc_decoder_placement(self._h, ctypes.byref(n), ctypes.byref(kept), ms))
        if n.value == 0:
            return None
        return {"candidates_ms": [round(ms[i], 4) for i in range(n.value)], "kept": kept.value}

    def array_addresses(self):
        """Device addresses of the streaming decoder's Q, R, channel and hard-bit arrays."""
        a = (ctypes.c_uint64 * 4)()
        _lib.check(_lib.load().ldpc_decoder_array_addresses(self._h, a))
        return {"Q": int(a[0]), "R": int(a[1]), "chan": int(a[2]), "hard": int(a[3])}

    def set_tap(self, it):
        _lib.check(_lib.load().ldpc_decoder_set_tap(self._h, int(it)))

    def dump(self, which, frames):
        per = self.N if which in (2, 3) else self.E
        a = np.empty((frames, per), np.float32)
        _lib.check(_lib.load().ldpc_decoder_dump(self._h, which, a.ctypes.data, a.size))
        return a

    def close(self):
        """ldpc_decoder_destroy; raises if the library reports page-locked blocks it could not release."""
        h, self._h = getattr(self, "_h", None), None
        if h and _lib is not None and _lib._lib is not None:
            _lib.check(_lib._lib.ldpc_decoder_destroy(h))

    def __del__(self):
        try:
            self.close()
        except Exception:
            pass
