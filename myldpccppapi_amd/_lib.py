"""Loader for libldpc_hip.so (the C ABI of include/ldpc_hip.h).

The library is built in-tree by `__graft_entry__.build()` / `make -C
myldpccppapi_amd/csrc`.  There is no fallback: if it is missing, loading fails
loudly, and every compute entry point fails without a HIP device.
"""
import ctypes
import os
import sys

_HERE = os.path.dirname(os.path.abspath(__file__))
# LDPC_HIP_LIB: load another build of the same library (kernel-variant experiments); read by this
# Python loader only -- the library itself reads no environment variables
LIB_PATH = os.environ.get("LDPC_HIP_LIB") or os.path.join(_HERE, "libldpc_hip.so")
_lib = None


class LdpcError(RuntimeError):
    def __init__(self, code, message):
        super().__init__("ldpc_hip error %d: %s" % (code, message))
        self.code = code


class DecoderConfig(ctypes.Structure):
    """Mirror of `ldpc_decoder_config` (include/ldpc_hip.h)."""
    _fields_ = [("struct_size", ctypes.c_uint32), ("K", ctypes.c_int32),
                ("max_batch", ctypes.c_int32), ("algo", ctypes.c_int32),
                ("msg_dtype", ctypes.c_int32), ("max_iter", ctypes.c_int32),
                ("llr_scale", ctypes.c_float), ("early_term", ctypes.c_int32),
                ("device", ctypes.c_int32), ("layer_rows", ctypes.c_int32),
                ("pack_mode", ctypes.c_int32), ("frames_per_lane", ctypes.c_int32),
                ("poll_interval", ctypes.c_int32),
                ("tune_flags", ctypes.c_int32), ("tune_rows_per_wave", ctypes.c_int32),
                ("tune_cols_per_wave", ctypes.c_int32), ("tune_link_rows", ctypes.c_int32),
                ("tune_compact", ctypes.c_int32), ("tune_ldsp_grid", ctypes.c_int32),
                ("tune_ldsp_shape", ctypes.c_int32), ("tune_place", ctypes.c_int32),
                ("host_input", ctypes.c_int32),
                ("host_copy_threads", ctypes.c_int32), ("tune_q_order", ctypes.c_int32)]


class DecodeStats(ctypes.Structure):
    """Mirror of `ldpc_decode_stats`."""
    _fields_ = [("iterations_launched", ctypes.c_int32), ("batch_time", ctypes.c_int32),
                ("frames", ctypes.c_int64), ("frames_converged", ctypes.c_int64),
                ("ms_total", ctypes.c_float), ("ms_check", ctypes.c_float),
                ("ms_var", ctypes.c_float), ("ms_other", ctypes.c_float),
                ("launches_check", ctypes.c_int32), ("launches_var", ctypes.c_int32),
                ("frame_rounds", ctypes.c_int64)]


class KernelTime(ctypes.Structure):
    """Mirror of `ldpc_kernel_time`."""
    _fields_ = [("phase", ctypes.c_int32), ("degree", ctypes.c_int32), ("launches", ctypes.c_int32),
                ("ms_total", ctypes.c_float), ("bytes_total", ctypes.c_int64), ("name", ctypes.c_char * 64),
                ("bytes_moved", ctypes.c_int64)]


#: every symbol include/ldpc_hip.h declares
EXPORTS = (
    "ldpc_abi_version", "ldpc_last_error", "ldpc_device_count", "ldpc_graph_create",
    "ldpc_graph_destroy", "ldpc_graph_info", "ldpc_decoder_config_init", "ldpc_decoder_create",
    "ldpc_decoder_create_multi", "ldpc_shard_range", "ldpc_decoder_destroy", "ldpc_decode", "ldpc_decode_device", "ldpc_out_bytes",
    "ldpc_decoder_set_timing", "ldpc_decoder_stats", "ldpc_decoder_kernel_times", "ldpc_decoder_set_tap",
    "ldpc_decoder_dump", "ldpc_awgn_device", "ldpc_count_errors_device", "ldpc_hbm_probe_device", "ldpc_hbm_sustained_device",
    "ldpc_host_block_plan", "ldpc_host_locked_ranges", "ldpc_decoder_link_form", "ldpc_decoder_placement", "ldpc_decoder_array_addresses",
)


def load():
    """dlopen the library and declare the prototypes.  Raises OSError if the
    library has not been built."""
    global _lib
    if _lib is not None:
        return _lib
    if not os.path.exists(LIB_PATH):
        raise OSError("%s not built: run `python -c 'import __graft_entry__ as g; g.build()'` "
                      "or `make -C myldpccppapi_amd/csrc`" % LIB_PATH)
    # One HIP runtime per process: PyTorch-ROCm ships its own libamdhip64 (same
    # soname).  If torch is importable, let it load first so both share it.
    if "torch" not in sys.modules:
        try:
            import torch  # noqa: F401
        except Exception:
            pass
    L = ctypes.CDLL(LIB_PATH, mode=ctypes.RTLD_GLOBAL)
    i32p = ctypes.POINTER(ctypes.c_int32)
    i64p = ctypes.POINTER(ctypes.c_int64)
    vp = ctypes.c_void_p
    L.ldpc_abi_version.restype = ctypes.c_int
    L.ldpc_last_error.restype = ctypes.c_char_p
    L.ldpc_device_count.argtypes = [ctypes.POINTER(ctypes.c_int)]
    L.ldpc_graph_create.argtypes = [i32p, i32p, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32,
                                    ctypes.POINTER(vp)]
    L.ldpc_graph_destroy.argtypes = [vp]
    L.ldpc_graph_info.argtypes = [vp, i32p, i32p, i64p, i32p, i32p]
    L.ldpc_decoder_config_init.argtypes = [ctypes.POINTER(DecoderConfig)]
    L.ldpc_decoder_config_init.restype = None
    L.ldpc_decoder_create.argtypes = [vp, ctypes.POINTER(DecoderConfig), ctypes.POINTER(vp)]
    L.ldpc_decoder_create_multi.argtypes = [vp, ctypes.POINTER(DecoderConfig), i32p, ctypes.c_int32,
                                            ctypes.POINTER(vp)]
    L.ldpc_shard_range.argtypes = [ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int32, i64p, i64p]
    L.ldpc_decoder_destroy.argtypes = [vp]
    L.ldpc_decode.argtypes = [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp]
    L.ldpc_decode_device.argtypes = [vp, vp, ctypes.c_int64, vp, ctypes.c_int64, vp, vp]
    L.ldpc_out_bytes.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32]
    L.ldpc_out_bytes.restype = ctypes.c_int64
    L.ldpc_decoder_set_timing.argtypes = [vp, ctypes.c_int]
    L.ldpc_decoder_stats.argtypes = [vp, ctypes.POINTER(DecodeStats)]
    L.ldpc_decoder_kernel_times.argtypes = [vp, ctypes.POINTER(KernelTime), ctypes.c_int32, i32p]
    L.ldpc_decoder_set_tap.argtypes = [vp, ctypes.c_int32]
    L.ldpc_decoder_dump.argtypes = [vp, ctypes.c_int32, vp, ctypes.c_int64]
    L.ldpc_awgn_device.argtypes = [vp, ctypes.c_int64, ctypes.c_int32, vp, ctypes.c_float, ctypes.c_uint64,
                                   ctypes.c_int64, ctypes.c_int32, vp]
    L.ldpc_count_errors_device.argtypes = [vp, vp, ctypes.c_int64, ctypes.c_int64, i64p, ctypes.c_int32, vp]
    L.ldpc_hbm_sustained_device.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32, ctypes.POINTER(ctypes.c_double)]
    L.ldpc_hbm_probe_device.argtypes = [ctypes.c_int32, ctypes.c_int64, ctypes.c_int32,
                                        ctypes.POINTER(ctypes.c_double), ctypes.POINTER(ctypes.c_double)]
    L.ldpc_host_block_plan.argtypes = [ctypes.c_uint64, ctypes.c_int64, ctypes.c_int32, ctypes.c_int32, ctypes.c_int64,
                                       ctypes.POINTER(ctypes.c_uint64)]
    L.ldpc_host_locked_ranges.argtypes = [i64p, i64p]
    L.ldpc_decoder_array_addresses.argtypes = [vp, ctypes.POINTER(ctypes.c_uint64)]
    L.ldpc_decoder_placement.argtypes = [vp, i32p, i32p, ctypes.POINTER(ctypes.c_float)]
    L.ldpc_decoder_link_form.argtypes = [vp, i32p, i32p, ctypes.POINTER(ctypes.c_float)]
    _lib = L
    return L


def check(rc):
    if rc != 0:
        raise LdpcError(rc, load().ldpc_last_error().decode("utf-8", "replace"))
