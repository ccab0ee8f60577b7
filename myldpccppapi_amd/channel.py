"""Seeded test channel: BPSK over AWGN in the reference's convention.

Coder::test (MyLdpc.cpp:1061-1078): bit 0 -> +1.0, bit 1 -> -1.0, plus N(0, sd^2)
noise; Test.cpp:56-57: sd = 10^(-SNR_dB/20).  The reference draws noise from libc
rand() seeded with time(0) (Test.cpp:29), i.e. it is not reproducible; here noise
comes from a counter-based generator (Philox) keyed by (seed, first frame), so any
frame range can be regenerated independently on any rank.
"""
import numpy as np


def snr_db_to_sd(snr_db):
    """Test.cpp:56: sd = 1 / 10^(snr/20)."""
    return float(1.0 / (10.0 ** (snr_db / 20.0)))


def bpsk(bits):
    """0 -> +1.0, 1 -> -1.0 (MyLdpc.cpp:1066-1069)."""
    return (1.0 - 2.0 * np.asarray(bits, np.float32)).astype(np.float32)


def awgn_frames(N, first_frame, frames, sd, seed=20260101, codewords=None):
    """float32 [frames, N] channel values for frames [first_frame, first_frame+frames).

    codewords: optional uint8 [frames, N] bits (default: the all-zero codeword, valid
    for every linear code)."""
    out = np.empty((frames, N), np.float32)
    for i in range(frames):
        g = np.random.Generator(np.random.Philox(key=[seed, first_frame + i]))
        out[i] = g.standard_normal(N, dtype=np.float32) * np.float32(sd)
    if codewords is None:
        out += np.float32(1.0)
    else:
        out += bpsk(codewords)
    return out


def awgn_device(N, first_frame, frames, sd, seed=20260101, codewords=None, device=0, out=None, stream=None):
    """Channel values generated ON THE GPU (ldpc_awgn_device, C ABI): float32 cuda tensor
    [frames, N].  Counter-based noise (csrc/ldpc_channel.h), a different stream of normals than
    awgn_frames() above; frame f depends on (seed, first_frame + f) only.
    codewords: optional uint8 cuda tensor [frames, N] of code bits (default all-zero codeword)."""
    import torch
    from . import _lib
    L = _lib.load()
    dev = torch.device("cuda", device)
    if out is None:
        out = torch.empty((frames, N), dtype=torch.float32, device=dev)
    assert out.is_contiguous() and out.dtype == torch.float32 and out.numel() >= frames * N
    bits_ptr = None
    if codewords is not None:
        assert codewords.is_cuda and codewords.dtype == torch.uint8 and codewords.is_contiguous()
        assert codewords.numel() == frames * N
        bits_ptr = codewords.data_ptr()
    if stream is None:
        stream = torch.cuda.current_stream(dev).cuda_stream
    _lib.check(L.ldpc_awgn_device(out.data_ptr(), frames, N, bits_ptr, float(sd), int(seed), int(first_frame),
                                  int(device), stream))
    return out


def count_errors_device(out_bytes, ref_bytes, frames, device=0, stream=None):
    """(bit errors, byte errors [the reference's ErrNum, Test.cpp:105-110], frame errors) between
    two packed decoder outputs held in cuda uint8 tensors; ref_bytes None = all-zero payload."""
    import ctypes
    import torch
    from . import _lib
    L = _lib.load()
    assert out_bytes.is_cuda and out_bytes.dtype == torch.uint8 and out_bytes.numel() % max(frames, 1) == 0
    per = out_bytes.numel() // max(frames, 1)
    res = (ctypes.c_int64 * 3)()
    if stream is None:
        stream = torch.cuda.current_stream(torch.device("cuda", device)).cuda_stream
    _lib.check(L.ldpc_count_errors_device(out_bytes.data_ptr(), None if ref_bytes is None else ref_bytes.data_ptr(),
                                          frames, per, res, int(device), stream))
    return int(res[0]), int(res[1]), int(res[2])


def unpack_bits(byte_array, K, frames):
    """Inverse of the reference's toChar packing for K % 8 == 0: uint8 [frames, K]."""
    a = np.asarray(byte_array, np.uint8).reshape(frames, K // 8)
    return np.unpackbits(a, axis=1, bitorder="little")
