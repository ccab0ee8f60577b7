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


def unpack_bits(byte_array, K, frames):
    """Inverse of the reference's toChar packing for K % 8 == 0: uint8 [frames, K]."""
    a = np.asarray(byte_array, np.uint8).reshape(frames, K // 8)
    return np.unpackbits(a, axis=1, bitorder="little")
