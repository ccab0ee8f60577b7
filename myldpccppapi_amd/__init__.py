"""myldpccppapi_amd -- MI355X-native batched LDPC decoding behind the API of
wing02/MyLdpcCppApi's `Coder` class.

Product layout:
  csrc/            HIP kernels + C ABI (libldpc_hip.so) + the C++ `Coder` (libmyldpc.so)
  capi.py          ctypes objects over the C ABI (Graph, Decoder); the C++ `Coder` of
                   include/MyLdpc.h is the mirror of the reference's class (there is no Python one)
  codes.py         parity-check matrices (802.16e seeds, DVB-S2 / 5G-NR profile codes)
  channel.py       seeded BPSK/AWGN test channel
  sharding.py      frame sharding over ranks + gather of decoded bytes (torch.distributed)

Nothing here imports oracle/ (test infrastructure).
"""
from .capi import (ALGO_LAYERED, ALGO_MS, ALGO_SP, PACK_BITS, PACK_BYTES, Decoder, Graph, LdpcError,  # noqa: F401
                   device_count, out_bytes, shard_range)
