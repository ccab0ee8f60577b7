"""TEST INFRASTRUCTURE (oracle): an independent systematic LDPC encoder by plain GF(2)
elimination, the checker of `Coder::encode` (csrc/MyLdpc.cpp).

What it pins: the reference's `encodeOnce` (MyLdpc.cpp:633-682) emits the codeword
[K info bits | M parity bits] (LSB-first bytes, :661-680) whose parity part it obtains from
the Richardson-Urbanke split of H (forEncoder, :137-165; Eigen, un-vendored).  H's parity part
[B T; D E] is nonsingular for the six 802.16e seeds, so the parity bits are the UNIQUE solution of
H_p p = H_s s over GF(2): any exact solver must produce the reference's bytes.  This module solves
that system by Gaussian elimination on bit-packed rows -- no knowledge of the dual-diagonal
structure the product encoder exploits.  Parity unpinned against the reference binary itself
(Eigen is absent here).

Only tests/ import this.
"""
import numpy as np


class Gf2Encoder:
    """Solve H [s | p]^T = 0 for p.  rows/cols: H's nonzeros (any order), M x N, K = N - M."""

    def __init__(self, rows, cols, M, N):
        self.M, self.N, self.K = int(M), int(N), int(N) - int(M)
        H = np.zeros((self.M, self.N), np.uint8)
        H[np.asarray(rows), np.asarray(cols)] = 1
        self.Hs = H[:, :self.K]
        # [H_p | I] -> [I | H_p^-1] with rows packed 8 columns per byte
        aug = np.concatenate([H[:, self.K:], np.eye(self.M, dtype=np.uint8)], axis=1)
        a = np.packbits(aug, axis=1)
        for c in range(self.M):
            byte, bit = c >> 3, 0x80 >> (c & 7)
            col = a[:, byte] & bit
            piv = c + int(np.argmax(col[c:] != 0))
            if not col[piv]:
                raise ValueError("parity part of H is singular (column %d)" % c)
            if piv != c:
                a[[c, piv]] = a[[piv, c]]
                col = a[:, byte] & bit
            mask = col != 0
            mask[c] = False
            a[mask] ^= a[c]
        inv = np.unpackbits(a, axis=1)[:, self.M:2 * self.M]
        self.Hp_inv = inv.astype(np.uint8)

    def parity(self, info_bits):
        """info_bits: uint8 [K] (0/1) -> parity bits uint8 [M]."""
        s = np.asarray(info_bits, np.uint8)
        lam = (self.Hs.astype(np.int64) @ s.astype(np.int64)) & 1
        return ((self.Hp_inv.astype(np.int64) @ lam) & 1).astype(np.uint8)

    def encode_bytes(self, src, src_len=None):
        """One frame as the reference lays it out (MyLdpc.cpp:639-680): the first K/8 source bytes
        (short input zero-padded) as LSB-first info bits, then the parity bits from bit K on;
        returns N/8 bytes."""
        kb = self.K // 8
        src = np.frombuffer(bytes(src), np.uint8)[:kb if src_len is None else min(kb, src_len)]
        info = np.zeros(self.K, np.uint8)
        info[:src.size * 8] = np.unpackbits(src, bitorder="little")
        code = np.concatenate([info, self.parity(info)])
        # the reference copies the source bytes verbatim and ORs the parity bits in from bit K
        # (when K % 8 != 0 the info bits K-K%8..K-1 are never read from the source: zero)
        return np.packbits(code, bitorder="little")
