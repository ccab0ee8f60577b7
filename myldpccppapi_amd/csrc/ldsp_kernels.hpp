/*
 * ldsp_kernels.hpp -- layered min-sum (the schedule of decodeOnceTDMP, decodeCL.c:307-426) for
 * quasi-cyclic codes that are too large for the LDS-resident kernels of fused_kernels.hpp and
 * for which one launch per layer over HBM (layered_kernels.hpp) is the wrong design as well:
 * 5G-NR BG1 at Z = 384 (N = 26112, E = 121344, 46 layers) is the case it was written for.
 *
 * Data placement, per frame (one workgroup = one frame at a time, lanes = the z rows of a layer):
 *   - posteriors of the block columns met by two or more layers live in LDS ([slot][z] floats,
 *     read and written at consecutive addresses: column = slot*z + (row + shift) mod z);
 *   - a block column met by ONE layer only (the 42 extension-parity columns of BG1) is touched
 *     by exactly one row per iteration, so its posterior travels with that row's record instead
 *     of occupying LDS: BG1 Z = 384 needs 26 * 384 * 4 B = 40 KB instead of 104 KB and four
 *     frames share a CU's 160 KB;
 *   - the check-to-variable messages are not stored as E floats.  A min-sum row sends two
 *     magnitudes only, R_k = cl_sign(q_k) * (k == argmin ? ac : ab), so the row's state is the
 *     16-byte record
 *         x = |ab|   y = |ac|   z = sign bits of R_0..R_d-1 | argmin << 24 | irregular << 29
 *         w = posterior of the row's single-layer column (if it has one)
 *     from which the next iteration rebuilds every R_k bit for bit (see ldsp_old_message).
 *     Row r of a layer is always lane r of the same workgroup: a record is read and written by
 *     one thread only, as one 16-byte load and store per row and iteration, requested one layer
 *     step ahead.  Workgroups are persistent and walk over the frames, so the record rings
 *     (M * 16 B per resident workgroup) stay in L2 / Infinity Cache: HBM sees the channel values
 *     once and the packed bits once.
 * Kernels in this file: layered_ldsp_kernel (+ _packed_: 64/z frames per wave for z <= 32) -- layered
 * min-sum; flood_ldsp_kernel<.., CHAIN> (+ _packed_) -- flooding min-sum with the MS kernel chain's
 * arithmetic (DecodeMS / DecodeCPU) or the fused reference kernel's (DecodeMSCL), two posterior images.
 * Arithmetic: the operations of layer_kernel / the oracle in the same order; where every q of a
 * wave's rows is a regular number (not zero, not NaN, product not underflowed) the sign algebra
 * is done on the bit patterns -- identical results, a third of the instructions; otherwise the
 * wave takes ldsp_row_any, which performs the reference's operations one by one.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <type_traits>
#include <vector>

#include "fused_kernels.hpp"

namespace ldpc {

constexpr int kLdspMaxDeg = 24;
constexpr int kLdspPackStride = 4 * kLdspMaxDeg;   /* LdspArgs::pack, ints per layer */
constexpr uint32_t kLdspIrregular = 1u << 29;
constexpr size_t kLdspMaxLds = 160 * 1024 - 512;     /* one workgroup's dynamic LDS */

struct LdspArgs {
    const float *__restrict__ llr;        /* [frames][N] */
    uint8_t *__restrict__ out;            /* packed bytes, toChar layout */
    int32_t *__restrict__ iters;          /* [frames] or nullptr */
    int32_t *__restrict__ summary;        /* [2]: max iters, converged frames */
    float *__restrict__ dump_p;           /* [frames][N] or nullptr (taps) */
    float *__restrict__ dump_r;           /* [frames][E] reference edge order, or nullptr */
    uint4 *__restrict__ recs;             /* [grid][layers*z] check records */
    uint32_t *__restrict__ zf;            /* [grid][layers*z] "message is a zero" bits of irregular records */
    const int32_t *__restrict__ hdr;      /* [layers][4]: LDS entries, has external column, its first index, its shift */
    const int32_t *__restrict__ pack;     /* [layers][4][24]: byte offset of the entry's LDS column (slot*z*4); 4 * its
                                             shift; their sum; 4 * (z - shift): ready-made operands, no scalar
                                             arithmetic per edge (ldsp_at uses the first two rows, ldsp_at_abs the others) */
    const int32_t *__restrict__ col_slot; /* [N/z]: LDS slot of the block column, -1 = travels with a record */
    const int32_t *__restrict__ layer_e0; /* [layers]: edge id of the layer's first edge */
    int64_t frames, out_bytes;
    int32_t N, E, K, z, layers, nb, lds_cols, max_iter, rounds, early_term;
};

/* The kernel stores to global memory inside its layer loop, so the compiler may not assume that
 * the code tables are unchanged and would fetch them with per-lane vector loads and a memory
 * round trip per layer step; read through the constant address space they stay scalar loads
 * (the tables are written once, by the host, before any launch). */
typedef const int32_t __attribute__((address_space(4))) *ldpc_const_i32;
__device__ __forceinline__ ldpc_const_i32 as_constant(const int32_t *p) { return (ldpc_const_i32)(uintptr_t)p; }

/* Workgroup barrier that orders LDS traffic only: the global loads / stores of this kernel are
 * private to their thread, and waiting for them (as __syncthreads() does) would put a memory
 * round trip into every layer step. */
__device__ __forceinline__ void lds_barrier()
{
    asm volatile("s_waitcnt lgkmcnt(0)\n\ts_barrier" ::: "memory");
}

__device__ __forceinline__ bool ldsp_regular(uint32_t bits)      /* finite non-zero, or infinite */
{
    return ((bits & 0x7fffffffu) - 1u) < 0x7f800000u;
}

__device__ __forceinline__ int ldsp_wrap(int r, int shift, int z)   /* (r + shift) mod z, r and shift < z */
{
    const uint32_t t = (uint32_t)r + (uint32_t)shift;
    const uint32_t tw = t - (uint32_t)z;                            /* wraps to a huge value when t < z */
    return (int)(t < tw ? t : tw);
}

/* address of P[column][(r + shift) mod z] from ready-made byte quantities: r4 = 4r, shift4 = 4*shift,
 * z4 = 4z, column_bytes = 4*slot*z -- add, subtract, unsigned minimum, add */
__device__ __forceinline__ float *ldsp_at(float *P, int column_bytes, int shift4, int r4, int z4)
{
    const uint32_t t = (uint32_t)r4 + (uint32_t)shift4;
    const uint32_t tw = t - (uint32_t)z4;                           /* wraps to a huge value when t < z4 */
    return reinterpret_cast<float *>(reinterpret_cast<char *>(P) + ((t < tw ? t : tw) + (uint32_t)column_bytes));
}

/* The same address when the posteriors start at LDS address 0 (a kernel without static LDS), from the table's
 * other two rows: r4 + (column_bytes + shift4), less z4 for the rows that wrap (r4 >= z4 - shift4) --
 * compare, select, three-operand add; an LDS pointer made from the number, no base to add */
typedef __attribute__((address_space(3))) float ldpc_lds_float;
__device__ __forceinline__ ldpc_lds_float *ldsp_at_abs(int colshift, int thresh, int r4, int negz4)
{
    const uint32_t wrap = (uint32_t)r4 >= (uint32_t)thresh ? (uint32_t)negz4 : 0u;
    uint32_t at;                    /* the compiler makes two adds of this sum */
    asm("v_add3_u32 %0, %1, %2, %3" : "=v"(at) : "v"(r4), "s"(colshift), "v"(wrap));
    return reinterpret_cast<ldpc_lds_float *>((uintptr_t)at);
}

/* Message k of a d-entry record: cl_sign(q) is +-1 for a regular q, so R = +-sel; it is +-0 for
 * q = +-0 and +0 for NaN, so R is a zero then (zf bit).  sel = sa*min is finite always. */
__device__ __forceinline__ uint32_t ldsp_old_message(const uint4 rec, uint32_t zf, int k, int d)
{
    const uint32_t mag = ((zf >> k) & 1u) ? 0u : ((k == (int)((rec.z >> 24) & 31u)) ? rec.y : rec.x);
    return mag | (((rec.z >> (d - 1 - k)) & 1u) << 31);
}

/* Any input: the reference's operations one by one (decodeCL.c:345-383) with run-time loops, q
 * parked in P between the two passes as the reference does. */
__device__ __forceinline__ uint4 ldsp_row_any(float *P, ldpc_const_i32 pk, int dl, int ext, int z, int r,
                                              const uint4 old, uint32_t *zfp, uint32_t *par)
{
    *par = 0;
    const int d = dl + ext;
    const uint32_t ozf = (old.z & kLdspIrregular) ? *zfp : 0u;
    float prod = 1.0f, b = 1000.0f, c = 1001.0f, qext = 0.0f;
    int bind = 31;
    for (int k = 0; k < d; ++k) {
        const float rold = __uint_as_float(ldsp_old_message(old, ozf, k, d));
        float q;
        if (k < dl) {
            float *p = ldsp_at(P, pk[k], pk[kLdspMaxDeg + k], r * 4, z * 4);
            q = *p - rold;
            *p = q;
        } else {
            q = __uint_as_float(old.w) - rold;
            qext = q;
        }
        prod *= q;
        const float mag = __builtin_fabsf(q);
        if (mag <= b) { c = b; b = mag; bind = k; }
        else if (mag > b && mag <= c) { c = mag; }
    }
    const float sa = cl_sign(prod);
    const float ab = sa * b, ac = sa * c;
    const uint32_t mab = __float_as_uint(ab) & 0x7fffffffu, mac = __float_as_uint(ac) & 0x7fffffffu;
    uint32_t signs = 0, zf = 0, pext = 0;
    for (int k = 0; k < d; ++k) {
        float *p = P;
        float q = qext;
        if (k < dl) {
            p = ldsp_at(P, pk[k], pk[kLdspMaxDeg + k], r * 4, z * 4);
            q = *p;
        }
        const float rn = cl_sign(q) * ((k == bind) ? ac : ab);
        const float pn = q + rn;
        if (k < dl) *p = pn;
        else pext = __float_as_uint(pn);
        if (pn < 0.0f) *par ^= 1u;
        const uint32_t rb = __float_as_uint(rn);
        signs = (signs << 1) | (rb >> 31);
        if ((rb & 0x7fffffffu) != ((k == bind) ? mac : mab)) zf |= 1u << k;   /* then it is a zero */
    }
    uint32_t word = signs | ((uint32_t)bind << 24);
    if (zf) {
        word |= kLdspIrregular;
        *zfp = zf;
    }
    return uint4{mab, mac, word, pext};
}

/* Exact row width (DL entries in LDS + EXT external), straight-line code.  Returns false,
 * wave-uniformly and with P untouched, when some row of the wave needs ldsp_row_any.  ABS: the
 * posteriors start at LDS address 0 (ldsp_at_abs).  PAR = false: *par is left alone (the caller takes
 * the parity of the one layer it needs from the posteriors, ldsp_row_parity). */
template <int DL, int EXT, bool ABS = false, bool PAR = true>
__device__ __forceinline__ bool ldsp_row(float *P, ldpc_const_i32 pk, int z, int r, const uint4 old, uint4 *out,
                                         uint32_t *par)
{
    constexpr int D = DL + EXT;
    constexpr int DLA = DL > 0 ? DL : 1;
    if (__ballot((old.z & kLdspIrregular) != 0u) != 0ull) return false;
    float q[D];
    typedef typename std::conditional<ABS, ldpc_lds_float, float>::type cell;     /* ABS: P is LDS address 0 */
    cell *at[DLA];
#pragma unroll
    for (int k = 0; k < DL; ++k) {
        if constexpr (ABS) at[k] = ldsp_at_abs(pk[2 * kLdspMaxDeg + k], pk[3 * kLdspMaxDeg + k], r * 4, -(z * 4));
        else at[k] = ldsp_at(P, pk[k], pk[kLdspMaxDeg + k], r * 4, z * 4);
    }
    const int obind = (int)((old.z >> 24) & 31u);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const uint32_t sel = (k == obind) ? old.y : old.x;
        const uint32_t rold = ((old.z << (31 - (D - 1 - k))) & 0x80000000u) | sel;
        q[k] = (k < DL ? *at[k < DL ? k : 0] : __uint_as_float(old.w)) - __uint_as_float(rold);
    }
    float prod = 1.0f, b = 1000.0f, c = 1001.0f;
    int bind = 31;                                                  /* none (all |q| > 1000) */
#pragma unroll
    for (int k = 0; k < D; ++k) {
        prod *= q[k];
        const float mag = __builtin_fabsf(q[k]);
        const bool le = mag <= b;
        bind = le ? k : bind;
        c = __builtin_amdgcn_fmed3f(b, mag, c);                     /* b <= c: the branches of decodeCL.c:359-365 */
        b = __builtin_amdgcn_fmed3f(0.0f, mag, b);                  /* min(b, mag): both >= 0 */
    }
    const uint32_t pb = __float_as_uint(prod);
    if (__ballot(!ldsp_regular(pb)) != 0ull) return false;
    const uint32_t ps = pb & 0x80000000u;                           /* cl_sign(prod) = +-1 */
    const uint32_t mb = __float_as_uint(b), mc = __float_as_uint(c);
    uint32_t signs = 0, pext = 0, px = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const uint32_t sel = (k == bind) ? mc : mb;
        const uint32_t rn = ((__float_as_uint(q[k]) ^ ps) & 0x80000000u) | sel;
        const float pn = q[k] + __uint_as_float(rn);
        if (k < DL) *at[k < DL ? k : 0] = pn;
        else pext = __float_as_uint(pn);
        signs = __builtin_amdgcn_alignbit(signs, rn, 31);           /* (signs << 1) | sign(rn) */
        if (PAR) px ^= __float_as_uint(pn);                         /* regular q, sel > 0: pn < 0 <=> sign bit */
    }
    *out = uint4{mb, mc, signs | ((uint32_t)bind << 24), pext};
    *par = px >> 31;
    return true;
}

/* parities of the hard decisions P < 0 over the LDS entries of the wave's rows, as a lane mask */
template <int DL>
__device__ __forceinline__ uint64_t ldsp_row_parity(const float *P, ldpc_const_i32 pk, int z, int r)
{
    constexpr int DLA = DL > 0 ? DL : 1;
    float v[DLA];
#pragma unroll
    for (int k = 0; k < DL; ++k) v[k] = *ldsp_at(const_cast<float *>(P), pk[k], pk[kLdspMaxDeg + k], r * 4, z * 4);
    uint64_t par = 0;
#pragma unroll
    for (int k = 0; k < DL; ++k) par ^= __ballot(v[k] < 0.0f);
    return par;
}

#define LDPC_LDSP_WIDTHS(X) X(0) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) \
    X(16) X(17) X(18) X(19) X(20) X(21) X(22) X(23)

/* Register budget: at BG1 Z = 384 a workgroup is 6 waves and 40 KB of LDS.  The dispatcher places
 * a workgroup's waves 2-2-1-1 on the four SIMDs, so three workgroups per CU need room for 6 waves
 * on a SIMD, i.e. at most 80 VGPRs (measured: 96 VGPRs -> two workgroups per CU, 30 ms instead of
 * 25 ms per batch; the widest rows, 19 + 1 entries, spill a few registers at 80). */
#ifndef LDPC_LDSP_WAVES_PER_EU
#define LDPC_LDSP_WAVES_PER_EU 6
#endif

template <int MAXW>
__global__ __launch_bounds__(64 * MAXW) __attribute__((amdgpu_waves_per_eu(LDPC_LDSP_WAVES_PER_EU)))
void layered_ldsp_kernel(const LdspArgs a)
{
    extern __shared__ float lds[];
    float *P = lds;                                                             /* [lds_cols][z] */
    const int r = (int)threadIdx.x, LANES = (int)blockDim.x;
    const int z = a.z;
    uint32_t *wg_flag = reinterpret_cast<uint32_t *>(lds + (((size_t)a.lds_cols * z + 1) & ~(size_t)1));
    const bool row = r < z;
    const size_t ring = (size_t)blockIdx.x * ((size_t)a.layers * z) + r;        /* [layer][z], mine: + r */
    uint4 *recs = a.recs + ring;
    uint32_t *zfs = a.zf + ring;
    const ldpc_const_i32 hdr = as_constant(a.hdr), pack = as_constant(a.pack), cslot = as_constant(a.col_slot);
    /* OR over the workgroup through one LDS word (no static LDS: P sits at LDS address 0 and the
     * table's byte offsets are final addresses) */
    auto wg_any = [&](bool pred) {
        if (r == 0) *wg_flag = 0u;
        lds_barrier();
        if (__ballot(pred) != 0ull && (r & 63) == 0) *wg_flag = 1u;
        lds_barrier();
        const uint32_t f = *wg_flag;
        lds_barrier();                                             /* before the word is cleared again */
        return f != 0u;
    };
    for (int64_t frame = blockIdx.x; frame < a.frames; frame += gridDim.x) {
        const float *y = a.llr + (size_t)frame * a.N;
        if (row) {
            for (int bc = 0; bc < a.nb; ++bc) {
                const int slot = cslot[bc];
                if (slot >= 0) P[slot * z + r] = y[bc * z + r];
            }
            /* iteration 0: R = 0 (|ab| = |ac| = 0, signs +); an external column starts from its
             * channel value.  Written to the ring so that every layer step finds its record there. */
            for (int l = 0; l < a.layers; ++l) {
                uint4 rec = uint4{0u, 0u, 0u, 0u};
                if (hdr[l * 4 + 1]) rec.w = __float_as_uint(y[hdr[l * 4 + 2] + ldsp_wrap(r, hdr[l * 4 + 3], z)]);
                recs[(size_t)l * z] = rec;
            }
        }
        uint4 cur = uint4{0u, 0u, 0u, 0u};
        if (row) cur = recs[0];
        __syncthreads();
        int time = 0;
        bool clean = false;
        /* lane mask of the wave's rows of layer l whose hard decisions have odd parity */
        auto layer_odd = [&](const int l) {
            const int dl = hdr[l * 4], ext = hdr[l * 4 + 1];
            const ldpc_const_i32 pk = pack + (size_t)l * kLdspPackStride;
            uint64_t par = 0;
            switch (dl) {
#define LDPC_LDSP_CASE(D) case D + 1: par = ldsp_row_parity<D + 1>(P, pk, z, r); break;
                LDPC_LDSP_WIDTHS(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
            default: break;
            }
            /* hard decision of the layer's external column: its posterior is in my record */
            if (ext) par ^= __ballot(__uint_as_float(recs[(size_t)l * z].w) < 0.0f);
            return par;
        };
        while (true) {
            for (int l = 0; l < a.layers; ++l) {
                /* the next layer step's record (wrapping into the next iteration), requested before
                 * this step's work; with a single layer it is this step's own output */
                const int ln = l + 1 < a.layers ? l + 1 : 0;
                /* by every lane, outside any branch: a conditional request makes the compiler copy the
                 * registers, and wait for them, where the branch ends -- at once.  Lanes beyond the last row
                 * read their neighbours' records (the rings end with spare ones). */
                uint4 nxt = recs[(size_t)ln * z];
                const int dl = hdr[l * 4], ext = hdr[l * 4 + 1];
                const ldpc_const_i32 pk = pack + (size_t)l * kLdspPackStride;
                uint4 rec = uint4{0u, 0u, 0u, 0u};
                if (row) {
                    uint32_t par = 0;                               /* (ldsp_row_any's; not used here) */
                    bool done = false;
                    if (ext) {
                        switch (dl) {
#define LDPC_LDSP_CASE(D) case D: done = ldsp_row<D, 1, true, false>(P, pk, z, r, cur, &rec, &par); break;
                            LDPC_LDSP_WIDTHS(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                    } else {
                        switch (dl) {
#define LDPC_LDSP_CASE(D) case D + 1: done = ldsp_row<D + 1, 0, true, false>(P, pk, z, r, cur, &rec, &par); break;
                            LDPC_LDSP_WIDTHS(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                    }
                    if (!done) rec = ldsp_row_any(P, pk, dl, ext, z, r, cur, zfs + (size_t)l * z, &par);
                }
                /* the requested record has had this step's work to arrive: take it -- on every path, not
                 * inside the branch above -- BEFORE the store below is issued, or the wait for it would
                 * cover the store as well and put a full memory round trip into every layer step */
                asm volatile("" : "+v"(nxt.x), "+v"(nxt.y), "+v"(nxt.z), "+v"(nxt.w) : : "memory");
                if (row) recs[(size_t)l * z] = rec;
                if (a.layers == 1) nxt = rec;
                lds_barrier();
                cur = nxt;
            }
            /* syndrome of the hard decisions: every round when a clean frame stops early, else only
             * after the last one (its only use then is the frame's converged flag) */
            ++time;
            int any_bad = 1;
            if (a.early_term || time == a.rounds) {
                /* the rows of the last layer first (their columns are in the iteration's final state like all
                 * others, but a frame that has not converged nearly always shows it there already): only
                 * when all of them are even the other layers are looked at.  The row code does not
                 * keep parities: one layer's worth of reads here is cheaper than an instruction per edge. */
                uint64_t bad = row ? layer_odd(a.layers - 1) : 0ull;
                if (!wg_any(bad != 0ull)) {
                    if (row)
                        for (int l = 0; l + 1 < a.layers; ++l) bad |= layer_odd(l);
                    any_bad = wg_any(bad != 0ull) ? 1 : 0;
                }
            }
            clean = !any_bad;
            if ((clean && a.early_term) || time == a.rounds) break;
        }
        /* toChar (decodeCL.c:414-423): the information columns sit in LDS at slot = block column */
        const int64_t base = frame * (int64_t)a.K / 8;
        for (int j = r; j < a.K / 8; j += LANES) {
            unsigned byte = 0;
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) byte |= (P[j * 8 + bit] < 0.0f ? 1u : 0u) << bit;
            if (base + j < a.out_bytes) a.out[base + j] = (uint8_t)byte;
        }
        if (a.dump_p && row) {
            for (int bc = 0; bc < a.nb; ++bc) {
                const int slot = cslot[bc];
                if (slot >= 0) a.dump_p[(size_t)frame * a.N + bc * z + r] = P[slot * z + r];
            }
            for (int l = 0; l < a.layers; ++l)
                if (hdr[l * 4 + 1])
                    a.dump_p[(size_t)frame * a.N + hdr[l * 4 + 2] + ldsp_wrap(r, hdr[l * 4 + 3], z)] =
                        __uint_as_float(recs[(size_t)l * z].w);
        }
        if (a.dump_r && row) {
            for (int l = 0; l < a.layers; ++l) {
                const int d = hdr[l * 4] + hdr[l * 4 + 1], e0 = a.layer_e0[l];
                const uint4 rec = recs[(size_t)l * z];
                const uint32_t zf = (rec.z & kLdspIrregular) ? zfs[(size_t)l * z] : 0u;
                for (int k = 0; k < d; ++k)
                    a.dump_r[(size_t)frame * a.E + e0 + r * d + k] = __uint_as_float(ldsp_old_message(rec, zf, k, d));
            }
        }
        if (r == 0) {
            const int it = clean ? time : a.max_iter;
            if (a.iters) a.iters[frame] = it;
            atomicMax(&a.summary[0], it);
            if (clean) atomicAdd(&a.summary[1], 1);
        }
        __syncthreads();                                           /* P is refilled for the next frame */
    }
}


/* Circulants of <= 32 rows (the reference's own Test.cpp code: z = 24): G = 64 / z frames share one
 * wave -- lanes [g z, (g + 1) z) are the rows of frame g -- each with its own posteriors in LDS and
 * its own record ring.  One wave per workgroup, so "barriers" only order the wave's own LDS
 * traffic; a frame whose syndrome is clean goes idle until the wave's last frame is done. */
template <int kUnused = 0>       /* a template only so that several translation units may include this header */
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LDPC_LDSP_WAVES_PER_EU)))
void layered_ldsp_packed_kernel(const LdspArgs a, const int G)
{
    extern __shared__ float lds[];
    const int lane = (int)threadIdx.x;
    const int z = a.z;
    const int g = lane / z, r = lane - g * z;
    const bool member = g < G;                                      /* lane belongs to a frame slot */
    const size_t frame_floats = ((size_t)a.lds_cols * z + 1) & ~(size_t)1;
    float *P = lds + (size_t)(member ? g : 0) * frame_floats;       /* [lds_cols][z] of my frame */
    uint64_t *extneg = reinterpret_cast<uint64_t *>(lds + (size_t)G * frame_floats);   /* [layers] lane masks */
    const size_t ring = ((size_t)blockIdx.x * G + (member ? g : 0)) * ((size_t)a.layers * z) + r;
    uint4 *recs = a.recs + ring;
    uint32_t *zfs = a.zf + ring;
    const ldpc_const_i32 hdr = as_constant(a.hdr), pack = as_constant(a.pack), cslot = as_constant(a.col_slot);
    const uint64_t gmask = (z >= 64 ? ~0ull : ((1ull << z) - 1ull)) << (member ? g * z : 0);
    for (int64_t frame0 = (int64_t)blockIdx.x * G; frame0 < a.frames; frame0 += (int64_t)gridDim.x * G) {
        const int64_t frame = frame0 + g;
        const bool mine = member && frame < a.frames;
        const float *y = a.llr + (size_t)(mine ? frame : 0) * a.N;
        if (mine) {
            for (int bc = 0; bc < a.nb; ++bc) {
                const int slot = cslot[bc];
                if (slot >= 0) P[slot * z + r] = y[bc * z + r];
            }
            for (int l = 0; l < a.layers; ++l) {
                uint4 rec = uint4{0u, 0u, 0u, 0u};
                if (hdr[l * 4 + 1]) rec.w = __float_as_uint(y[hdr[l * 4 + 2] + ldsp_wrap(r, hdr[l * 4 + 3], z)]);
                recs[(size_t)l * z] = rec;
            }
        }
        uint4 cur = uint4{0u, 0u, 0u, 0u};
        if (mine) cur = recs[0];
        lds_barrier();
        int time = 0, my_iters = a.max_iter;
        bool active = mine, clean = false;
        while (__ballot(active) != 0ull) {
            uint32_t last_bad = 0;
            for (int l = 0; l < a.layers; ++l) {
                const int ln = l + 1 < a.layers ? l + 1 : 0;
                uint4 nxt = uint4{0u, 0u, 0u, 0u};
                if (active && a.layers > 1) nxt = recs[(size_t)ln * z];
                const int dl = hdr[l * 4], ext = hdr[l * 4 + 1];
                const ldpc_const_i32 pk = pack + (size_t)l * kLdspPackStride;
                if (active) {
                    uint4 rec;
                    uint32_t par = 0;
                    bool done = false;
                    if (ext) {
                        switch (dl) {
#define LDPC_LDSP_CASE(D) case D: done = ldsp_row<D, 1>(P, pk, z, r, cur, &rec, &par); break;
                            LDPC_LDSP_WIDTHS(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                    } else {
                        switch (dl) {
#define LDPC_LDSP_CASE(D) case D + 1: done = ldsp_row<D + 1, 0>(P, pk, z, r, cur, &rec, &par); break;
                            LDPC_LDSP_WIDTHS(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                    }
                    if (!done) rec = ldsp_row_any(P, pk, dl, ext, z, r, cur, zfs + (size_t)l * z, &par);
                    last_bad = par;
                    asm volatile("" : "+v"(nxt.x), "+v"(nxt.y), "+v"(nxt.z), "+v"(nxt.w) : : "memory");
                    recs[(size_t)l * z] = rec;
                    if (a.layers == 1) nxt = rec;
                    if (ext) {
                        const uint64_t neg = __ballot(__uint_as_float(rec.w) < 0.0f);
                        const uint64_t act = __ballot(true);
                        if (lane == (int)__builtin_ctzll(act)) extneg[l] = neg;
                    }
                }
                lds_barrier();
                cur = nxt;
            }
            ++time;
            const bool check = a.early_term || time == a.rounds;
            const uint64_t last_mask = __ballot(active && last_bad);
            bool any_bad = true;
            if (check && __ballot(active && (last_mask & gmask) == 0ull) != 0ull) {
                /* some frame's last layer is all even: the full syndrome, for the frames that need it */
                uint64_t bad = 0;
                const bool need = active && (last_mask & gmask) == 0ull;
                if (need) {
                    for (int l = 0; l < a.layers; ++l) {
                        const int dl = hdr[l * 4], ext = hdr[l * 4 + 1];
                        const ldpc_const_i32 pk = pack + (size_t)l * kLdspPackStride;
                        uint64_t par = 0;
                        switch (dl) {
#define LDPC_LDSP_CASE(D) case D + 1: par = ldsp_row_parity<D + 1>(P, pk, z, r); break;
                            LDPC_LDSP_WIDTHS(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                        if (ext) par ^= extneg[l];
                        bad |= par;
                    }
                    any_bad = (bad & gmask) != 0ull;
                }
            }
            if (active) {
                clean = check && !any_bad;
                if ((clean && a.early_term) || time == a.rounds) {
                    active = false;
                    my_iters = clean ? time : a.max_iter;
                }
            }
            lds_barrier();
        }
        if (mine) {
            const int64_t base = frame * (int64_t)a.K / 8;
            for (int j = r; j < a.K / 8; j += z) {
                unsigned byte = 0;
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) byte |= (P[j * 8 + bit] < 0.0f ? 1u : 0u) << bit;
                if (base + j < a.out_bytes) a.out[base + j] = (uint8_t)byte;
            }
            if (a.dump_p) {
                for (int bc = 0; bc < a.nb; ++bc) {
                    const int slot = cslot[bc];
                    if (slot >= 0) a.dump_p[(size_t)frame * a.N + bc * z + r] = P[slot * z + r];
                }
                for (int l = 0; l < a.layers; ++l)
                    if (hdr[l * 4 + 1])
                        a.dump_p[(size_t)frame * a.N + hdr[l * 4 + 2] + ldsp_wrap(r, hdr[l * 4 + 3], z)] =
                            __uint_as_float(recs[(size_t)l * z].w);
            }
            if (a.dump_r) {
                for (int l = 0; l < a.layers; ++l) {
                    const int d = hdr[l * 4] + hdr[l * 4 + 1], e0 = a.layer_e0[l];
                    const uint4 rec = recs[(size_t)l * z];
                    const uint32_t zf = (rec.z & kLdspIrregular) ? zfs[(size_t)l * z] : 0u;
                    for (int k = 0; k < d; ++k)
                        a.dump_r[(size_t)frame * a.E + e0 + r * d + k] = __uint_as_float(ldsp_old_message(rec, zf, k, d));
                }
            }
            if (r == 0) {
                if (a.iters) a.iters[frame] = my_iters;
                atomicMax(&a.summary[0], my_iters);
                if (clean) atomicAdd(&a.summary[1], 1);
            }
        }
        lds_barrier();                                             /* P is refilled for the next frames */
    }
}


/* ------------------------------------------------------------------------------------------
 * Flooding min-sum (DecodeMS / DecodeCPU: the arithmetic of refreshRMS, refreshPostPMS,
 * refreshQMS, decodeCL.c:113-186) with the same placement: posteriors in LDS, one 16-byte record
 * per check row {min1, min2, signs | argmin}.  Every R_k = +-min is exactly (magnitude, sign bit),
 * so there is no irregular case here.  Per iteration every row reads the OLD posteriors
 * (q_k = P_old[col] - R_old,k is refreshQMS applied on the fly), and adds its new messages to the
 * NEW posteriors, which start from the channel values: two LDS images, layers in ascending order
 * with a barrier in between, so that every column receives y + R_1 + R_2 + ... in ascending row
 * order as refreshPostPMS computes it.  Hard decision !(p > 0), syndrome, stop when clean. */
template <int DL, int EXT>
__device__ __forceinline__ void ldsp_flood_row(const float *Pold, float *Pnew, ldpc_const_i32 pk, int z, int r,
                                               const uint4 old, float pext_old, float yext, uint4 *out,
                                               uint64_t *par_mask, uint64_t *ext_mask)
{
    constexpr int D = DL + EXT;
    constexpr int DLA = DL > 0 ? DL : 1;
    float q[D];
    uint32_t off[DLA];
#pragma unroll
    for (int k = 0; k < DL; ++k) {
        const uint32_t t = (uint32_t)(r * 4) + (uint32_t)pk[kLdspMaxDeg + k];
        const uint32_t tw = t - (uint32_t)(z * 4);
        off[k] = (t < tw ? t : tw) + (uint32_t)pk[k];
    }
    const int oidx = (int)((old.z >> 24) & 31u);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const uint32_t sel = (k == oidx) ? old.y : old.x;
        const uint32_t rold = ((old.z << (31 - (D - 1 - k))) & 0x80000000u) | sel;
        const float pin = k < DL ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(Pold) + off[k < DL ? k : 0])
                                 : pext_old;
        q[k] = pin - __uint_as_float(rold);
    }
    float m1 = 1000.0f, m2 = 1000.0f;
    int idx = 31;                                                   /* none */
    uint32_t par = 0, neg[D];
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const float mag = __builtin_fabsf(q[k]);
        neg[k] = (q[k] < 0.0f) ? 0x80000000u : 0u;
        par ^= neg[k];
        const bool lt1 = mag < m1, lt2 = mag < m2;                  /* NaN: neither */
        m2 = lt1 ? m1 : (lt2 ? mag : m2);
        m1 = lt1 ? mag : m1;
        idx = lt1 ? k : idx;
    }
    const uint32_t b1 = __float_as_uint(m1), b2 = __float_as_uint(m2);
    uint32_t signs = 0, pext = 0;
    uint64_t pm = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const uint32_t rn = ((k == idx) ? b2 : b1) ^ (par ^ neg[k]);    /* s ? -b : b, b >= 0 */
        float pn;
        if (k < DL) {
            float *p = reinterpret_cast<float *>(reinterpret_cast<char *>(Pnew) + off[k < DL ? k : 0]);
            pn = *p + __uint_as_float(rn);
            *p = pn;
        } else {
            pn = yext + __uint_as_float(rn);                        /* a single-layer column: y + its one message */
            pext = __float_as_uint(pn);
            *ext_mask = __ballot(!(pn > 0.0f));
        }
        signs = __builtin_amdgcn_alignbit(signs, rn, 31);
        pm ^= __ballot(!(pn > 0.0f));
    }
    *out = uint4{b1, b2, signs | ((uint32_t)idx << 24), pext};
    *par_mask = pm;
}

/* The same placement with the arithmetic of the reference's fused flooding kernel decodeOnceMS
 * (DecodeMSCL, decodeCL.c:482-512): product sign and the 1000 / 1001 two-minimum rule with <=, i.e.
 * the check rule of the layered kernel above (same record format, same irregular slow path), hard
 * decision P < 0. */
__device__ __forceinline__ uint4 ldsp_mscl_row_any(const float *Pold, float *Pnew, ldpc_const_i32 pk, int dl, int ext,
                                                   int z, int r, const uint4 old, float pext_old, float yext,
                                                   uint32_t *zfp, uint64_t *par_mask, uint64_t *ext_mask)
{
    const int d = dl + ext;
    const uint32_t ozf = (old.z & kLdspIrregular) ? *zfp : 0u;
    float prod = 1.0f, b = 1000.0f, c = 1001.0f;
    int bind = 31;
    for (int k = 0; k < d; ++k) {
        const float rold = __uint_as_float(ldsp_old_message(old, ozf, k, d));
        const float pin = k < dl ? *ldsp_at(const_cast<float *>(Pold), pk[k], pk[kLdspMaxDeg + k], r * 4, z * 4) : pext_old;
        const float q = pin - rold;
        prod *= q;
        const float mag = __builtin_fabsf(q);
        if (mag <= b) { c = b; b = mag; bind = k; }
        else if (mag > b && mag <= c) { c = mag; }
    }
    const float sa = cl_sign(prod);
    const float ab = sa * b, ac = sa * c;
    const uint32_t mab = __float_as_uint(ab) & 0x7fffffffu, mac = __float_as_uint(ac) & 0x7fffffffu;
    uint32_t signs = 0, zf = 0, pext = 0;
    uint64_t pm = 0;
    for (int k = 0; k < d; ++k) {
        const float rold = __uint_as_float(ldsp_old_message(old, ozf, k, d));   /* P_old is untouched: same q again */
        float *pn_at = k < dl ? ldsp_at(Pnew, pk[k], pk[kLdspMaxDeg + k], r * 4, z * 4) : nullptr;
        const float pin = k < dl ? *ldsp_at(const_cast<float *>(Pold), pk[k], pk[kLdspMaxDeg + k], r * 4, z * 4) : pext_old;
        const float q = pin - rold;
        const float rn = cl_sign(q) * ((k == bind) ? ac : ab);
        const float pn = (k < dl ? *pn_at : yext) + rn;
        if (k < dl) *pn_at = pn;
        else { pext = __float_as_uint(pn); *ext_mask = __ballot(pn < 0.0f); }
        const uint32_t rb = __float_as_uint(rn);
        signs = (signs << 1) | (rb >> 31);
        if ((rb & 0x7fffffffu) != ((k == bind) ? mac : mab)) zf |= 1u << k;
        pm ^= __ballot(pn < 0.0f);
    }
    uint32_t word = signs | ((uint32_t)bind << 24);
    if (zf) {
        word |= kLdspIrregular;
        *zfp = zf;
    }
    *par_mask = pm;
    return uint4{mab, mac, word, pext};
}

template <int DL, int EXT>
__device__ __forceinline__ bool ldsp_mscl_row(const float *Pold, float *Pnew, ldpc_const_i32 pk, int z, int r,
                                              const uint4 old, float pext_old, float yext, uint4 *out,
                                              uint64_t *par_mask, uint64_t *ext_mask)
{
    constexpr int D = DL + EXT;
    constexpr int DLA = DL > 0 ? DL : 1;
    if (__ballot((old.z & kLdspIrregular) != 0u) != 0ull) return false;
    float q[D];
    uint32_t off[DLA];
#pragma unroll
    for (int k = 0; k < DL; ++k) {
        const uint32_t t = (uint32_t)(r * 4) + (uint32_t)pk[kLdspMaxDeg + k];
        const uint32_t tw = t - (uint32_t)(z * 4);
        off[k] = (t < tw ? t : tw) + (uint32_t)pk[k];
    }
    const int obind = (int)((old.z >> 24) & 31u);
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const uint32_t sel = (k == obind) ? old.y : old.x;
        const uint32_t rold = ((old.z << (31 - (D - 1 - k))) & 0x80000000u) | sel;
        const float pin = k < DL ? *reinterpret_cast<const float *>(reinterpret_cast<const char *>(Pold) + off[k < DL ? k : 0])
                                 : pext_old;
        q[k] = pin - __uint_as_float(rold);
    }
    float prod = 1.0f, b = 1000.0f, c = 1001.0f;
    int bind = 31;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        prod *= q[k];
        const float mag = __builtin_fabsf(q[k]);
        const bool le = mag <= b;
        bind = le ? k : bind;
        c = __builtin_amdgcn_fmed3f(b, mag, c);
        b = __builtin_amdgcn_fmed3f(0.0f, mag, b);
    }
    const uint32_t pb = __float_as_uint(prod);
    if (__ballot(!ldsp_regular(pb)) != 0ull) return false;         /* nothing written yet */
    const uint32_t ps = pb & 0x80000000u;
    const uint32_t mb = __float_as_uint(b), mc = __float_as_uint(c);
    uint32_t signs = 0, pext = 0;
    uint64_t pm = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) {
        const uint32_t sel = (k == bind) ? mc : mb;
        const uint32_t rn = ((__float_as_uint(q[k]) ^ ps) & 0x80000000u) | sel;
        float pn;
        if (k < DL) {
            float *p = reinterpret_cast<float *>(reinterpret_cast<char *>(Pnew) + off[k < DL ? k : 0]);
            pn = *p + __uint_as_float(rn);
            *p = pn;
        } else {
            pn = yext + __uint_as_float(rn);
            pext = __float_as_uint(pn);
            *ext_mask = __ballot(pn < 0.0f);
        }
        signs = __builtin_amdgcn_alignbit(signs, rn, 31);
        pm ^= __ballot(pn < 0.0f);
    }
    *out = uint4{mb, mc, signs | ((uint32_t)bind << 24), pext};
    *par_mask = pm;
    return true;
}

/* hard decision: !(p > 0) in the MS chain (decodeCL.c:161-165), p < 0 in the fused reference kernels (:541) */
template <bool CHAIN> __device__ __forceinline__ bool ldsp_flood_bit(float p) { return CHAIN ? !(p > 0.0f) : (p < 0.0f); }

template <int D, bool CHAIN>
__device__ __forceinline__ uint64_t ldsp_flood_parity(const float *P, ldpc_const_i32 pk, int z, int r)
{
    float v[D];
#pragma unroll
    for (int k = 0; k < D; ++k) v[k] = *ldsp_at(const_cast<float *>(P), pk[k], pk[kLdspMaxDeg + k], r * 4, z * 4);
    uint64_t par = 0;
#pragma unroll
    for (int k = 0; k < D; ++k) par ^= __ballot(ldsp_flood_bit<CHAIN>(v[k]));
    return par;
}

#define LDPC_LDSP_WIDTHS1(X) X(1) X(2) X(3) X(4) X(5) X(6) X(7) X(8) X(9) X(10) X(11) X(12) X(13) X(14) X(15) X(16) \
    X(17) X(18) X(19) X(20) X(21) X(22) X(23) X(24)

template <int MAXW, bool CHAIN>
__global__ __launch_bounds__(64 * MAXW) __attribute__((amdgpu_waves_per_eu(LDPC_LDSP_WAVES_PER_EU)))
void flood_ldsp_kernel(const LdspArgs a)
{
    extern __shared__ float lds[];
    const int r = (int)threadIdx.x, LANES = (int)blockDim.x, MW = LANES >> 6, wave = r >> 6;
    const int z = a.z;
    const size_t image = ((size_t)a.lds_cols * z + 1) & ~(size_t)1;
    float *Pa = lds, *Pb = lds + image;                             /* old / new posteriors, [lds_cols][z] each */
    uint64_t *extneg = reinterpret_cast<uint64_t *>(lds + 2 * image);   /* [layers][MW] */
    uint32_t *wg_flag = reinterpret_cast<uint32_t *>(extneg + (size_t)a.layers * MW);
    const bool row = r < z;
    uint4 *recs = a.recs + (size_t)blockIdx.x * ((size_t)a.layers * z) + r;
    uint32_t *zfs = a.zf + (size_t)blockIdx.x * ((size_t)a.layers * z) + r;
    const ldpc_const_i32 hdr = as_constant(a.hdr), pack = as_constant(a.pack), cslot = as_constant(a.col_slot);
    auto wg_any = [&](bool pred) {
        if (r == 0) *wg_flag = 0u;
        lds_barrier();
        if (__ballot(pred) != 0ull && (r & 63) == 0) *wg_flag = 1u;
        lds_barrier();
        const uint32_t f = *wg_flag;
        lds_barrier();
        return f != 0u;
    };
    auto fill = [&](float *P, const float *y) {                    /* the LDS-resident columns' channel values */
        if (row)
            for (int bc = 0; bc < a.nb; ++bc) {
                const int slot = cslot[bc];
                if (slot >= 0) P[slot * z + r] = y[bc * z + r];
            }
    };
    for (int64_t frame = blockIdx.x; frame < a.frames; frame += gridDim.x) {
        const float *y = a.llr + (size_t)frame * a.N;
        fill(Pa, y);                                               /* Q_0 = y: P_0 = y, R_0 = 0 */
        int time = 0;
        bool clean = false;
        uint4 cur = uint4{0u, 0u, 0u, 0u};
        while (true) {
            fill(Pb, y);                                           /* refreshPostPMS starts from the channel value */
            __syncthreads();
            uint64_t last_bad = 0;
            for (int l = 0; l < a.layers; ++l) {
                const int ln = l + 1 < a.layers ? l + 1 : 0;
                uint4 nxt = uint4{0u, 0u, 0u, 0u};
                if (row && a.layers > 1 && (time > 0 || ln == 0)) nxt = recs[(size_t)ln * z];
                const int dl = hdr[l * 4], ext = hdr[l * 4 + 1];
                const ldpc_const_i32 pk = pack + (size_t)l * kLdspPackStride;
                if (row) {
                    float yext = 0.0f;
                    if (ext) yext = y[hdr[l * 4 + 2] + ldsp_wrap(r, hdr[l * 4 + 3], z)];
                    const float pext_old = time == 0 ? yext : __uint_as_float(cur.w);
                    uint4 rec = cur;
                    uint64_t pm = 0, em = 0;
                    bool done = CHAIN;                             /* the chain arithmetic has no slow path */
                    if (ext) {
                        switch (dl) {
#define LDPC_LDSP_CASE(D) case D:                                                                                  \
                            if (CHAIN) ldsp_flood_row<D, 1>(Pa, Pb, pk, z, r, cur, pext_old, yext, &rec, &pm, &em);        \
                            else done = ldsp_mscl_row<D, 1>(Pa, Pb, pk, z, r, cur, pext_old, yext, &rec, &pm, &em);         \
                            break;
                            LDPC_LDSP_WIDTHS(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                    } else {
                        switch (dl) {
#define LDPC_LDSP_CASE(D) case D + 1:                                                                              \
                            if (CHAIN) ldsp_flood_row<D + 1, 0>(Pa, Pb, pk, z, r, cur, 0.0f, 0.0f, &rec, &pm, &em);        \
                            else done = ldsp_mscl_row<D + 1, 0>(Pa, Pb, pk, z, r, cur, 0.0f, 0.0f, &rec, &pm, &em);         \
                            break;
                            LDPC_LDSP_WIDTHS(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                    }
                    if (!CHAIN && !done)
                        rec = ldsp_mscl_row_any(Pa, Pb, pk, dl, ext, z, r, cur, pext_old, yext, zfs + (size_t)l * z, &pm, &em);
                    if (ext && (r & 63) == 0) extneg[l * MW + wave] = em;
                    last_bad = pm;
                    asm volatile("" : "+v"(nxt.x), "+v"(nxt.y), "+v"(nxt.z), "+v"(nxt.w) : : "memory");
                    recs[(size_t)l * z] = rec;
                    if (a.layers == 1) nxt = rec;
                }
                lds_barrier();
                cur = nxt;
            }
            ++time;
            int any_bad = 1;
            /* the last layer's rows have just written the final posteriors of their columns */
            if ((a.early_term || time == a.rounds) && !wg_any(row && last_bad != 0ull)) {
                uint64_t bad = 0;
                if (row) {
                    for (int l = 0; l < a.layers; ++l) {
                        const ldpc_const_i32 pk = pack + (size_t)l * kLdspPackStride;
                        uint64_t par = 0;
                        switch (hdr[l * 4]) {
#define LDPC_LDSP_CASE(D) case D: par = ldsp_flood_parity<D, CHAIN>(Pb, pk, z, r); break;
                            LDPC_LDSP_WIDTHS1(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                        if (hdr[l * 4 + 1]) par ^= extneg[l * MW + wave];
                        bad |= par;
                    }
                }
                any_bad = wg_any(bad != 0ull) ? 1 : 0;
            }
            clean = !any_bad;
            float *t = Pa; Pa = Pb; Pb = t;                         /* the new posteriors are the next round's old ones */
            if ((clean && a.early_term) || time == a.rounds) break;
        }
        /* Pa holds the final posteriors; the information columns sit at slot = block column */
        const int64_t base = frame * (int64_t)a.K / 8;
        for (int j = r; j < a.K / 8; j += LANES) {
            unsigned byte = 0;
#pragma unroll
            for (int bit = 0; bit < 8; ++bit) byte |= (ldsp_flood_bit<CHAIN>(Pa[j * 8 + bit]) ? 1u : 0u) << bit;
            if (base + j < a.out_bytes) a.out[base + j] = (uint8_t)byte;
        }
        if (a.dump_p && row) {
            for (int bc = 0; bc < a.nb; ++bc) {
                const int slot = cslot[bc];
                if (slot >= 0) a.dump_p[(size_t)frame * a.N + bc * z + r] = Pa[slot * z + r];
            }
            for (int l = 0; l < a.layers; ++l)
                if (hdr[l * 4 + 1])
                    a.dump_p[(size_t)frame * a.N + hdr[l * 4 + 2] + ldsp_wrap(r, hdr[l * 4 + 3], z)] =
                        __uint_as_float(recs[(size_t)l * z].w);
        }
        if (a.dump_r && row) {
            for (int l = 0; l < a.layers; ++l) {
                const int d = hdr[l * 4] + hdr[l * 4 + 1], e0 = a.layer_e0[l];
                const uint4 rec = recs[(size_t)l * z];
                const uint32_t zf = (rec.z & kLdspIrregular) ? zfs[(size_t)l * z] : 0u;
                for (int k = 0; k < d; ++k)
                    a.dump_r[(size_t)frame * a.E + e0 + r * d + k] = __uint_as_float(ldsp_old_message(rec, zf, k, d));
            }
        }
        if (r == 0) {
            const int it = clean ? time : a.max_iter;
            if (a.iters) a.iters[frame] = it;
            atomicMax(&a.summary[0], it);
            if (clean) atomicAdd(&a.summary[1], 1);
        }
        __syncthreads();
    }
}

/* flood_ldsp_kernel for circulants of <= 32 rows: G = 64 / z frames per wave, one wave per
 * workgroup (see layered_ldsp_packed_kernel). */
template <bool CHAIN>
__global__ __launch_bounds__(64) __attribute__((amdgpu_waves_per_eu(LDPC_LDSP_WAVES_PER_EU)))
void flood_ldsp_packed_kernel(const LdspArgs a, const int G)
{
    extern __shared__ float lds[];
    const int lane = (int)threadIdx.x;
    const int z = a.z;
    const int g = lane / z, r = lane - g * z;
    const bool member = g < G;
    const size_t image = ((size_t)a.N + 1) & ~(size_t)1;
    float *Pa = lds + (size_t)(member ? g : 0) * 2 * image, *Pb = Pa + image;
    uint4 *recs = a.recs + ((size_t)blockIdx.x * G + (member ? g : 0)) * ((size_t)a.layers * z) + r;
    uint32_t *zfs = a.zf + ((size_t)blockIdx.x * G + (member ? g : 0)) * ((size_t)a.layers * z) + r;
    const ldpc_const_i32 hdr = as_constant(a.hdr), pack = as_constant(a.pack);
    const uint64_t gmask = (z >= 64 ? ~0ull : ((1ull << z) - 1ull)) << (member ? g * z : 0);
    for (int64_t frame0 = (int64_t)blockIdx.x * G; frame0 < a.frames; frame0 += (int64_t)gridDim.x * G) {
        const int64_t frame = frame0 + g;
        const bool mine = member && frame < a.frames;
        const float *y = a.llr + (size_t)(mine ? frame : 0) * a.N;
        if (mine)
            for (int n = r; n < a.N; n += z) Pa[n] = y[n];
        int time = 0, my_iters = a.max_iter;
        bool active = mine, clean = false;
        uint4 cur = uint4{0u, 0u, 0u, 0u};
        while (__ballot(active) != 0ull) {
            if (active)
                for (int n = r; n < a.N; n += z) Pb[n] = y[n];
            lds_barrier();
            uint64_t last_bad = 0;
            for (int l = 0; l < a.layers; ++l) {
                const int ln = l + 1 < a.layers ? l + 1 : 0;
                uint4 nxt = uint4{0u, 0u, 0u, 0u};
                if (active && a.layers > 1 && (time > 0 || ln == 0)) nxt = recs[(size_t)ln * z];
                const int d = hdr[l * 4];
                const ldpc_const_i32 pk = pack + (size_t)l * kLdspPackStride;
                if (active) {
                    uint4 rec = cur;
                    uint64_t pm = 0, em = 0;
                    bool done = CHAIN;
                    switch (d) {
#define LDPC_LDSP_CASE(D) case D:                                                                                  \
                        if (CHAIN) ldsp_flood_row<D, 0>(Pa, Pb, pk, z, r, cur, 0.0f, 0.0f, &rec, &pm, &em);                \
                        else done = ldsp_mscl_row<D, 0>(Pa, Pb, pk, z, r, cur, 0.0f, 0.0f, &rec, &pm, &em);                 \
                        break;
                        LDPC_LDSP_WIDTHS1(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                    default: break;
                    }
                    if (!CHAIN && !done)
                        rec = ldsp_mscl_row_any(Pa, Pb, pk, d, 0, z, r, cur, 0.0f, 0.0f, zfs + (size_t)l * z, &pm, &em);
                    last_bad = pm;
                    asm volatile("" : "+v"(nxt.x), "+v"(nxt.y), "+v"(nxt.z), "+v"(nxt.w) : : "memory");
                    recs[(size_t)l * z] = rec;
                    if (a.layers == 1) nxt = rec;
                }
                lds_barrier();
                cur = nxt;
            }
            ++time;
            const bool check = a.early_term || time == a.rounds;
            bool any_bad = true;
            const bool need = active && check && (last_bad & gmask) == 0ull;
            if (__ballot(need) != 0ull) {
                uint64_t bad = 0;
                if (need) {
                    for (int l = 0; l < a.layers; ++l) {
                        const ldpc_const_i32 pk = pack + (size_t)l * kLdspPackStride;
                        switch (hdr[l * 4]) {
#define LDPC_LDSP_CASE(D) case D: bad |= ldsp_flood_parity<D, CHAIN>(Pb, pk, z, r); break;
                            LDPC_LDSP_WIDTHS1(LDPC_LDSP_CASE)
#undef LDPC_LDSP_CASE
                        default: break;
                        }
                    }
                    any_bad = (bad & gmask) != 0ull;
                }
            }
            if (active) {
                clean = check && !any_bad;
                float *t = Pa; Pa = Pb; Pb = t;
                if ((clean && a.early_term) || time == a.rounds) {
                    active = false;
                    my_iters = clean ? time : a.max_iter;
                }
            }
            lds_barrier();
        }
        if (mine) {
            const int64_t base = frame * (int64_t)a.K / 8;
            for (int j = r; j < a.K / 8; j += z) {
                unsigned byte = 0;
#pragma unroll
                for (int bit = 0; bit < 8; ++bit) byte |= (ldsp_flood_bit<CHAIN>(Pa[j * 8 + bit]) ? 1u : 0u) << bit;
                if (base + j < a.out_bytes) a.out[base + j] = (uint8_t)byte;
            }
            if (a.dump_p)
                for (int n = r; n < a.N; n += z) a.dump_p[(size_t)frame * a.N + n] = Pa[n];
            if (a.dump_r) {
                for (int l = 0; l < a.layers; ++l) {
                    const int d = hdr[l * 4], e0 = a.layer_e0[l];
                    const uint4 rec = recs[(size_t)l * z];
                    const uint32_t zf = (rec.z & kLdspIrregular) ? zfs[(size_t)l * z] : 0u;
                    for (int k = 0; k < d; ++k)
                        a.dump_r[(size_t)frame * a.E + e0 + r * d + k] = __uint_as_float(ldsp_old_message(rec, zf, k, d));
                }
            }
            if (r == 0) {
                if (a.iters) a.iters[frame] = my_iters;
                atomicMax(&a.summary[0], my_iters);
                if (clean) atomicAdd(&a.summary[1], 1);
            }
        }
        lds_barrier();
    }
}

/* ---------------------------------------------------------------- host side */

struct LdspPlan {
    bool eligible = false;
    int32_t z = 0, layers = 0, N = 0, E = 0, M = 0, nb = 0, lds_cols = 0, ext_cols = 0;
    int32_t *hdr = nullptr, *pack = nullptr, *col_slot = nullptr, *layer_e0 = nullptr;   /* device */
    uint4 *recs = nullptr;
    uint32_t *zf = nullptr;
    float *dump_p = nullptr, *dump_r = nullptr;
    int64_t dump_frames = 0;
    int flood = 0;                      /* 0 layered kernels; flood_ldsp_kernel with 1: the MS chain's arithmetic (DecodeMS /
                                           DecodeCPU), 2: the fused reference kernel's (DecodeMSCL) */
    int32_t grid = 0, block = 0, maxw = 0, per_cu = 0, wg_frames = 1;   /* wg_frames: frames per one-wave workgroup (z <= 32) */
    size_t lds_bytes = 0;
};

inline void ldsp_plan_destroy(LdspPlan *pl)
{
    for (void *p : {(void *)pl->hdr, (void *)pl->pack, (void *)pl->col_slot, (void *)pl->layer_e0, (void *)pl->recs,
                    (void *)pl->zf, (void *)pl->dump_p, (void *)pl->dump_r})
        if (p) (void)hipFree(p);
    *pl = LdspPlan();
}

/* the plan builder and the launcher reference every kernel above: compiled by engine_ldsp.hip only
 * (LDPC_ENGINE_LDSP); the host driver calls engine_ldsp_plan_create / engine_ldsp_run */
#ifdef LDPC_ENGINE_LDSP
typedef void (*LdspKernel)(const LdspArgs);
inline LdspKernel ldsp_kernel_for(int maxw) { return maxw <= 8 ? layered_ldsp_kernel<8> : layered_ldsp_kernel<16>; }

/* Detect the QC structure, decide which block columns travel with a record (met by one layer
 * only, last entry of that layer's rows, not an information column), lay the others out in LDS,
 * size the persistent grid and allocate its record rings.  eligible = false (and hipSuccess)
 * when the code does not fit this kernel. */
inline LdspKernel flood_ldsp_kernel_for(int maxw, int kind)
{
    if (kind == 1) return maxw <= 8 ? flood_ldsp_kernel<8, true> : flood_ldsp_kernel<16, true>;
    return maxw <= 8 ? flood_ldsp_kernel<8, false> : flood_ldsp_kernel<16, false>;
}

inline hipError_t ldsp_plan_create(LdspPlan *pl, int32_t M, int32_t N, int64_t E, const std::vector<int32_t> &row_ptr,
                                   const std::vector<int32_t> &cols, int32_t z, int32_t K, int64_t max_batch, int device,
                                   const Tune &tune, int flood = 0)
{
    std::vector<int32_t> lp, bc, sh, e0;
    pl->eligible = false;
    if (z <= 0 || z > 1024 || !fused_detect_qc(M, N, row_ptr, cols, z, lp, bc, sh, e0)) return hipSuccess;
    const int layers = M / z, nb = N / z;
    int max_deg = 0;
    for (int l = 0; l < layers; ++l) max_deg = std::max(max_deg, lp[l + 1] - lp[l]);
    if (max_deg > kLdspMaxDeg) return hipSuccess;
    std::vector<int32_t> deg(nb, 0);
    for (int32_t b : bc) ++deg[b];
    const bool will_pack = z <= 32 && !tune_forced_off(tune.ldsp_pack);
    const bool allow_ext = !tune_forced_off(tune.ldsp_ext) && !(flood && will_pack);   /* the packed flooding kernel keeps every column in LDS */
    std::vector<int32_t> slot(nb, 0), hdr((size_t)layers * 4, 0), pack((size_t)layers * kLdspPackStride, 0);
    std::vector<char> external(nb, 0);
    int ext_cols = 0;
    for (int l = 0; l < layers; ++l) {
        const int last = bc[lp[l + 1] - 1];
        if (allow_ext && deg[last] == 1 && (int64_t)last * z >= K) { external[last] = 1; ++ext_cols; }
    }
    int lds_cols = 0;
    for (int b = 0; b < nb; ++b) slot[b] = external[b] ? -1 : lds_cols++;
    for (int l = 0; l < layers; ++l) {
        const int d = lp[l + 1] - lp[l], last = lp[l + 1] - 1;
        const int ext = external[bc[last]] ? 1 : 0;
        hdr[l * 4 + 0] = d - ext;
        hdr[l * 4 + 1] = ext;
        hdr[l * 4 + 2] = ext ? bc[last] * z : 0;
        hdr[l * 4 + 3] = ext ? sh[last] : 0;
        for (int k = 0; k < d - ext; ++k) {
            const int32_t col4 = slot[bc[lp[l] + k]] * z * 4, shift4 = sh[lp[l] + k] * 4;
            int32_t *row = &pack[(size_t)l * kLdspPackStride];
            row[k] = col4;
            row[kLdspMaxDeg + k] = shift4;
            row[2 * kLdspMaxDeg + k] = col4 + shift4;
            row[3 * kLdspMaxDeg + k] = z * 4 - shift4;
        }
    }
    int mw = (z + 63) / 64;
    if (tune.ldsp_waves) mw = std::min(16, std::max(mw, tune.ldsp_waves));   /* idle waves appended */
    int frames_per_wg = 1;
    if (mw == 1 && will_pack) frames_per_wg = 64 / z;
    const size_t frame_bytes = ((((size_t)lds_cols * z + 1) & ~(size_t)1)) * sizeof(float) * (flood ? 2 : 1);   /* flooding: old and new image */
    while (frames_per_wg > 1 && frames_per_wg * frame_bytes + (size_t)layers * sizeof(uint64_t) + 8 > 60 * 1024) --frames_per_wg;
    /* behind the posteriors: a flag word, and (all kernels but layered_ldsp_kernel, which reads them from its
     * records) the lane masks of the external columns' hard decisions */
    const bool masks = flood || frames_per_wg > 1;
    const size_t lds_bytes = frames_per_wg * frame_bytes + (masks ? (size_t)layers * mw * sizeof(uint64_t) : 0) + 8;
    if (lds_bytes > kLdspMaxLds || lds_cols >= 32768) return hipSuccess;
    pl->wg_frames = frames_per_wg;
    pl->flood = flood;
    pl->z = z; pl->layers = layers; pl->N = N; pl->E = (int32_t)E; pl->M = M; pl->nb = nb;
    pl->lds_cols = lds_cols; pl->ext_cols = ext_cols; pl->lds_bytes = lds_bytes;
    pl->maxw = mw <= 8 ? 8 : 16;
    pl->block = 64 * mw;
    auto up = [](int32_t **dst, const std::vector<int32_t> &v) {
        hipError_t e = hipMalloc((void **)dst, v.size() * sizeof(int32_t));
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    };
    hipError_t e;
    if ((e = up(&pl->hdr, hdr)) || (e = up(&pl->pack, pack)) || (e = up(&pl->col_slot, slot)) || (e = up(&pl->layer_e0, e0)))
        return e;
    const void *k = flood ? (pl->wg_frames > 1 ? (flood == 1 ? (const void *)flood_ldsp_packed_kernel<true> : (const void *)flood_ldsp_packed_kernel<false>)
                                               : (const void *)flood_ldsp_kernel_for(mw <= 8 ? 8 : 16, flood))
                    : pl->wg_frames > 1 ? (const void *)layered_ldsp_packed_kernel<0> : (const void *)ldsp_kernel_for(mw <= 8 ? 8 : 16);
    /* the attribute belongs to the function, not to this plan: always the maximum, so that decoders
     * of different codes can coexist */
    if ((e = hipFuncSetAttribute(k, hipFuncAttributeMaxDynamicSharedMemorySize, (int)kLdspMaxLds))) return e;
    int per_cu = 0, cus = 0;
    if ((e = hipOccupancyMaxActiveBlocksPerMultiprocessor(&per_cu, k, pl->block, lds_bytes))) return e;
    if ((e = hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, device))) return e;
    if (per_cu < 1 || cus < 1) return hipErrorInvalidValue;
    /* The record rings of all resident workgroups are re-read once per iteration: they must stay in the
     * Infinity Cache (256 MiB less what streams through between two uses).  BG1 at Z = 384: 276 KiB per frame;
     * 3 workgroups per CU = 207 MiB: 23.5 ms per batch; 4 per CU (LDS and registers allow it) = 276 MiB: 27.7 ms
     * (profiles/r02_ab_ldsp_resident.txt). */
    const size_t ring_bytes = (size_t)frames_per_wg * M * sizeof(uint4);
    const int cache_cap = (int)std::max<size_t>(1, ((size_t)224 << 20) / ((size_t)cus * ring_bytes));
    if (tune.ldsp_per_cu) per_cu = std::max(1, tune.ldsp_per_cu);   /* a forced value ignores both rules: the grid is persistent */
    else per_cu = std::min(per_cu, cache_cap);
    pl->per_cu = per_cu;
    pl->grid = (int32_t)std::min<int64_t>((std::max<int64_t>(max_batch, 1) + pl->wg_frames - 1) / pl->wg_frames, (int64_t)per_cu * cus);
    if (tune.ldsp_grid) pl->grid = std::max(1, std::min(pl->grid, tune.ldsp_grid));
    /* + 1024 (the largest workgroup): lanes beyond the last row request records too (never used) */
    if ((e = hipMalloc((void **)&pl->recs, ((size_t)pl->grid * pl->wg_frames * M + 1024) * sizeof(uint4)))) return e;
    if ((e = hipMalloc((void **)&pl->zf, ((size_t)pl->grid * pl->wg_frames * M + 1024) * sizeof(uint32_t)))) return e;
    pl->eligible = true;
    return hipSuccess;
}

/* one persistent grid for the whole batch */
inline hipError_t ldsp_run(LdspPlan *pl, const FusedRun &r, hipStream_t s, int32_t *launched)
{
    hipError_t e;
    if ((e = hipMemsetAsync(r.summary, 0, 2 * sizeof(int32_t), s))) return e;
    const int rounds = r.tap_iter ? (r.tap_iter < r.max_iter ? r.tap_iter : r.max_iter) : r.max_iter;
    if (r.tap_iter && pl->dump_frames < r.frames) {
        if (pl->dump_p) (void)hipFree(pl->dump_p);
        if (pl->dump_r) (void)hipFree(pl->dump_r);
        pl->dump_p = pl->dump_r = nullptr;
        if ((e = hipMalloc((void **)&pl->dump_p, (size_t)r.frames * pl->N * sizeof(float)))) return e;
        if ((e = hipMalloc((void **)&pl->dump_r, (size_t)r.frames * pl->E * sizeof(float)))) return e;
        pl->dump_frames = r.frames;
    }
    LdspArgs a{r.llr_dev, r.out_dev, r.iters_dev, r.summary, r.tap_iter ? pl->dump_p : nullptr,
               r.tap_iter ? pl->dump_r : nullptr, pl->recs, pl->zf, pl->hdr, pl->pack, pl->col_slot, pl->layer_e0,
               r.frames, r.out_dev ? r.out_bytes : 0, pl->N, pl->E, r.K, pl->z, pl->layers, pl->nb, pl->lds_cols,
               r.max_iter, rounds, r.early_term};
    const unsigned grid = (unsigned)std::min<int64_t>((r.frames + pl->wg_frames - 1) / pl->wg_frames, pl->grid);
    if (!pl->eligible || grid == 0) return hipErrorInvalidValue;
    if (pl->flood == 1 && pl->wg_frames > 1) flood_ldsp_packed_kernel<true><<<grid, 64, pl->lds_bytes, s>>>(a, pl->wg_frames);
    else if (pl->flood && pl->wg_frames > 1) flood_ldsp_packed_kernel<false><<<grid, 64, pl->lds_bytes, s>>>(a, pl->wg_frames);
    else if (pl->flood) flood_ldsp_kernel_for(pl->maxw, pl->flood)<<<grid, pl->block, pl->lds_bytes, s>>>(a);
    else if (pl->wg_frames > 1) layered_ldsp_packed_kernel<0><<<grid, 64, pl->lds_bytes, s>>>(a, pl->wg_frames);
    else ldsp_kernel_for(pl->maxw)<<<grid, pl->block, pl->lds_bytes, s>>>(a);
    *launched = rounds;
    return hipGetLastError();
}

#endif  /* LDPC_ENGINE_LDSP */

}  // namespace ldpc
