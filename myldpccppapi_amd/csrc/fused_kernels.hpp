/*
 * fused_kernels.hpp -- whole-decode-in-one-launch layered min-sum for SHORT quasi-cyclic
 * codes: the MI355X counterpart of the reference's fused kernel decodeOnceTDMP
 * (decodeCL.c:307-426, launched by Coder::decodeOnceTDMPCL, MyLdpc.cpp:850-868).
 *
 * For codes of a few thousand bits, streaming the messages through HBM once per layer
 * launch (layered_kernels.hpp) is the wrong design: a frame's whole state -- posteriors
 * P[N] and messages R[E], (N+E)*4 B = 9.6 KB at N = 576, 38 KB at N = 2304 -- fits in LDS
 * (160 KB per CU).  So, like the reference, one frame is decoded start to finish by one
 * group of lanes with its state on chip, and the launch count drops from
 * layers*iterations to one.  Unlike the reference:
 *   - the lanes of a wave are the z rows of the current layer, so P is read and written at
 *     consecutive LDS addresses (column = block*z + (row + shift) mod z): conflict-free;
 *     R is kept layer-major [layer][k][row] for the same reason (the reference keeps R in
 *     private arrays indexed [12][24] and rebuilds a column table per work-item);
 *   - the circulant structure is DETECTED from the edge list (any QC code, not only the
 *     802.16e seeds; any z, not z <= 127 / N <= 32767) and held as a small table of
 *     (block column, shift) pairs that every lane reads through the scalar cache;
 *   - codes with z <= 64 need no barrier at all: one 64-lane wave owns one frame (LDS
 *     operations of a wave are ordered); larger z use ceil(z/64) waves per frame and
 *     workgroup barriers between layers; several frames share a CU's LDS (4 at N = 2304);
 *   - the missing barrier after the posterior fill and the uninitialised `bInd` of the
 *     reference kernel do not exist here.
 * Arithmetic and results are those of layer_kernel / the oracle's layered decoder, bit for
 * bit (same fp32 operations in the same order).
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <algorithm>
#include <cstdlib>
#include <vector>

#include "layered_kernels.hpp"
#include "ldpc_expf.h"
#include "tune.hpp"

namespace ldpc {

struct FusedArgs {
    const float *__restrict__ llr;        /* [frames][N] frame-major, as the caller gives it */
    uint8_t *__restrict__ out;            /* packed bytes, toChar layout */
    int32_t *__restrict__ iters;          /* [frames] or nullptr */
    int32_t *__restrict__ summary;        /* [2]: max iters, converged frames */
    float *__restrict__ dump_p;           /* [frames][N] or nullptr (taps) */
    float *__restrict__ dump_r;           /* [frames][E] reference edge order, or nullptr */
    uint8_t *__restrict__ conv;           /* [frames] 1 = syndrome clean, or nullptr */
    float *__restrict__ dump_q;           /* sum-product taps: [frames][E] q0-q1 */
    uint8_t *__restrict__ dump_b;         /* sum-product taps: [frames][N] hard bits */
    float llr_scale;                      /* sum-product: exp(llr_scale * y), decodeCL.c:9 */
    const int32_t *__restrict__ layer_ptr;/* [layers+1] into the entry tables */
    const int32_t *__restrict__ ent_bc;   /* block column of entry */
    const int32_t *__restrict__ ent_sh;   /* circulant shift of entry */
    const int32_t *__restrict__ layer_e0; /* [layers] edge id of the layer's first edge */
    const int32_t *__restrict__ ent_pack; /* [layers][pack_w]: (block column << 16) | shift, padded: ONE wide
                                             scalar load per layer instead of two per edge (the per-edge
                                             scalar loads, ~45 serialised K$ round trips per layer step,
                                             were what bounded the first version of these kernels) */
    int32_t pack_w;
    /* column view for the flooding variant: block column bc is met by entries
     * bcol_ptr[bc]..bcol_ptr[bc+1], ascending layer = ascending row */
    const int32_t *__restrict__ bcol_ptr; /* [N/z + 1] */
    const int32_t *__restrict__ bcol_e0;  /* LDS slot of (layer, k): layer_e0[l] + k*z */
    const int32_t *__restrict__ bcol_sh;  /* shift of that block */
    int64_t frames, out_bytes;
    int32_t N, E, K, z, layers, max_iter, rounds, early_term;
};

/* One row of one layer (decodeCL.c:345-383) for lane-row r.  DMAX > 0: the row's d <= DMAX
 * inputs are requested from LDS back to back (a row's columns are distinct, rows of a layer
 * share none, so nothing aliases), the product / two-minimum chain runs in registers, and the
 * results are stored once.  The run-time loop version (DMAX == 0) re-reads P between its two
 * passes like the reference; it serialises ~2d LDS round trips per row and was the reason the
 * first fused kernel ran no faster than streaming HBM at full work.  Same values either way. */
template <int DMAX>
__device__ __forceinline__ void fused_layer_row(float *P, float *Rl, const int32_t *bc, const int32_t *sh,
                                                const int32_t *pk, int d, int z, int r)
{
    if (DMAX == 0) {
        float prod = 1.0f, b = 1000.0f, c = 1001.0f;   /* decodeCL.c:346-348 */
        int bind = -1;
        for (int k = 0; k < d; ++k) {                  /* :350-367 */
            int t = r + sh[k];
            t = t >= z ? t - z : t;
            const int col = bc[k] * z + t;
            const float q = P[col] - Rl[k * z + r];
            prod *= q;
            P[col] = q;                                /* :357 parks q in lP */
            const float mag = __builtin_fabsf(q);
            if (mag <= b) { c = b; b = mag; bind = k; }
            else if (mag > b && mag <= c) { c = mag; }
        }
        const float sa = cl_sign(prod);                /* :369 */
        const float ab = sa * b, ac = sa * c;
        for (int k = 0; k < d; ++k) {                  /* :371-383 */
            int t = r + sh[k];
            t = t >= z ? t - z : t;
            const int col = bc[k] * z + t;
            const float q = P[col];
            const float rn = cl_sign(q) * ((k == bind) ? ac : ab);
            Rl[k * z + r] = rn;
            P[col] = q + rn;
        }
        return;
    }
    constexpr int DM = DMAX > 0 ? DMAX : 1;
    float q[DM];
    int col[DM], ent[DM];
#pragma unroll
    for (int k = 0; k < DM; ++k) ent[k] = pk[k];           /* unconditional, contiguous: s_load_dwordxN */
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        int t = r + (ent[k] & 0xffff);
        t = t >= z ? t - z : t;
        col[k] = (ent[k] >> 16) * z + t;
    }
#pragma unroll
    for (int k = 0; k < DM; ++k)
        if (k < d) q[k] = P[col[k]] - Rl[k * z + r];
    float prod = 1.0f, b = 1000.0f, c = 1001.0f;
    int bind = -1;
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        if (k < d) {
            prod *= q[k];
            const float mag = __builtin_fabsf(q[k]);
            if (mag <= b) { c = b; b = mag; bind = k; }
            else if (mag > b && mag <= c) { c = mag; }
        }
    }
    const float sa = cl_sign(prod);
    const float ab = sa * b, ac = sa * c;
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        if (k < d) {
            const float rn = cl_sign(q[k]) * ((k == bind) ? ac : ab);
            Rl[k * z + r] = rn;
            P[col[k]] = q[k] + rn;
        }
    }
}

/* Row parity of the hard decisions P < 0 (decodeCL.c:393-404), loads first as above. */
/* hard decision: P < 0 (fused reference kernels, decodeCL.c:389,541) or !(P > 0) (MS chain, :161-165) */
template <bool NOTPOS> __device__ __forceinline__ int fused_bit(float p)
{
    return NOTPOS ? (!(p > 0.0f) ? 1 : 0) : ((p < 0.0f) ? 1 : 0);
}

template <int DMAX, bool NOTPOS = false>
__device__ __forceinline__ int fused_row_parity(const float *P, const int32_t *bc, const int32_t *sh,
                                                const int32_t *pk, int d, int z, int r)
{
    int par = 0;
    if (DMAX == 0) {
        for (int k = 0; k < d; ++k) {
            int t = r + sh[k];
            t = t >= z ? t - z : t;
            par ^= fused_bit<NOTPOS>(P[bc[k] * z + t]);
        }
        return par;
    }
    constexpr int DM = DMAX > 0 ? DMAX : 1;
    float v[DM];
    int ent[DM];
#pragma unroll
    for (int k = 0; k < DM; ++k) ent[k] = pk[k];
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        int t = r + (ent[k] & 0xffff);
        t = t >= z ? t - z : t;
        v[k] = P[(ent[k] >> 16) * z + t];                  /* padded entries repeat entry 0: valid address */
    }
#pragma unroll
    for (int k = 0; k < DM; ++k)
        if (k < d) par ^= fused_bit<NOTPOS>(v[k]);
    return par;
}

/* One workgroup = one frame = MW waves (MW = ceil(z/64)); MW == 1 needs no barriers. */
template <int MW, int DMAX>
__global__ __launch_bounds__(64 * MW) void fused_layered_kernel(const FusedArgs a)
{
    extern __shared__ float lds[];
    const int lane_in_frame = (int)threadIdx.x;
    const int64_t frame = (int64_t)blockIdx.x;
    float *P = lds;
    float *R = P + a.N;
    constexpr int LANES = 64 * MW;
    const int z = a.z;
    const int r = lane_in_frame;               /* row inside the layer */

    /* MW == 1: a wave's LDS operations execute in program order, so other lanes' earlier
     * writes are visible to later reads; the fence only stops the compiler from reordering. */
    auto sync = [&]() {
        if (MW == 1) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    };

    /* lP = postCode, lR = 0 (decodeCL.c:331-340) */
    const float *y = a.llr + (size_t)frame * a.N;
    for (int n = lane_in_frame; n < a.N; n += LANES) P[n] = y[n];
    for (int e = lane_in_frame; e < a.E; e += LANES) R[e] = 0.0f;
    sync();

    int time = 0;
    bool clean = false;
    while (true) {
        for (int l = 0; l < a.layers; ++l) {
            const int p0 = a.layer_ptr[l], d = a.layer_ptr[l + 1] - p0;
            float *Rl = R + a.layer_e0[l];              /* [d][z] */
            if (r < z) fused_layer_row<DMAX>(P, Rl, a.ent_bc + p0, a.ent_sh + p0, a.ent_pack + (size_t)l * a.pack_w, d, z, r);
            sync();                                            /* :385 */
        }
        /* bits = P < 0, row parities (decodeCL.c:387-404) */
        int bad = 0;
        if (r < z) {
            for (int l = 0; l < a.layers; ++l) {
                const int p0 = a.layer_ptr[l], d = a.layer_ptr[l + 1] - p0;
                bad |= fused_row_parity<DMAX>(P, a.ent_bc + p0, a.ent_sh + p0, a.ent_pack + (size_t)l * a.pack_w, d, z, r);
            }
        }
        const int any_bad = (MW == 1) ? (__ballot(bad != 0) != 0ull) : __syncthreads_or(bad);
        ++time;
        clean = !any_bad;
        if ((clean && a.early_term) || time == a.rounds) break; /* :407-410 */
        sync();
    }
    sync();

    /* toChar (decodeCL.c:414-423): byte j = bits 8j..8j+7 of the first K */
    const int64_t base = frame * (int64_t)a.K / 8;
    for (int j = lane_in_frame; j < a.K / 8; j += LANES) {
        unsigned byte = 0;
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) byte |= (P[j * 8 + bit] < 0.0f ? 1u : 0u) << bit;
        if (base + j < a.out_bytes) a.out[base + j] = (uint8_t)byte;
    }
    if (a.dump_p)
        for (int n = lane_in_frame; n < a.N; n += LANES) a.dump_p[(size_t)frame * a.N + n] = P[n];
    if (a.dump_r) {
        /* back to the reference's row-major edge order: edge (layer l, row r, k) */
        for (int l = 0; l < a.layers; ++l) {
            const int d = a.layer_ptr[l + 1] - a.layer_ptr[l], e0 = a.layer_e0[l];
            for (int i = lane_in_frame; i < d * z; i += LANES) {
                const int rr = i / d, k = i % d;
                a.dump_r[(size_t)frame * a.E + e0 + i] = R[e0 + k * z + rr];
            }
        }
    }
    if (lane_in_frame == 0) {
        /* iters: first clean iteration, else max_iter (a frame clean only at the cap counts as
         * converged at `time`, like state_kernel) */
        const int it = clean ? time : a.max_iter;
        if (a.iters) a.iters[frame] = it;
        if (a.conv) a.conv[frame] = clean ? 1 : 0;
        atomicMax(&a.summary[0], it);
        if (clean) atomicAdd(&a.summary[1], 1);
    }
}

/* z <= 32: G = 64 / z frames share one wave (lanes [g*z, (g+1)*z) = frame g's rows), so a
 * z = 24 code (the reference's own Test.cpp configuration) uses 48 of 64 lanes instead of 24.
 * Each frame keeps its own LDS image and leaves the loop on its own clean syndrome (its lanes
 * go idle); the wave runs until its last frame is done.  Same arithmetic, same results. */
template <int DMAX>
__global__ __launch_bounds__(64) void fused_layered_packed_kernel(const FusedArgs a, const int G)
{
    extern __shared__ float lds[];
    const int lane = (int)threadIdx.x;
    const int z = a.z;
    const int g = lane / z;
    const int r = lane - g * z;
    const int64_t frame = (int64_t)blockIdx.x * G + g;
    const bool mine = g < G && frame < a.frames;             /* lane belongs to a real frame */
    float *P = lds + (size_t)(mine ? g : 0) * (a.N + a.E);
    float *R = P + a.N;
    auto sync = [&]() {
        __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
        __builtin_amdgcn_wave_barrier();
    };
    if (mine) {
        const float *y = a.llr + (size_t)frame * a.N;
        for (int n = r; n < a.N; n += z) P[n] = y[n];
        for (int e = r; e < a.E; e += z) R[e] = 0.0f;
    }
    sync();
    const uint64_t grp_mask = (z >= 64 ? ~0ull : ((1ull << z) - 1ull)) << (g < G ? g * z : 0);
    int time = 0, my_iters = a.max_iter;
    bool active = mine, clean = false;
    while (__ballot(active) != 0ull) {
        for (int l = 0; l < a.layers; ++l) {
            const int p0 = a.layer_ptr[l], d = a.layer_ptr[l + 1] - p0;
            float *Rl = R + a.layer_e0[l];
            if (active) fused_layer_row<DMAX>(P, Rl, a.ent_bc + p0, a.ent_sh + p0, a.ent_pack + (size_t)l * a.pack_w, d, z, r);
            sync();
        }
        int bad = 0;
        if (active) {
            for (int l = 0; l < a.layers; ++l) {
                const int p0 = a.layer_ptr[l], d = a.layer_ptr[l + 1] - p0;
                bad |= fused_row_parity<DMAX>(P, a.ent_bc + p0, a.ent_sh + p0, a.ent_pack + (size_t)l * a.pack_w, d, z, r);
            }
        }
        const uint64_t bad_mask = __ballot(bad != 0);
        ++time;
        if (active) {
            clean = (bad_mask & grp_mask) == 0ull;
            if ((clean && a.early_term) || time == a.rounds) {
                active = false;
                my_iters = clean ? time : a.max_iter;
            }
        }
        sync();
    }
    if (!mine) return;
    const int64_t base = frame * (int64_t)a.K / 8;
    for (int j = r; j < a.K / 8; j += z) {
        unsigned byte = 0;
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) byte |= (P[j * 8 + bit] < 0.0f ? 1u : 0u) << bit;
        if (base + j < a.out_bytes) a.out[base + j] = (uint8_t)byte;
    }
    if (a.dump_p)
        for (int n = r; n < a.N; n += z) a.dump_p[(size_t)frame * a.N + n] = P[n];
    if (a.dump_r) {
        for (int l = 0; l < a.layers; ++l) {
            const int d = a.layer_ptr[l + 1] - a.layer_ptr[l], e0 = a.layer_e0[l];
            for (int i = r; i < d * z; i += z) {
                const int rr = i / d, k = i % d;
                a.dump_r[(size_t)frame * a.E + e0 + i] = R[e0 + k * z + rr];
            }
        }
    }
    if (r == 0) {
        if (a.iters) a.iters[frame] = my_iters;
        atomicMax(&a.summary[0], my_iters);
        if (clean) atomicAdd(&a.summary[1], 1);
    }
}

/* Flooding row (decodeCL.c:482-512): R_k = sign(q_k) * sign(prod) * (k == argmin ? min2 : min1)
 * with q_k = P[col_k] - R_k; P is only read. */
template <int DMAX>
__device__ __forceinline__ void fused_flood_row(const float *P, float *Rl, const int32_t *bc, const int32_t *sh,
                                                const int32_t *pk, int d, int z, int r)
{
    if (DMAX == 0) {
        float prod = 1.0f, b = 1000.0f, c = 1001.0f;
        int bind = -1;
        for (int k = 0; k < d; ++k) {                  /* :486-501 */
            int t = r + sh[k];
            t = t >= z ? t - z : t;
            const float q = P[bc[k] * z + t] - Rl[k * z + r];
            Rl[k * z + r] = cl_sign(q);
            prod *= q;
            const float mag = __builtin_fabsf(q);
            if (mag <= b) { c = b; b = mag; bind = k; }
            else if (mag > b && mag <= c) { c = mag; }
        }
        const float sa = cl_sign(prod);                /* :502 */
        const float ab = sa * b, ac = sa * c;
        for (int k = 0; k < d; ++k)                    /* :504-513 */
            Rl[k * z + r] = Rl[k * z + r] * ((k == bind) ? ac : ab);
        return;
    }
    constexpr int DM = DMAX > 0 ? DMAX : 1;
    float q[DM];
    int col[DM], ent[DM];
#pragma unroll
    for (int k = 0; k < DM; ++k) ent[k] = pk[k];
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        int t = r + (ent[k] & 0xffff);
        t = t >= z ? t - z : t;
        col[k] = (ent[k] >> 16) * z + t;
    }
#pragma unroll
    for (int k = 0; k < DM; ++k)
        if (k < d) q[k] = P[col[k]] - Rl[k * z + r];
    float prod = 1.0f, b = 1000.0f, c = 1001.0f;
    int bind = -1;
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        if (k < d) {
            prod *= q[k];
            const float mag = __builtin_fabsf(q[k]);
            if (mag <= b) { c = b; b = mag; bind = k; }
            else if (mag > b && mag <= c) { c = mag; }
        }
    }
    const float sa = cl_sign(prod);
    const float ab = sa * b, ac = sa * c;
#pragma unroll
    for (int k = 0; k < DM; ++k)
        if (k < d) Rl[k * z + r] = cl_sign(q[k]) * ((k == bind) ? ac : ab);
}

/* Flooding row with the arithmetic of the MS kernel chain / decodeCPU (refreshRMS,
 * decodeCL.c:126-147): sign = XOR of (q_j < 0) over the others, magnitude = fmin chain over the
 * others' |q_j| from 1000 (two-smallest form, exact).  q_k = P[col_k] - R_k is refreshQMS
 * (:185) applied on the fly. */
template <int DMAX>
__device__ __forceinline__ void fused_chain_row(const float *P, float *Rl, const int32_t *bc, const int32_t *sh,
                                                const int32_t *pk, int d, int z, int r)
{
    if (DMAX == 0) {
        float m1 = 1000.0f, m2 = 1000.0f;
        int idx = -1;
        unsigned par = 0;
        for (int k = 0; k < d; ++k) {
            int t = r + sh[k];
            t = t >= z ? t - z : t;
            const float q = P[bc[k] * z + t] - Rl[k * z + r];
            Rl[k * z + r] = q;                                  /* parked for the second pass */
            const float mag = __builtin_fabsf(q);
            par ^= (q < 0.0f) ? 1u : 0u;
            if (mag < m1) { m2 = m1; m1 = mag; idx = k; }
            else if (mag < m2) { m2 = mag; }
        }
        for (int k = 0; k < d; ++k) {
            const float q = Rl[k * z + r];
            const float b = (k == idx) ? m2 : m1;
            Rl[k * z + r] = (par ^ ((q < 0.0f) ? 1u : 0u)) ? -b : b;
        }
        return;
    }
    constexpr int DM = DMAX > 0 ? DMAX : 1;
    float q[DM];
    int ent[DM];
#pragma unroll
    for (int k = 0; k < DM; ++k) ent[k] = pk[k];
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        int t = r + (ent[k] & 0xffff);
        t = t >= z ? t - z : t;
        if (k < d) q[k] = P[(ent[k] >> 16) * z + t] - Rl[k * z + r];
    }
    float m1 = 1000.0f, m2 = 1000.0f;
    int idx = -1;
    unsigned par = 0;
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        if (k < d) {
            const float mag = __builtin_fabsf(q[k]);
            par ^= (q[k] < 0.0f) ? 1u : 0u;
            if (mag < m1) { m2 = m1; m1 = mag; idx = k; }
            else if (mag < m2) { m2 = mag; }
        }
    }
#pragma unroll
    for (int k = 0; k < DM; ++k) {
        if (k < d) {
            const float b = (k == idx) ? m2 : m1;
            Rl[k * z + r] = (par ^ ((q[k] < 0.0f) ? 1u : 0u)) ? -b : b;
        }
    }
}

/* Flooding counterpart: the reference's fused kernel decodeOnceMS (decodeCL.c:432-567,
 * DecodeMSCL).  Every iteration: all rows from the same posteriors (check node with the
 * product sign and the 1000/1001 two-minimum rule, :482-512), then every posterior rebuilt
 * as y + sum of its column's R in ascending row order (:515-531), bits = P < 0, syndrome.
 * Same LDS layout and lane mapping as fused_layered_kernel; y is re-read from global memory
 * (L2) instead of being kept in LDS. */
/* CHAIN = false: decodeOnceMS arithmetic (DecodeMSCL).  CHAIN = true: the MS kernel chain's
 * arithmetic (DecodeMS / DecodeCPU: XOR sign, fmin from 1000, bit = !(p > 0), decodeCL.c:126-186),
 * so that short QC codes decode in one launch in those modes too, bit-identical to the
 * streaming kernels. */
template <int MW, int DMAX, bool CHAIN>
__global__ __launch_bounds__(64 * MW) void fused_flood_kernel(const FusedArgs a)
{
    extern __shared__ float lds[];
    const int tid = (int)threadIdx.x;
    const int64_t frame = (int64_t)blockIdx.x;
    float *P = lds;
    float *R = P + a.N;
    constexpr int LANES = 64 * MW;
    const int z = a.z;
    const int r = tid;
    auto sync = [&]() {
        if (MW == 1) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    };
    const float *y = a.llr + (size_t)frame * a.N;
    for (int n = tid; n < a.N; n += LANES) P[n] = y[n];       /* :469-471 */
    for (int e = tid; e < a.E; e += LANES) R[e] = 0.0f;       /* :474-476 */
    sync();
    int time = 0;
    bool clean = false;
    while (true) {
        if (r < z) {
            for (int l = 0; l < a.layers; ++l) {               /* rows never touch P here */
                const int p0 = a.layer_ptr[l], d = a.layer_ptr[l + 1] - p0;
                if (CHAIN)
                    fused_chain_row<DMAX>(P, R + a.layer_e0[l], a.ent_bc + p0, a.ent_sh + p0, a.ent_pack + (size_t)l * a.pack_w, d, z, r);
                else
                    fused_flood_row<DMAX>(P, R + a.layer_e0[l], a.ent_bc + p0, a.ent_sh + p0, a.ent_pack + (size_t)l * a.pack_w, d, z, r);
            }
        }
        sync();                                                /* :514 */
        for (int n = tid; n < a.N; n += LANES) {               /* :515-531 */
            const int bc = n / z, t = n - bc * z;
            float tmp = y[n];
            for (int j = a.bcol_ptr[bc]; j < a.bcol_ptr[bc + 1]; ++j) {
                int rr = t - a.bcol_sh[j];
                rr = rr < 0 ? rr + z : rr;
                tmp += R[a.bcol_e0[j] + rr];
            }
            P[n] = tmp;
        }
        sync();                                                /* :532 */
        int bad = 0;
        if (r < z) {
            for (int l = 0; l < a.layers; ++l) {
                const int p0 = a.layer_ptr[l], d = a.layer_ptr[l + 1] - p0;
                bad |= fused_row_parity<DMAX, CHAIN>(P, a.ent_bc + p0, a.ent_sh + p0, a.ent_pack + (size_t)l * a.pack_w, d, z, r);
            }
        }
        const int any_bad = (MW == 1) ? (__ballot(bad != 0) != 0ull) : __syncthreads_or(bad);
        ++time;
        clean = !any_bad;
        if ((clean && a.early_term) || time == a.rounds) break; /* :555-559 */
    }
    sync();
    const int64_t base = frame * (int64_t)a.K / 8;
    for (int j = tid; j < a.K / 8; j += LANES) {                /* :561-569 */
        unsigned byte = 0;
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) byte |= (unsigned)fused_bit<CHAIN>(P[j * 8 + bit]) << bit;
        if (base + j < a.out_bytes) a.out[base + j] = (uint8_t)byte;
    }
    if (a.dump_p)
        for (int n = tid; n < a.N; n += LANES) a.dump_p[(size_t)frame * a.N + n] = P[n];
    if (a.dump_r) {
        for (int l = 0; l < a.layers; ++l) {
            const int d = a.layer_ptr[l + 1] - a.layer_ptr[l], e0 = a.layer_e0[l];
            for (int i = tid; i < d * z; i += LANES) {
                const int rr = i / d, k = i % d;
                a.dump_r[(size_t)frame * a.E + e0 + i] = R[e0 + k * z + rr];
            }
        }
    }
    if (tid == 0) {
        const int it = clean ? time : a.max_iter;
        if (a.iters) a.iters[frame] = it;
        atomicMax(&a.summary[0], it);
        if (clean) atomicAdd(&a.summary[1], 1);
    }
}

/* Sum-product in the probability domain (decodeCL.c:3-108: decodeInit, refreshR, hardDecision,
 * checkResult, refreshQ), whole decode in LDS for short QC codes -- same stored quantities and
 * the same fp32 operation order as check_kernel<SP> / var_kernel<SP>: Q = q0-q1 and R = r0-r1
 * per edge, T = exp(scale*y) per column, products left to right in ascending edge id with the
 * own edge skipped, IEEE divides, ties and NaN keep the previous hard bit.
 * LDS per frame: T[N] + Q[E] + R[E] floats + N bit bytes.  Rows: lanes = the z rows of a layer
 * (as in the other fused kernels).  Columns: lanes = consecutive columns; a column of block
 * column bc at offset t meets, for its j-th entry (layer, k, shift), row (t - shift) mod z. */
template <int MW, int DMAX, int DVMAX>
__global__ __launch_bounds__(64 * MW) void fused_sp_kernel(const FusedArgs a)
{
    extern __shared__ float lds[];
    const int tid = (int)threadIdx.x;
    const int64_t frame = (int64_t)blockIdx.x;
    float *T = lds;
    float *Q = T + a.N;
    float *R = Q + a.E;
    uint8_t *bits = reinterpret_cast<uint8_t *>(R + a.E);
    constexpr int LANES = 64 * MW;
    const int z = a.z;
    const int r = tid;
    auto sync = [&]() {
        if (MW == 1) {
            __builtin_amdgcn_fence(__ATOMIC_SEQ_CST, "wavefront");
            __builtin_amdgcn_wave_barrier();
        } else {
            __syncthreads();
        }
    };
    /* decodeInit, decodeCL.c:3-22 */
    const float *y = a.llr + (size_t)frame * a.N;
    for (int n = tid; n < a.N; n += LANES) {
        const int bc = n / z, t = n - bc * z;
        const float e = ldpc_expf(a.llr_scale * y[n]);
        T[n] = e;
        bits[n] = 0;
        const float q = e / (1.0f + e) - 1.0f / (1.0f + e);
        for (int j = a.bcol_ptr[bc]; j < a.bcol_ptr[bc + 1]; ++j) {
            int rr = t - a.bcol_sh[j];
            rr = rr < 0 ? rr + z : rr;
            Q[a.bcol_e0[j] + rr] = q;
        }
    }
    sync();
    int time = 0;
    bool clean = false;
    while (true) {
        /* refreshR, decodeCL.c:25-41 */
        if (r < z) {
            for (int l = 0; l < a.layers; ++l) {
                const int d = a.layer_ptr[l + 1] - a.layer_ptr[l];
                const int e0 = a.layer_e0[l];
                float x[DMAX];
#pragma unroll
                for (int k = 0; k < DMAX; ++k)
                    if (k < d) x[k] = Q[e0 + k * z + r];
                float pre = 1.0f;                  /* 1.0f * x is exact: the chain starts at the first factor */
#pragma unroll
                for (int k = 0; k < DMAX; ++k) {
                    if (k < d) {
                        float p = pre;
#pragma unroll
                        for (int j = k + 1; j < DMAX; ++j)
                            if (j < d) p *= x[j];
                        R[e0 + k * z + r] = p;
                        pre *= x[k];
                    }
                }
            }
        }
        sync();
        /* hardDecision (:64-86) + refreshQ (:43-62), shared prefix products */
        for (int n = tid; n < a.N; n += LANES) {
            const int bc = n / z, t = n - bc * z;
            const int jb = a.bcol_ptr[bc], dv = a.bcol_ptr[bc + 1] - jb;
            int slot[DVMAX];
            float r0[DVMAX], r1[DVMAX];
#pragma unroll
            for (int j = 0; j < DVMAX; ++j) {
                if (j < dv) {
                    int rr = t - a.bcol_sh[jb + j];
                    rr = rr < 0 ? rr + z : rr;
                    slot[j] = a.bcol_e0[jb + j] + rr;
                    const float dj = R[slot[j]];
                    r0[j] = (1.0f + dj) * 0.5f;
                    r1[j] = (1.0f - dj) * 0.5f;
                }
            }
            const float den = 1.0f + T[n];
            float pre0 = T[n] / den, pre1 = 1.0f / den;
#pragma unroll
            for (int k = 0; k < DVMAX; ++k) {
                if (k < dv) {
                    float t0 = pre0, t1 = pre1;
#pragma unroll
                    for (int j = k + 1; j < DVMAX; ++j)
                        if (j < dv) { t0 *= r0[j]; t1 *= r1[j]; }
                    const float ssum = t0 + t1;
                    Q[slot[k]] = t0 / ssum - t1 / ssum;
                    pre0 *= r0[k];
                    pre1 *= r1[k];
                }
            }
            if (pre0 > pre1) bits[n] = 0;
            else if (pre0 < pre1) bits[n] = 1;          /* ties / NaN: unchanged, :78-82 */
        }
        sync();
        /* checkResult, :88-108 */
        int bad = 0;
        if (r < z) {
            for (int l = 0; l < a.layers; ++l) {
                const int p0 = a.layer_ptr[l], d = a.layer_ptr[l + 1] - p0;
                const int32_t *pk = a.ent_pack + (size_t)l * a.pack_w;
                int par = 0;
                for (int k = 0; k < d; ++k) {
                    int t = r + (pk[k] & 0xffff);
                    t = t >= z ? t - z : t;
                    par ^= bits[(pk[k] >> 16) * z + t];
                }
                bad |= par;
            }
        }
        const int any_bad = (MW == 1) ? (__ballot(bad != 0) != 0ull) : __syncthreads_or(bad);
        ++time;
        clean = !any_bad;
        if ((clean && a.early_term) || time == a.rounds) break;    /* MyLdpc.cpp:1031-1039 */
    }
    sync();
    const int64_t base = frame * (int64_t)a.K / 8;
    for (int j = tid; j < a.K / 8; j += LANES) {                    /* toChar, :188-199 */
        unsigned byte = 0;
#pragma unroll
        for (int bit = 0; bit < 8; ++bit) byte |= (unsigned)bits[j * 8 + bit] << bit;
        if (base + j < a.out_bytes) a.out[base + j] = (uint8_t)byte;
    }
    if (a.dump_p)
        for (int n = tid; n < a.N; n += LANES) {
            a.dump_p[(size_t)frame * a.N + n] = T[n];
            a.dump_b[(size_t)frame * a.N + n] = bits[n];
        }
    if (a.dump_r) {
        for (int l = 0; l < a.layers; ++l) {
            const int d = a.layer_ptr[l + 1] - a.layer_ptr[l], e0 = a.layer_e0[l];
            for (int i = tid; i < d * z; i += LANES) {
                const int rr = i / d, k = i % d;
                a.dump_r[(size_t)frame * a.E + e0 + i] = R[e0 + k * z + rr];
                a.dump_q[(size_t)frame * a.E + e0 + i] = Q[e0 + k * z + rr];
            }
        }
    }
    if (tid == 0) {
        const int it = clean ? time : a.max_iter;
        if (a.iters) a.iters[frame] = it;
        atomicMax(&a.summary[0], it);
        if (clean) atomicAdd(&a.summary[1], 1);
    }
}

/* ---------------------------------------------------------------- host side */

struct FusedPlan {
    bool eligible = false;
    int32_t z = 0, layers = 0, N = 0, E = 0, M = 0, max_deg = 0;
    int32_t *layer_ptr = nullptr, *ent_bc = nullptr, *ent_sh = nullptr, *layer_e0 = nullptr; /* device */
    int32_t *bcol_ptr = nullptr, *bcol_e0 = nullptr, *bcol_sh = nullptr;
    int32_t *ent_pack = nullptr;
    int32_t pack_w = 0;
    float *dump_p = nullptr, *dump_r = nullptr, *dump_q = nullptr;
    uint8_t *conv = nullptr, *dump_b = nullptr;
    int32_t max_col_deg = 0;
    bool eligible_sp = false;
    size_t lds_sp = 0;
    int64_t dump_frames = 0;
    size_t lds_per_frame = 0;
};

inline void fused_plan_destroy(FusedPlan *pl)
{
    for (void *p : {(void *)pl->layer_ptr, (void *)pl->ent_bc, (void *)pl->ent_sh, (void *)pl->layer_e0,
                    (void *)pl->bcol_ptr, (void *)pl->bcol_e0, (void *)pl->bcol_sh, (void *)pl->ent_pack,
                    (void *)pl->dump_p, (void *)pl->dump_r, (void *)pl->conv, (void *)pl->dump_q, (void *)pl->dump_b})
        if (p) (void)hipFree(p);
    *pl = FusedPlan();
}

/* Detect the circulant structure of every layer: all z rows of a layer have the same
 * degree and row r's k-th edge sits in column bc_k*z + (r + s_k) mod z.  Returns true and
 * fills the host tables when the whole matrix has that form. */
inline bool fused_detect_qc(int32_t M, int32_t N, const std::vector<int32_t> &row_ptr,
                            const std::vector<int32_t> &cols, int32_t z, std::vector<int32_t> &layer_ptr,
                            std::vector<int32_t> &bc, std::vector<int32_t> &sh, std::vector<int32_t> &e0)
{
    if (z <= 0 || M % z || N % z) return false;
    const int layers = M / z;
    layer_ptr.assign(1, 0);
    bc.clear(); sh.clear(); e0.clear();
    for (int l = 0; l < layers; ++l) {
        const int m0 = l * z, d = row_ptr[m0 + 1] - row_ptr[m0];
        if (d <= 0) return false;
        e0.push_back(row_ptr[m0]);
        for (int k = 0; k < d; ++k) {
            const int c = cols[row_ptr[m0] + k];
            bc.push_back(c / z);
            sh.push_back(c % z);
        }
        for (int r = 1; r < z; ++r) {
            const int m = m0 + r;
            if (row_ptr[m + 1] - row_ptr[m] != d) return false;
            for (int k = 0; k < d; ++k) {
                const int want = bc[layer_ptr[l] + k] * z + (r + sh[layer_ptr[l] + k]) % z;
                if (cols[row_ptr[m] + k] != want) return false;
            }
        }
        /* row-major order inside a row = ascending block column; a block column may appear once */
        for (int k = 1; k < d; ++k)
            if (bc[layer_ptr[l] + k] <= bc[layer_ptr[l] + k - 1]) return false;
        layer_ptr.push_back(layer_ptr[l] + d);
    }
    return true;
}

constexpr size_t kFusedMaxLdsPerFrame = 48 * 1024;   /* >= 3 frames per CU (160 KB LDS) */

inline hipError_t fused_plan_create(FusedPlan *pl, int32_t M, int32_t N, int64_t E,
                                    const std::vector<int32_t> &row_ptr, const std::vector<int32_t> &cols,
                                    int32_t z)
{
    std::vector<int32_t> lp, bc, sh, e0;
    pl->eligible = false;
    if (z > 256 || (size_t)(N + E) * 4 > kFusedMaxLdsPerFrame) return hipSuccess;
    if (!fused_detect_qc(M, N, row_ptr, cols, z, lp, bc, sh, e0)) return hipSuccess;
    pl->z = z; pl->layers = M / z; pl->N = N; pl->E = (int32_t)E; pl->M = M;
    pl->max_deg = 0;
    for (size_t l = 0; l + 1 < lp.size(); ++l) pl->max_deg = std::max(pl->max_deg, lp[l + 1] - lp[l]);
    pl->lds_per_frame = (size_t)(N + E) * 4;
    auto up = [](int32_t **dst, const std::vector<int32_t> &v) {
        hipError_t e = hipMalloc((void **)dst, v.size() * sizeof(int32_t));
        if (e != hipSuccess) return e;
        return hipMemcpy(*dst, v.data(), v.size() * sizeof(int32_t), hipMemcpyHostToDevice);
    };
    hipError_t e;
    if ((e = up(&pl->layer_ptr, lp)) || (e = up(&pl->ent_bc, bc)) || (e = up(&pl->ent_sh, sh)) ||
        (e = up(&pl->layer_e0, e0)))
        return e;
    /* padded (block column << 16 | shift) rows, one wide scalar load per layer */
    pl->pack_w = pl->max_deg <= 8 ? 8 : pl->max_deg <= 16 ? 16 : 24;
    {
        std::vector<int32_t> pk((size_t)(M / z) * pl->pack_w, 0);
        for (int l = 0; l < M / z; ++l)
            for (int k = 0; k < pl->pack_w; ++k) {
                const int j = lp[l] + (k < lp[l + 1] - lp[l] ? k : 0);
                pk[(size_t)l * pl->pack_w + k] = (bc[j] << 16) | sh[j];
            }
        if ((e = up(&pl->ent_pack, pk))) return e;
    }
    /* column view: entries of every block column in ascending layer order */
    const int nb = N / z, layers = M / z;
    std::vector<int32_t> cp((size_t)nb + 1, 0), ce0, csh;
    for (int l = 0; l < layers; ++l)
        for (int j = lp[l]; j < lp[l + 1]; ++j) ++cp[bc[j] + 1];
    for (int b = 0; b < nb; ++b) cp[b + 1] += cp[b];
    ce0.assign(cp[nb], 0); csh.assign(cp[nb], 0);
    std::vector<int32_t> fill(cp.begin(), cp.end() - 1);
    for (int l = 0; l < layers; ++l)
        for (int j = lp[l]; j < lp[l + 1]; ++j) {
            const int slot = fill[bc[j]]++;
            ce0[slot] = e0[l] + (j - lp[l]) * z;
            csh[slot] = sh[j];
        }
    if ((e = up(&pl->bcol_ptr, cp)) || (e = up(&pl->bcol_e0, ce0)) || (e = up(&pl->bcol_sh, csh))) return e;
    pl->max_col_deg = 0;
    for (int b = 0; b < nb; ++b) pl->max_col_deg = std::max(pl->max_col_deg, cp[b + 1] - cp[b]);
    pl->lds_sp = (size_t)(N + 2 * E) * 4 + (((size_t)N + 3) & ~(size_t)3);
    /* sum-product keeps Q and R (2 E floats): up to 96 KB per frame, e.g. (2304, 1152) = 70 KB, two
     * frames per CU -- for the small batches this path serves, latency counts, not occupancy */
    pl->eligible_sp = pl->max_deg <= 24 && pl->max_col_deg <= 8 && pl->lds_sp <= 96 * 1024;
    pl->eligible = true;
    return hipSuccess;
}

struct FusedRun {
    const float *llr_dev;
    int64_t frames;
    uint8_t *out_dev;
    int64_t out_bytes;
    int32_t *iters_dev;
    int32_t K, max_iter, tap_iter, early_term;
    int32_t *summary;
    int32_t flooding;   /* 0: layered (decodeOnceTDMP), 1: flooding (decodeOnceMS), 2: flooding with the MS chain's
                           arithmetic, 3: sum-product */
    float llr_scale;
    int32_t loop_rows;  /* tuning: run-time row loops instead of the unrolled widths */
    int32_t no_pack;    /* tuning: one frame per wave also for circulants of <= 32 rows */
};

/* instantiates every kernel above: compiled by engine_fused.hip only (LDPC_ENGINE_FUSED) */
#ifdef LDPC_ENGINE_FUSED
inline hipError_t fused_run(FusedPlan *pl, const FusedRun &r, hipStream_t s, int32_t *launched)
{
    hipError_t e;
    if ((e = hipMemsetAsync(r.summary, 0, 2 * sizeof(int32_t), s))) return e;
    const int rounds = r.tap_iter ? (r.tap_iter < r.max_iter ? r.tap_iter : r.max_iter) : r.max_iter;
    if (r.tap_iter && pl->dump_frames < r.frames) {
        for (void *p : {(void *)pl->dump_p, (void *)pl->dump_r, (void *)pl->dump_q, (void *)pl->dump_b})
            if (p) (void)hipFree(p);
        pl->dump_p = pl->dump_r = pl->dump_q = nullptr;
        pl->dump_b = nullptr;
        if ((e = hipMalloc((void **)&pl->dump_p, (size_t)r.frames * pl->N * sizeof(float)))) return e;
        if ((e = hipMalloc((void **)&pl->dump_r, (size_t)r.frames * pl->E * sizeof(float)))) return e;
        if ((e = hipMalloc((void **)&pl->dump_q, (size_t)r.frames * pl->E * sizeof(float)))) return e;
        if ((e = hipMalloc((void **)&pl->dump_b, (size_t)r.frames * pl->N))) return e;
        pl->dump_frames = r.frames;
    }
    FusedArgs a{r.llr_dev, r.out_dev, r.iters_dev, r.summary, r.tap_iter ? pl->dump_p : nullptr,
                r.tap_iter ? pl->dump_r : nullptr, nullptr, r.tap_iter ? pl->dump_q : nullptr,
                r.tap_iter ? pl->dump_b : nullptr, r.llr_scale, pl->layer_ptr, pl->ent_bc, pl->ent_sh,
                pl->layer_e0, pl->ent_pack, pl->pack_w, pl->bcol_ptr, pl->bcol_e0, pl->bcol_sh, r.frames, r.out_dev ? r.out_bytes : 0, pl->N, pl->E, r.K, pl->z, pl->layers,
                r.max_iter, rounds, r.early_term};
    const int mw = (pl->z + 63) / 64;
    const unsigned grid = (unsigned)r.frames;
    /* DMAX: unrolled row width (8 / 16 / 24), 0 = run-time loops for anything wider */
    const int dm = r.loop_rows ? 0 : (pl->max_deg <= 8 ? 8 : pl->max_deg <= 16 ? 16 : pl->max_deg <= 24 ? 24 : 0);
#define LDPC_FUSED_LAUNCH(KERNEL, MWV, ...)                                                      \
    do {                                                                                         \
        if (dm == 8) KERNEL<MWV, 8 __VA_ARGS__><<<grid, 64 * MWV, pl->lds_per_frame, s>>>(a);    \
        else if (dm == 16) KERNEL<MWV, 16 __VA_ARGS__><<<grid, 64 * MWV, pl->lds_per_frame, s>>>(a); \
        else if (dm == 24) KERNEL<MWV, 24 __VA_ARGS__><<<grid, 64 * MWV, pl->lds_per_frame, s>>>(a); \
        else KERNEL<MWV, 0 __VA_ARGS__><<<grid, 64 * MWV, pl->lds_per_frame, s>>>(a);            \
    } while (0)
#define LDPC_FUSED_BY_MW(KERNEL, ...)                                                            \
    switch (mw) {                                                                                \
    case 1: LDPC_FUSED_LAUNCH(KERNEL, 1, __VA_ARGS__); break;                                    \
    case 2: LDPC_FUSED_LAUNCH(KERNEL, 2, __VA_ARGS__); break;                                    \
    case 3: LDPC_FUSED_LAUNCH(KERNEL, 3, __VA_ARGS__); break;                                    \
    case 4: LDPC_FUSED_LAUNCH(KERNEL, 4, __VA_ARGS__); break;                                    \
    default: return hipErrorInvalidValue;                                                        \
    }
#define LDPC_COMMA ,
    if (r.flooding == 3) {
        switch (mw * 100 + dm) {
#define LDPC_SP_CASE(MWV, DMV) case MWV * 100 + DMV:                                                                     \
        if (pl->lds_sp > 60 * 1024 &&                                                                                   \
            (e = hipFuncSetAttribute((const void *)fused_sp_kernel<MWV, DMV, 8>, hipFuncAttributeMaxDynamicSharedMemorySize, \
                                     96 * 1024)))                                                                       \
            return e;                                                                                                   \
        fused_sp_kernel<MWV, DMV, 8><<<grid, 64 * MWV, pl->lds_sp, s>>>(a);                                             \
        break
            LDPC_SP_CASE(1, 8); LDPC_SP_CASE(1, 16); LDPC_SP_CASE(1, 24);
            LDPC_SP_CASE(2, 8); LDPC_SP_CASE(2, 16); LDPC_SP_CASE(2, 24);
            LDPC_SP_CASE(3, 8); LDPC_SP_CASE(3, 16); LDPC_SP_CASE(3, 24);
            LDPC_SP_CASE(4, 8); LDPC_SP_CASE(4, 16); LDPC_SP_CASE(4, 24);
#undef LDPC_SP_CASE
        default: return hipErrorInvalidValue;
        }
    } else if (r.flooding == 2) {
        LDPC_FUSED_BY_MW(fused_flood_kernel, LDPC_COMMA true)
    } else if (r.flooding) {
        LDPC_FUSED_BY_MW(fused_flood_kernel, LDPC_COMMA false)
    } else if (pl->z <= 32 && !r.no_pack) {
        int G = 64 / pl->z;
        while (G > 1 && (size_t)G * pl->lds_per_frame > 60 * 1024) --G;   /* default dynamic-LDS limit */
        const unsigned pgrid = (unsigned)((r.frames + G - 1) / G);
        const size_t plds = G * pl->lds_per_frame;
        if (dm == 8) fused_layered_packed_kernel<8><<<pgrid, 64, plds, s>>>(a, G);
        else if (dm == 16) fused_layered_packed_kernel<16><<<pgrid, 64, plds, s>>>(a, G);
        else if (dm == 24) fused_layered_packed_kernel<24><<<pgrid, 64, plds, s>>>(a, G);
        else fused_layered_packed_kernel<0><<<pgrid, 64, plds, s>>>(a, G);
    } else {
        LDPC_FUSED_BY_MW(fused_layered_kernel, )
    }
#undef LDPC_COMMA
#undef LDPC_FUSED_BY_MW
#undef LDPC_FUSED_LAUNCH
    *launched = rounds;
    return hipGetLastError();
}

#endif  /* LDPC_ENGINE_FUSED */

}  // namespace ldpc
