/*
 * layered_kernels.hpp -- layered (TDMP) min-sum on gfx950.
 *
 * Semantic model: the reference's fused kernel decodeOnceTDMP
 * (decodeCL.c:307-426), generalised from its WiMAX-seed addressing to any edge
 * list whose rows come in layers of `layer_rows` rows with pairwise disjoint
 * columns.  (The reference's host-layered path, MyLdpc.cpp:889-976, mis-sizes its
 * layers -- :907,958 -- and is not followed.)
 *
 * Per frame: posteriors P[N] (start: channel values) and messages R[E] (start 0).
 * One pass over layer l, every row of the layer independently (decodeCL.c:345-383):
 *     q_k = P[col_k] - R_k ; a = prod q_k (fp32, ascending k) ; two smallest |q_k|
 *     R_k = sign(q_k) * (sign(a) * (k == argmin ? min2 : min1)) ; P[col_k] = q_k + R_k
 * After the last layer: bits = P < 0, syndrome, stop when clean or at max_iter
 * (:387-412).
 *
 * MI355X layout: frames are the lanes, P[tile][n][F] and R[tile][e][F] with
 * F = 64*V, as in flood_kernels.hpp.  One launch per layer (rows of a layer are
 * independent, layers are ordered); a row's R segments are consecutive, its P
 * segments are gathered.  16*E bytes per frame-iteration (R and P each read and
 * written once per edge).  Frozen frames keep their hard bits (bit masks), their
 * P/R keep evolving unread -- no per-lane predication in the hot loop.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <map>
#include <vector>

#include "flood_kernels.hpp"

namespace ldpc {

struct LayerArgs {
    float *__restrict__ P;                /* [T][N][F] */
    float *__restrict__ R;                /* [T][E][F] */
    const int32_t *__restrict__ cls_e0;   /* rows of this (layer, degree) group: first edge id */
    const int32_t *__restrict__ edge_col; /* [E] */
    uint64_t *hard;                       /* [T][N][V] */
    const uint64_t *__restrict__ done;    /* [T][V] */
    int64_t E;
    int32_t N;
    int32_t n_rows;
    int32_t rows_per_wave;
    int32_t degree;
};

/* OpenCL sign(): +-1, +-0 for +-0, 0 for NaN. */
__device__ __forceinline__ float cl_sign(float x)
{
    return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f));
}

/* HOST = false: the arithmetic of the fused kernel decodeOnceTDMP (above).
 * HOST = true: the reference's host-layered path, Coder::decodeOnceTDMP (MyLdpc.cpp:889-976) over
 * refreshQTDMP / refreshRTDMP / refreshPostPTDMP (decodeCL.c:228-259, 283-290): q_k = P - R_k, the
 * check node of the MS kernel chain (sign = XOR of `< 0`, magnitude = fmin chain from 1000 ==
 * check_ms), P = q_k + R_k; the hard decision is taken once per iteration by layered_hard_kernel. */
template <int D, int V, bool HOST = false>
__global__ __launch_bounds__(kBlock) void layer_kernel(const LayerArgs a)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.y;
    if (tile_finished<V>(a.done, tile)) return;
    const int wave = (int)blockIdx.x * kWavesPerBlock + wave_id_in_block();
    const int r_begin = wave * a.rows_per_wave;
    const int r_end = min(r_begin + a.rows_per_wave, a.n_rows);
    float *Pt = a.P + (size_t)tile * (size_t)a.N * F + (size_t)lane * V;
    float *Rt = a.R + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    uint64_t *hard_t = a.hard + (size_t)tile * (size_t)a.N * V;
    uint64_t frozen[V];
#pragma unroll
    for (int v = 0; v < V; ++v) frozen[v] = a.done[(size_t)tile * V + v];

    for (int r = r_begin; r < r_end; ++r) {
        const int e0 = a.cls_e0[r];
        int col[D];
#pragma unroll
        for (int k = 0; k < D; ++k) col[k] = a.edge_col[e0 + k];
        float p[D][V], m[D][V];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            vload<V>(m[k], Rt + (size_t)(e0 + k) * F);
            vload<V>(p[k], Pt + (size_t)col[k] * F);
        }
        if (HOST) {
#pragma unroll
            for (int k = 0; k < D; ++k)
#pragma unroll
                for (int v = 0; v < V; ++v) p[k][v] = p[k][v] - m[k][v];      /* refreshQTDMP */
            check_ms<D, V>(p, m);                                                 /* refreshRTDMP */
#pragma unroll
            for (int k = 0; k < D; ++k) {
#pragma unroll
                for (int v = 0; v < V; ++v) p[k][v] = p[k][v] + m[k][v];      /* refreshPostPTDMP */
                vstore<V>(Rt + (size_t)(e0 + k) * F, m[k]);
                vstore<V>(Pt + (size_t)col[k] * F, p[k]);
            }
            continue;
        }
#pragma unroll
        for (int v = 0; v < V; ++v) {
            float prod = 1.0f, b = 1000.0f, c = 1001.0f;   /* decodeCL.c:346-348 */
            int bind = -1;
            float sg[D];
#pragma unroll
            for (int k = 0; k < D; ++k) {                  /* :350-367 */
                const float q = p[k][v] - m[k][v];
                sg[k] = cl_sign(q);
                prod *= q;
                p[k][v] = q;
                const float mag = __builtin_fabsf(q);
                if (mag <= b) { c = b; b = mag; bind = k; }
                else if (mag > b && mag <= c) { c = mag; }
            }
            const float sa = cl_sign(prod);                /* :369 */
            const float ab = sa * b, ac = sa * c;
#pragma unroll
            for (int k = 0; k < D; ++k) {                  /* :371-383 */
                const float rn = sg[k] * ((k == bind) ? ac : ab);
                m[k][v] = rn;
                p[k][v] = p[k][v] + rn;
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            vstore<V>(Rt + (size_t)(e0 + k) * F, m[k]);
            vstore<V>(Pt + (size_t)col[k] * F, p[k]);
#pragma unroll
            for (int v = 0; v < V; ++v) {                  /* :388-389, kept current per write */
                const uint64_t w = __ballot(p[k][v] < 0.0f);
                if (lane == 0) {
                    const uint64_t old = hard_t[(size_t)col[k] * V + v];
                    hard_t[(size_t)col[k] * V + v] = (old & frozen[v]) | (w & ~frozen[v]);
                }
            }
        }
    }
}

/* Rows of any degree: same arithmetic with run-time loops, values re-read. */
template <int V>
__global__ __launch_bounds__(kBlock) void layer_kernel_generic(const LayerArgs a)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.y;
    if (tile_finished<V>(a.done, tile)) return;
    const int wave = (int)blockIdx.x * kWavesPerBlock + wave_id_in_block();
    const int r_begin = wave * a.rows_per_wave;
    const int r_end = min(r_begin + a.rows_per_wave, a.n_rows);
    const int D = a.degree;
    float *Pt = a.P + (size_t)tile * (size_t)a.N * F + (size_t)lane * V;
    float *Rt = a.R + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    uint64_t *hard_t = a.hard + (size_t)tile * (size_t)a.N * V;
    uint64_t frozen[V];
#pragma unroll
    for (int v = 0; v < V; ++v) frozen[v] = a.done[(size_t)tile * V + v];

    for (int r = r_begin; r < r_end; ++r) {
        const int e0 = a.cls_e0[r];
        float prod[V], b[V], c[V], sa[V];
        int bind[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { prod[v] = 1.0f; b[v] = 1000.0f; c[v] = 1001.0f; bind[v] = -1; }
        for (int k = 0; k < D; ++k) {
            const int col = a.edge_col[e0 + k];
            float m[V], p[V];
            vload<V>(m, Rt + (size_t)(e0 + k) * F);
            vload<V>(p, Pt + (size_t)col * F);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float q = p[v] - m[v];
                prod[v] *= q;
                p[v] = q;
                const float mag = __builtin_fabsf(q);
                if (mag <= b[v]) { c[v] = b[v]; b[v] = mag; bind[v] = k; }
                else if (mag > b[v] && mag <= c[v]) { c[v] = mag; }
            }
            vstore<V>(Pt + (size_t)col * F, p);        /* q parked in P, as decodeCL.c:357 */
        }
#pragma unroll
        for (int v = 0; v < V; ++v) sa[v] = cl_sign(prod[v]);
        for (int k = 0; k < D; ++k) {
            const int col = a.edge_col[e0 + k];
            float q[V], rn[V];
            vload<V>(q, Pt + (size_t)col * F);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                rn[v] = cl_sign(q[v]) * ((k == bind[v]) ? sa[v] * c[v] : sa[v] * b[v]);
                q[v] = q[v] + rn[v];
            }
            vstore<V>(Rt + (size_t)(e0 + k) * F, rn);
            vstore<V>(Pt + (size_t)col * F, q);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const uint64_t w = __ballot(q[v] < 0.0f);
                if (lane == 0) {
                    const uint64_t old = hard_t[(size_t)col * V + v];
                    hard_t[(size_t)col * V + v] = (old & frozen[v]) | (w & ~frozen[v]);
                }
            }
        }
    }
}

/* hardDecisionTDMP, decodeCL.c:261-280, after the last layer of an iteration: P > 0 -> 0, P < 0 -> 1,
 * otherwise (0, NaN) the bit stays; frozen frames keep theirs.  One wave per column. */
template <int V>
__global__ __launch_bounds__(kBlock) void layered_hard_kernel(const float *__restrict__ P, uint64_t *hard,
                                                              const uint64_t *__restrict__ done, int32_t N)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const int tile = blockIdx.y;
    if (tile_finished<V>(done, tile)) return;
    const int n = (int)blockIdx.x * kWavesPerBlock + wave_id_in_block();
    if (n >= N) return;
    float p[V];
    vload<V>(p, P + ((size_t)tile * N + n) * F + (size_t)lane * V);
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const uint64_t old = hard[((size_t)tile * N + n) * V + v];
        const bool oldb = (old >> lane) & 1ull;
        const bool b = (p[v] > 0.0f) ? false : ((p[v] < 0.0f) ? true : oldb);
        const uint64_t w = __ballot(b);
        const uint64_t frozen = done[(size_t)tile * V + v];
        if (lane == 0) hard[((size_t)tile * N + n) * V + v] = (old & frozen) | (w & ~frozen);
    }
}

struct LayeredInitArgs {
    const float *__restrict__ llr;  /* [frames][N] */
    float *__restrict__ P;          /* [T][N][F] */
    uint64_t *__restrict__ hard;    /* [T][N][V] */
    int64_t frames;
    int32_t N;
    int32_t zero_bits;              /* host-layered path: bits start at 0 (the reference: undefined) */
};

/* lP = postCode (decodeCL.c:331-334), transposed into the tile layout; bits = y < 0. */
template <int V>
__global__ __launch_bounds__(kBlock) void layered_init_kernel(const LayeredInitArgs a)
{
    constexpr int F = 64 * V;
    constexpr int LD = F + 1;
    __shared__ float patch[kInitCols * LD];
    const int tile = blockIdx.y;
    const int n0 = blockIdx.x * kInitCols;
    {
        const int c = threadIdx.x & (kInitCols - 1);
        const int n = n0 + c;
        for (int f = threadIdx.x / kInitCols; f < F; f += kBlock / kInitCols) {
            const int64_t frame = (int64_t)tile * F + f;
            float y = 1.0f;
            if (frame < a.frames && n < a.N) y = a.llr[(size_t)frame * a.N + n];
            patch[c * LD + f] = y;
        }
    }
    __syncthreads();
    const int lane = threadIdx.x & 63;
    for (int c = threadIdx.x >> 6; c < kInitCols; c += kWavesPerBlock) {
        const int n = n0 + c;
        if (n >= a.N) break;
        float y[V];
#pragma unroll
        for (int v = 0; v < V; ++v) y[v] = patch[c * LD + lane * V + v];
        vstore<V>(a.P + ((size_t)tile * a.N + n) * F + (size_t)lane * V, y);
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const uint64_t w = __ballot(y[v] < 0.0f);
            if (lane == 0) a.hard[((size_t)tile * a.N + n) * V + v] = a.zero_bits ? 0ull : w;
        }
    }
}

/* ------------------------------------------------------------- host side */

constexpr int kMaxUnrolledLayerDegree = 24;

struct LayerGroup {
    int layer = 0, degree = 0, count = 0;
    int32_t *e0 = nullptr; /* device */
};

struct LayeredPlan {
    int32_t M = 0, N = 0, layer_rows = 0;
    int64_t E = 0;
    int T = 0, V = 1;
    int host_arith = 0;             /* 1: LDPC_ALGO_LAYERED_HOST (layer_kernel<.., HOST = true>) */
    float *P = nullptr, *R = nullptr;
    std::vector<LayerGroup> groups; /* ordered by layer */
};

inline void layered_plan_destroy(LayeredPlan *pl)
{
    for (auto &g : pl->groups)
        if (g.e0) (void)hipFree(g.e0);
    pl->groups.clear();
    if (pl->P) (void)hipFree(pl->P);
    if (pl->R) (void)hipFree(pl->R);
    pl->P = pl->R = nullptr;
}

/* returns 0, -1 for an invalid layering, -2 for a HIP failure */
inline int layered_plan_create(LayeredPlan *pl, int32_t M, int32_t N, int64_t E,
                               const std::vector<int32_t> &row_ptr, const std::vector<int32_t> &cols,
                               int32_t layer_rows, int T, int V)
{
    if (layer_rows <= 0 || M % layer_rows) return -1;
    pl->M = M; pl->N = N; pl->E = E; pl->layer_rows = layer_rows; pl->T = T; pl->V = V;
    std::vector<int32_t> seen((size_t)N, -1);
    const int layers = M / layer_rows;
    for (int l = 0; l < layers; ++l) {
        std::map<int, std::vector<int32_t>> by_deg;
        for (int32_t m = l * layer_rows; m < (l + 1) * layer_rows; ++m) {
            for (int32_t p = row_ptr[m]; p < row_ptr[m + 1]; ++p) {
                if (seen[cols[p]] == l) return -1;  /* two rows of one layer share a column */
                seen[cols[p]] = l;
            }
            const int deg = row_ptr[m + 1] - row_ptr[m];
            if (deg > 0) by_deg[deg].push_back(row_ptr[m]);
        }
        for (auto &kv : by_deg) {
            LayerGroup g;
            g.layer = l; g.degree = kv.first; g.count = (int)kv.second.size();
            if (hipMalloc((void **)&g.e0, kv.second.size() * sizeof(int32_t)) != hipSuccess) return -2;
            pl->groups.push_back(g);
            if (hipMemcpy(g.e0, kv.second.data(), kv.second.size() * sizeof(int32_t),
                          hipMemcpyHostToDevice) != hipSuccess) return -2;
        }
    }
    const size_t TF = (size_t)T * 64 * V;
    if (hipMalloc((void **)&pl->P, TF * N * sizeof(float)) != hipSuccess) return -2;
    if (hipMalloc((void **)&pl->R, TF * (size_t)E * sizeof(float)) != hipSuccess) return -2;
    return 0;
}

struct LayeredRun {
    /* optional per-launch timing hooks (HIP events on the decode stream) */
    hipError_t (*span_begin)(void *ctx, hipStream_t s, int kind, int degree, int64_t bytes) = nullptr;
    hipError_t (*span_end)(void *ctx, hipStream_t s) = nullptr;
    void *span_ctx = nullptr;
    const float *llr_dev;
    int64_t frames;
    uint8_t *out_dev;
    int64_t out_bytes;
    int32_t *iters_dev;
    int32_t K, max_iter, tap_iter, early_term, pack_mode;
    uint64_t *hard, *failw, *done;
    int32_t *iters;
    const int32_t *row_ptr, *edge_col;
    int32_t *summary;
};

/* The launching code instantiates every kernel above: only the translation unit that defines
 * LDPC_ENGINE_LAYERED (engine_layered.hip) compiles it; the host driver calls engine_layered_run. */
#ifdef LDPC_ENGINE_LAYERED
using LayerFn = void (*)(const LayerArgs);
template <int V, int D, bool HOST = false> struct LayerTable {
    static void fill(LayerFn *t) { t[D] = layer_kernel<D, V, HOST>; LayerTable<V, D - 1, HOST>::fill(t); }
};
template <int V, bool HOST> struct LayerTable<V, 0, HOST> {
    static void fill(LayerFn *t) { t[0] = HOST ? nullptr : layer_kernel_generic<V>; }
};


template <int V> __global__ void layered_summary_kernel(const int32_t *iters, const uint64_t *done,
                                                        int64_t frames, int32_t *summary)
{
    constexpr int F = 64 * V;
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    const int64_t tile = f / F;
    const int fi = (int)(f % F);
    atomicMax(&summary[0], iters[f]);
    if ((done[tile * V + fi % V] >> (fi / V)) & 1ull) atomicAdd(&summary[1], 1);
}

template <int V>
inline hipError_t layered_run_v(LayeredPlan *pl, const LayeredRun &r, hipStream_t s, int32_t *launched)
{
    constexpr int F = 64 * V;
    const int tiles = (int)((r.frames + F - 1) / F);
    const size_t slot = (size_t)pl->T * V;
    LayerFn table[kMaxUnrolledLayerDegree + 1];
    if (pl->host_arith) LayerTable<V, kMaxUnrolledLayerDegree, true>::fill(table);
    else LayerTable<V, kMaxUnrolledLayerDegree>::fill(table);
    const int rounds = r.tap_iter ? (r.tap_iter < r.max_iter ? r.tap_iter : r.max_iter) : r.max_iter;
    hipError_t e;
    if ((e = hipMemsetAsync(r.failw, 0, (size_t)(r.max_iter + 2) * slot * sizeof(uint64_t), s))) return e;
    if ((e = hipMemsetAsync(r.summary, 0, 2 * sizeof(int32_t), s))) return e;
    if ((e = hipMemsetAsync(pl->R, 0, (size_t)tiles * F * (size_t)pl->E * sizeof(float), s))) return e;
    {
        LayeredInitArgs ia{r.llr_dev, pl->P, r.hard, r.frames, pl->N, pl->host_arith};
        dim3 grid((pl->N + kInitCols - 1) / kInitCols, tiles);
        layered_init_kernel<V><<<grid, kBlock, 0, s>>>(ia);
        StateArgs st{r.done, nullptr, r.iters, nullptr, r.frames, 0, r.max_iter, 1};
        state_kernel<V><<<tiles, 64, 0, s>>>(st);
    }
    int it = 0;
    for (it = 1; it <= rounds; ++it) {
        for (auto &g : pl->groups) {
            LayerArgs a{pl->P, pl->R, g.e0, r.edge_col, r.hard, r.done, pl->E, pl->N, g.count,
                        4, g.degree};
            const int waves = (g.count + a.rows_per_wave - 1) / a.rows_per_wave;
            dim3 grid((waves + kWavesPerBlock - 1) / kWavesPerBlock, tiles);
            const int k = g.degree <= kMaxUnrolledLayerDegree ? g.degree : 0;
            if (!table[k]) return hipErrorInvalidValue;      /* host arithmetic: unrolled degrees only */
            if (r.span_begin && (e = r.span_begin(r.span_ctx, s, 2, g.degree,
                                                  (int64_t)16 * g.degree * g.count * r.frames))) return e;
            table[k]<<<grid, kBlock, 0, s>>>(a);
            if (r.span_end && (e = r.span_end(r.span_ctx, s))) return e;
        }
        if (pl->host_arith) {
            dim3 hgrid((pl->N + kWavesPerBlock - 1) / kWavesPerBlock, tiles);
            layered_hard_kernel<V><<<hgrid, kBlock, 0, s>>>(pl->P, r.hard, r.done, pl->N);
        }
        /* syndrome of this round's bits, freeze (decodeCL.c:393-410) */
        if (!r.early_term && it != rounds) continue;
        uint64_t *fw = r.failw + (size_t)it * slot;
        SyndromeArgs sa{r.row_ptr, r.edge_col, r.hard, fw, r.done, pl->M, pl->N, 0, 0};
        dim3 sgrid((pl->M + kBlock - 1) / kBlock, tiles);
        syndrome_kernel<V><<<sgrid, kBlock, 0, s>>>(sa);
        StateArgs st{r.done, fw, r.iters, nullptr, r.frames, it, r.max_iter, 1};
        state_kernel<V><<<tiles, 64, 0, s>>>(st);
    }
    *launched = rounds;
    PackArgs pa{r.hard, r.out_dev, r.iters, r.iters_dev, r.frames, r.out_bytes, pl->N, r.K, r.pack_mode};
    if (r.out_dev) {
        if (r.pack_mode == 0) {
            pack_kernel<V><<<pack_grid<V>(r.K, tiles), kBlock, 0, s>>>(pa);
        } else {
            const int64_t n = r.out_bytes > r.frames ? r.out_bytes : r.frames;
            pack_kernel<V><<<(unsigned)((n + kBlock - 1) / kBlock), kBlock, 0, s>>>(pa);
        }
    }
    layered_summary_kernel<V><<<(unsigned)((r.frames + 255) / 256), 256, 0, s>>>(r.iters, r.done, r.frames,
                                                                                 r.summary);
    return hipGetLastError();
}

inline hipError_t layered_run(LayeredPlan *pl, const LayeredRun &r, hipStream_t s, int32_t *launched)
{
    if (pl->V == 1) return layered_run_v<1>(pl, r, s, launched);
    if (pl->V == 2) return layered_run_v<2>(pl, r, s, launched);
    return layered_run_v<4>(pl, r, s, launched);
}

#endif  /* LDPC_ENGINE_LAYERED */

/* which: 0 = R [frame][E], 2 = P [frame][N], 3 = bits [frame][N] */
inline hipError_t layered_dump(LayeredPlan *pl, int which, float *host_out, int64_t count, int64_t frames,
                               const uint64_t *hard_dev, const int32_t *)
{
    const int V = pl->V, F = 64 * V;
    const int tiles = (int)((frames + F - 1) / F);
    if (which == 0 || which == 2) {
        const int64_t per = which == 0 ? pl->E : pl->N;
        if (count != frames * per) return hipErrorInvalidValue;
        const float *src = which == 0 ? pl->R : pl->P;
        std::vector<float> tile((size_t)per * F);
        for (int t = 0; t < tiles; ++t) {
            hipError_t e = hipMemcpy(tile.data(), src + (size_t)t * per * F, tile.size() * sizeof(float),
                                     hipMemcpyDeviceToHost);
            if (e != hipSuccess) return e;
            for (int fi = 0; fi < F; ++fi) {
                const int64_t f = (int64_t)t * F + fi;
                if (f >= frames) break;
                for (int64_t i = 0; i < per; ++i) host_out[f * per + i] = tile[(size_t)i * F + fi];
            }
        }
        return hipSuccess;
    }
    if (which == 3) {
        if (count != frames * pl->N) return hipErrorInvalidValue;
        std::vector<uint64_t> w((size_t)tiles * pl->N * V);
        hipError_t e = hipMemcpy(w.data(), hard_dev, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost);
        if (e != hipSuccess) return e;
        for (int64_t f = 0; f < frames; ++f) {
            const int64_t t = f / F;
            const int fi = (int)(f % F);
            for (int32_t n = 0; n < pl->N; ++n)
                host_out[f * pl->N + n] = (float)((w[((size_t)t * pl->N + n) * V + fi % V] >> (fi / V)) & 1ull);
        }
        return hipSuccess;
    }
    return hipErrorInvalidValue;
}

}  // namespace ldpc
