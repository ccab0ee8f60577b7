/*
 * flood_kernels.hpp -- gfx950 kernels for two-phase (flooding) LDPC belief
 * propagation: sum-product in the probability domain and min-sum, fp32.
 *
 * Replaces the reference's kernel chain (decodeCL.c):
 *   check_kernel<SP>  <- refreshR        decodeCL.c:25-41
 *   var_kernel<SP>    <- hardDecision    decodeCL.c:64-86  + refreshQ :43-62
 *   check_kernel<MS>  <- refreshRMS      decodeCL.c:126-147
 *   syndrome_kernel   <- checkResult     decodeCL.c:88-108
 *   var_kernel<MS>    <- refreshPostPMS  decodeCL.c:149-171 + refreshQMS :175-186
 *   init_kernel       <- decodeInit :3-22 / decodeInitMS :113-124
 *   pack_kernel       <- toChar :188-199 (and decodeCPU's bit packing MyLdpc.cpp:765-774)
 *
 * Design (MI355X-first, not the reference's layout):
 *   - The reference indexes messages [frame][edge] with the frame as NDRange
 *     dim 0, so adjacent work-items are E floats apart, and every edge re-walks
 *     its whole row/column (O(d^2) reads).  Here FRAMES ARE THE LANES: a tile is
 *     F = 64*V frames (V = 1, 2 or 4 frames per lane) and every per-edge message
 *     of a tile is one contiguous F*4-byte segment, msg[tile][edge][F].  A
 *     wave-instruction moves one whole segment (256 B .. 1 KiB, 16 B per lane at
 *     V = 4), H's indices are wave-uniform (scalar loads, SGPRs), each message is
 *     read once and written once per half-iteration, and the check / variable
 *     reductions are per-lane register chains -- no cross-lane traffic, no LDS,
 *     no MFMA (sparse gather/reduce, HBM-bound).
 *   - Check phase: a row's edges are consecutive edge ids, so it streams.
 *     Variable phase: a column's edges are gathered/scattered as whole segments.
 *   - fp32 results are bit-identical to the reference's operation order:
 *     products/sums run left to right in ascending edge id with the own edge
 *     skipped (prefix shared, tail recomputed), IEEE division, no FMA
 *     contraction (compile with -ffp-contract=off).
 *   - SP messages are stored as ONE float per edge and direction: the reference
 *     only ever consumes q0-q1 (decodeCL.c:37) and r0 = (1+d)/2, r1 = (1-d)/2 are
 *     exact functions of d (:39-40), so storing d_q = fl(q0-q1) and d_r = d loses
 *     nothing: 16*E + 4*N bytes per frame-iteration.
 *   - Hard bits live as bit masks hard[tile][n][V] (bit l of word v = frame
 *     V*l+v): the row syndrome is an XOR of wave-uniform 64-bit words.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ldpc_expf.h"

namespace ldpc {

constexpr int kAlgoSP = 0;
constexpr int kAlgoMS = 1;
constexpr int kCompactCapacity = 512;   /* frames a child decoder takes over (8 tiles of 64) */
constexpr int kLastCapacity = 64;       /* ... and the last decoder of the chain: one tile */
constexpr int kBlock = 256;          /* 4 waves */
constexpr int kWavesPerBlock = 4;
constexpr int kMaxUnrolledDegree = 16;       /* variable-node kernels, sum-product check kernels */
constexpr int kMaxUnrolledCheckDegreeMS = 32; /* min-sum check rows: O(D) registers in narrow waves */

/* ---- V-wide per-lane vectors --------------------------------------------- */
/* Message streams are touched once per kernel (7.4 GB per launch at B = 4096):
 * LDPC_NT_LOAD / LDPC_NT_STORE = 1 mark them non-temporal (measured on MI355X: both on
 * -4.3 % step time, loads only +2.8 %, stores only -1.6 %). */
#ifndef LDPC_NT_LOAD
#define LDPC_NT_LOAD 1
#endif
#ifndef LDPC_NT_STORE
#define LDPC_NT_STORE 1
#endif
typedef float vf2 __attribute__((ext_vector_type(2)));
typedef float vf4 __attribute__((ext_vector_type(4)));

template <typename T> __device__ __forceinline__ T ld_stream(const T *p)
{
#if LDPC_NT_LOAD
    return __builtin_nontemporal_load(p);
#else
    return *p;
#endif
}
template <typename T> __device__ __forceinline__ void st_stream(T *p, T v)
{
#if LDPC_NT_STORE
    __builtin_nontemporal_store(v, p);
#else
    *p = v;
#endif
}

template <int V> __device__ __forceinline__ void vload(float (&d)[V], const float *p);
template <> __device__ __forceinline__ void vload<1>(float (&d)[1], const float *p) { d[0] = ld_stream(p); }
template <> __device__ __forceinline__ void vload<2>(float (&d)[2], const float *p)
{
    const vf2 t = ld_stream(reinterpret_cast<const vf2 *>(p));
    d[0] = t.x; d[1] = t.y;
}
template <> __device__ __forceinline__ void vload<4>(float (&d)[4], const float *p)
{
    const vf4 t = ld_stream(reinterpret_cast<const vf4 *>(p));
    d[0] = t.x; d[1] = t.y; d[2] = t.z; d[3] = t.w;
}
template <int V> __device__ __forceinline__ void vstore(float *p, const float (&s)[V]);
template <> __device__ __forceinline__ void vstore<1>(float *p, const float (&s)[1]) { st_stream(p, s[0]); }
template <> __device__ __forceinline__ void vstore<2>(float *p, const float (&s)[2])
{
    vf2 t; t.x = s[0]; t.y = s[1];
    st_stream(reinterpret_cast<vf2 *>(p), t);
}
template <> __device__ __forceinline__ void vstore<4>(float *p, const float (&s)[4])
{
    vf4 t; t.x = s[0]; t.y = s[1]; t.z = s[2]; t.w = s[3];
    st_stream(reinterpret_cast<vf4 *>(p), t);
}

/* fp16 message storage (msg_dtype = LDPC_MSG_F16): 2 bytes per message, arithmetic in fp32.
 * Loads widen exactly; stores round to nearest even. */
typedef _Float16 hf;
typedef _Float16 vh2 __attribute__((ext_vector_type(2)));
typedef _Float16 vh4 __attribute__((ext_vector_type(4)));

template <int V> __device__ __forceinline__ void vload(float (&d)[V], const hf *p);
template <> __device__ __forceinline__ void vload<1>(float (&d)[1], const hf *p) { d[0] = (float)ld_stream(p); }
template <> __device__ __forceinline__ void vload<2>(float (&d)[2], const hf *p)
{
    const vh2 t = ld_stream(reinterpret_cast<const vh2 *>(p));
    d[0] = (float)t.x; d[1] = (float)t.y;
}
template <> __device__ __forceinline__ void vload<4>(float (&d)[4], const hf *p)
{
    const vh4 t = ld_stream(reinterpret_cast<const vh4 *>(p));
    d[0] = (float)t.x; d[1] = (float)t.y; d[2] = (float)t.z; d[3] = (float)t.w;
}
template <int V> __device__ __forceinline__ void vstore(hf *p, const float (&s)[V]);
template <> __device__ __forceinline__ void vstore<1>(hf *p, const float (&s)[1]) { st_stream(p, (hf)s[0]); }
template <> __device__ __forceinline__ void vstore<2>(hf *p, const float (&s)[2])
{
    vh2 t; t.x = (hf)s[0]; t.y = (hf)s[1];
    st_stream(reinterpret_cast<vh2 *>(p), t);
}
template <> __device__ __forceinline__ void vstore<4>(hf *p, const float (&s)[4])
{
    vh4 t; t.x = (hf)s[0]; t.y = (hf)s[1]; t.z = (hf)s[2]; t.w = (hf)s[3];
    st_stream(reinterpret_cast<vh4 *>(p), t);
}

__device__ __forceinline__ int wave_id_in_block()
{
    return __builtin_amdgcn_readfirstlane((int)(threadIdx.x >> 6));
}

template <int V> __device__ __forceinline__ bool tile_finished(const uint64_t *done, int tile)
{
    bool all = true;
#pragma unroll
    for (int v = 0; v < V; ++v) all = all && (done[(size_t)tile * V + v] == ~0ull);
    return all;
}

/* ---- device-side tail (early termination without host polling) --------------------------------
 * An asynchronous caller has all max_iter rounds enqueued up front.  When only a few frames of a
 * large batch are still running, tail_gather_kernel (below) moves their state into `n_tiles`
 * OVERFLOW tiles that follow the batch's own tiles in every array and flips state[0]; from then on
 * the blocks of the first n_tiles grid rows of every launch work on the overflow tiles and all
 * other blocks leave at once -- the decision is taken on the device, no launch is added per round
 * except the (empty) gather attempt.  state == nullptr: feature off. */
struct TailRef {
    const int32_t *state;   /* [0] handed over, [1] frames handed over, [2] round of the hand-over */
    int32_t base_tile;      /* index of the first overflow tile = tiles of max_batch */
    int32_t n_tiles;        /* overflow tiles */
};

/* The tile a block with grid row `row` works on, or -1 if it has nothing to do. */
template <int V> __device__ __forceinline__ int tile_select(const TailRef &t, const uint64_t *done, int row)
{
    int tile = row;
    if (t.state && __builtin_amdgcn_readfirstlane(t.state[0])) {
        if (row >= t.n_tiles) return -1;
        tile = t.base_tile + row;
    }
    return tile_finished<V>(done, tile) ? -1 : tile;
}

/* ========================================================================= */
/*                               check node                                   */
/* ========================================================================= */

struct CheckArgs {
    const void *__restrict__ Q;          /* [T][E][F] variable->check (float or fp16) */
    void *__restrict__ R;                /* [T][E][F] check->variable          */
    const int32_t *__restrict__ cls_e0;  /* [n_rows] first edge id of each row of this degree class */
    const uint64_t *__restrict__ done;   /* [T][V] frozen frames                */
    int64_t E;
    int32_t n_rows;
    int32_t rows_per_wave;
    int32_t degree;                      /* generic kernel only */
    TailRef tail;
    int32_t tiles_first = 0;             /* grid is (tiles, blocks) instead of (blocks, tiles): grid_pos() */
    /* min-sum, round 1 only: the variable->check messages of round 0 are the channel values (decodeInitMS,
     * decodeCL.c:121: q = y), so they are read from the channel array by column instead of from Q, which the input
     * transpose then does not have to write at all (a third of its traffic).  nullptr: read Q. */
    const void *__restrict__ first_chan = nullptr;     /* [T][N][F] */
    const int32_t *__restrict__ edge_col = nullptr;    /* [E] column of every edge */
    int32_t N = 0;
    /* Where the variable->check message of edge e lives in Q: slot qpos[e] (nullptr: slot e).  Q is stored in the
     * order its WRITERS produce it -- the variable-node kernels column by column, then the column-fused check
     * kernel's own edges row by row -- so that every store of a round streams; the check kernels, which read Q,
     * gather instead (tools/gather_probe.hip: random 1-KiB reads cost nothing, random writes 9-15 %). */
    const int32_t *__restrict__ qpos = nullptr;        /* [E]; never null in a launch (the identity where Q is in edge order) */
};

/* Q slots (CheckArgs::qpos) of the D edges e0 ... of a row: ONE load by the wave's first D lanes and a broadcast each.
 * The values are wave-uniform, so a message's address is a scalar row address plus the lane's 32-bit offset (the
 * slots of a row are unrelated: one 64-bit vector address each would cost two registers per message in flight). */
template <int D> __device__ __forceinline__ void row_slots(const int32_t *__restrict__ qpos, int e0, int lane, int (&slot)[D])
{
    static_assert(D <= 64, "one lane per edge");
    const int mine = qpos[e0 + (lane < D ? lane : D - 1)];
#pragma unroll
    for (int k = 0; k < D; ++k) slot[k] = __builtin_amdgcn_readlane(mine, k);
}
/* slot of one edge */
__device__ __forceinline__ int edge_slot(const int32_t *__restrict__ qpos, int e)
{
    return __builtin_amdgcn_readfirstlane(qpos[e]);
}

/* Where a block stands in the launch.  Blocks are dispatched with blockIdx.x varying fastest: a grid of
 * (tiles, blocks) therefore has the blocks in flight at any moment spread over ALL tiles of the batch --
 * sixteen regions 58 MB apart instead of one moving window -- which the memory system rewards (the
 * column-fused check kernel: 1.39 -> 1.27 ms, profiles/r02_ab_tile_fastest_grid.txt).  The host asks for it
 * whenever the block count fits gridDim.y. */
struct GridPos { int row, block; };
__device__ __forceinline__ GridPos grid_pos(int tiles_first)
{
    return tiles_first ? GridPos{(int)blockIdx.x, (int)blockIdx.y} : GridPos{(int)blockIdx.y, (int)blockIdx.x};
}

/* Sum-product, decodeCL.c:32-40: out_k = prod_{j != k} x_j, multiplied left to
 * right in ascending j starting from 1.0f (1*x is exact, so the chain starts at
 * the first factor).  The prefix x_0..x_{k-1} is shared between outputs. */
template <int D, int V>
__device__ __forceinline__ void check_sp(const float (&x)[D][V], float (&out)[D][V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) {
        if (D == 1) { out[0][v] = 1.0f; continue; }
        {
            float p = x[1][v];
#pragma unroll
            for (int j = 2; j < D; ++j) p *= x[j][v];
            out[0][v] = p;
        }
        float pre = x[0][v];
#pragma unroll
        for (int k = 1; k < D; ++k) {
            float p = pre;
#pragma unroll
            for (int j = k + 1; j < D; ++j) p *= x[j][v];
            out[k][v] = p;
            pre *= x[k][v];
        }
    }
}

/* Min-sum, decodeCL.c:132-146: sign = XOR of (x_j < 0) over j != k, magnitude =
 * fmin chain over |x_j| starting at 1000.  min is exact and order-free, so the
 * two-smallest form gives the same floats; NaNs are skipped as fmin skips them. */
template <int D, int V>
__device__ __forceinline__ void check_ms(const float (&x)[D][V], float (&out)[D][V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) {
        float m1 = 1000.0f, m2 = 1000.0f;
        int idx = -1;
        unsigned par = 0;
#pragma unroll
        for (int j = 0; j < D; ++j) {
            const float a = __builtin_fabsf(x[j][v]);
            par ^= (x[j][v] < 0.0f) ? 1u : 0u;
            if (a < m1) { m2 = m1; m1 = a; idx = j; }
            else if (a < m2) { m2 = a; }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            const float b = (k == idx) ? m2 : m1;
            const unsigned s = par ^ ((x[k][v] < 0.0f) ? 1u : 0u);
            out[k][v] = s ? -b : b;
        }
    }
}

/* V = frames per lane of the LAYOUT (tile = 64*V frames); W <= V = floats per lane this
 * kernel moves: a row's 64*V-float segment is covered by V/W waves of 64*W floats each
 * (consecutive waves of a block).  Narrow waves (W = 1) need fewer registers and run at higher
 * occupancy: measured 6.0 TB/s against 5.5 TB/s for W = V = 4 on the degree-7 rows. */
template <int ALGO, int D, int V, int W, typename T>
__global__ __launch_bounds__(kBlock) void check_kernel(const CheckArgs a)
{
    constexpr size_t F = 64 * V;
    constexpr int SUB = V / W;
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    const int wave = gp.block * kWavesPerBlock + wave_id_in_block();
    const int sub = wave % SUB;
    const int r_begin = (wave / SUB) * a.rows_per_wave;
    const int r_end = min(r_begin + a.rows_per_wave, a.n_rows);
    const unsigned lane_off = (unsigned)sub * 64 * W + (unsigned)lane * W;
    const T *Qu = static_cast<const T *>(a.Q) + (size_t)tile * (size_t)a.E * F;      /* wave-uniform: + slot * F + lane_off */
    T *Rt = static_cast<T *>(a.R) + (size_t)tile * (size_t)a.E * F + lane_off;

    const T *Ct = (ALGO == kAlgoMS && a.first_chan)
                      ? static_cast<const T *>(a.first_chan) + (size_t)tile * (size_t)a.N * F + lane_off : nullptr;
    for (int r = r_begin; r < r_end; ++r) {
        const int e0 = a.cls_e0[r];
        float x[D][W], out[D][W];
        if (ALGO == kAlgoMS && Ct) {
#pragma unroll
            for (int k = 0; k < D; ++k) vload<W>(x[k], Ct + (size_t)a.edge_col[e0 + k] * F);
        } else {
            int sl[D];
            row_slots<D>(a.qpos, e0, lane, sl);
#pragma unroll
            for (int k = 0; k < D; ++k) vload<W>(x[k], Qu + (size_t)sl[k] * F + lane_off);
        }
        if (ALGO == kAlgoSP) check_sp<D, W>(x, out); else check_ms<D, W>(x, out);
#pragma unroll
        for (int k = 0; k < D; ++k) vstore<W>(Rt + (size_t)(e0 + k) * F, out[k]);
    }
}

/* One row of any degree with run-time loops (rows wider than the unrolled kernels, and the few
 * left-over rows a linked check launch takes along).  Min-sum: two passes over the row.  Sum-product:
 * the exact left-to-right product needs one pass per output, as the reference does (L1/L2 serve the
 * repeats).  Qt / Rt: this lane's V values of the tile. */
template <int ALGO, int V, typename T>
__device__ __forceinline__ void check_row_generic(const T *Qu, unsigned lane_off, T *Rt, int e0, int D,
                                                  const int32_t *__restrict__ qpos)
{
    constexpr size_t F = 64 * V;
    if (ALGO == kAlgoMS) {
        /* two passes over the row instead of one per output: min1/min2/argmin and the sign
         * parity first, then each output from its own re-read value (L2 serves the re-read) */
        float m1[V], m2[V];
        int idx[V];
        unsigned par[V];
#pragma unroll
        for (int v = 0; v < V; ++v) { m1[v] = 1000.0f; m2[v] = 1000.0f; idx[v] = -1; par[v] = 0; }
        for (int j = 0; j < D; ++j) {
            float xj[V];
            vload<V>(xj, Qu + (size_t)edge_slot(qpos, e0 + j) * F + lane_off);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float m = __builtin_fabsf(xj[v]);
                par[v] ^= (xj[v] < 0.0f) ? 1u : 0u;
                if (m < m1[v]) { m2[v] = m1[v]; m1[v] = m; idx[v] = j; }
                else if (m < m2[v]) { m2[v] = m; }
            }
        }
        for (int k = 0; k < D; ++k) {
            float xk[V], o[V];
            vload<V>(xk, Qu + (size_t)edge_slot(qpos, e0 + k) * F + lane_off);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float b = (k == idx[v]) ? m2[v] : m1[v];
                const unsigned sg = par[v] ^ ((xk[v] < 0.0f) ? 1u : 0u);
                o[v] = sg ? -b : b;
            }
            vstore<V>(Rt + (size_t)(e0 + k) * F, o);
        }
        return;
    }
    for (int k = 0; k < D; ++k) {
        float o[V];
        float p[V];
#pragma unroll
        for (int v = 0; v < V; ++v) p[v] = 1.0f;
        for (int j = 0; j < D; ++j) {
            if (j == k) continue;
            float xj[V];
            vload<V>(xj, Qu + (size_t)edge_slot(qpos, e0 + j) * F + lane_off);
#pragma unroll
            for (int v = 0; v < V; ++v) p[v] *= xj[v];
        }
#pragma unroll
        for (int v = 0; v < V; ++v) o[v] = p[v];
        vstore<V>(Rt + (size_t)(e0 + k) * F, o);
    }
}

/* Any degree; only launched above the unrolled degrees. */
template <int ALGO, int V, typename T>
__global__ __launch_bounds__(kBlock) void check_kernel_generic(const CheckArgs a)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    const int wave = gp.block * kWavesPerBlock + wave_id_in_block();
    const int r_begin = wave * a.rows_per_wave;
    const int r_end = min(r_begin + a.rows_per_wave, a.n_rows);
    const T *Qu = static_cast<const T *>(a.Q) + (size_t)tile * (size_t)a.E * F;
    T *Rt = static_cast<T *>(a.R) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    for (int r = r_begin; r < r_end; ++r) check_row_generic<ALGO, V, T>(Qu, (unsigned)lane * V, Rt, a.cls_e0[r], a.degree, a.qpos);
}

/* ---- several degree classes in ONE launch -------------------------------------------------
 * A round used to cost one launch per row / column degree class (DVB-S2 rate 1/2: 2 + 4, two of
 * them a single row / column).  A group launch covers every class of a degree bucket: the table
 * says which block range belongs to which class, the block's (wave-uniform) degree picks the
 * unrolled body.  Same bodies, same operation order, same bits; registers are those of the
 * bucket's widest degree, hence the buckets. */
struct GroupClass {
    int32_t block_begin;        /* first blockIdx.x of this class */
    int32_t degree;
    int32_t count;              /* rows / columns */
    int32_t pad;
    const int32_t *ids;         /* rows: first edge ids; columns: column ids */
    const int32_t *edges;       /* columns: [count][degree] edge ids */
    int64_t q_base;             /* columns: first Q slot of the class's D streams (VarArgs::q_base), -1: slot = edge id */
};

template <int ALGO, int D, int V, int W, typename T>
__device__ __forceinline__ void check_rows(const T *Qu, unsigned lane_off, T *Rt, const int32_t *__restrict__ e0s, int r_begin, int r_end,
                                           const T *Ct, const int32_t *__restrict__ edge_col,
                                           const int32_t *__restrict__ qpos)
{
    const int lane = threadIdx.x & 63;
    constexpr size_t F = 64 * V;
    for (int r = r_begin; r < r_end; ++r) {
        const int e0 = e0s[r];
        float x[D][W], out[D][W];
        if (ALGO == kAlgoMS && Ct) {            /* round 1: q = y, by column (CheckArgs::first_chan) */
#pragma unroll
            for (int k = 0; k < D; ++k) vload<W>(x[k], Ct + (size_t)edge_col[e0 + k] * F);
        } else {
            int sl[D];
            row_slots<D>(qpos, e0, lane, sl);
#pragma unroll
            for (int k = 0; k < D; ++k) vload<W>(x[k], Qu + (size_t)sl[k] * F + lane_off);
        }
        if (ALGO == kAlgoSP) check_sp<D, W>(x, out); else check_ms<D, W>(x, out);
#pragma unroll
        for (int k = 0; k < D; ++k) vstore<W>(Rt + (size_t)(e0 + k) * F, out[k]);
    }
}

template <int ALGO, int V, typename T, int D, int DLO, int W = 1> struct CheckDispatch {
    static __device__ __forceinline__ void run(int deg, const T *Qu, unsigned lane_off, T *Rt, const int32_t *e0s, int rb, int re,
                                               const T *Ct, const int32_t *edge_col, const int32_t *qpos)
    {
        if (deg == D) check_rows<ALGO, D, V, W, T>(Qu, lane_off, Rt, e0s, rb, re, Ct, edge_col, qpos);
        else CheckDispatch<ALGO, V, T, D - 1, DLO, W>::run(deg, Qu, lane_off, Rt, e0s, rb, re, Ct, edge_col, qpos);
    }
};
template <int ALGO, int V, typename T, int DLO, int W> struct CheckDispatch<ALGO, V, T, DLO, DLO, W> {
    static __device__ __forceinline__ void run(int, const T *Qu, unsigned lane_off, T *Rt, const int32_t *e0s, int rb, int re,
                                               const T *Ct, const int32_t *edge_col, const int32_t *qpos)
    {
        check_rows<ALGO, DLO, V, W, T>(Qu, lane_off, Rt, e0s, rb, re, Ct, edge_col, qpos);
    }
};

/* degrees DLO..DHI; W values per lane (V / W consecutive waves cover a row's segment).  W = 1: narrow waves,
 * the form of all fp32 kernels (256 B per wave-instruction).  fp16 messages at V = 4 use W = 2: with one
 * 2-byte value per lane a wave-instruction moves 128 B and the degree-30 rows of the rate-9/10 code ran at
 * 5.0 TB/s against 6.0 for the fp32 narrow form. */
template <int ALGO, int V, typename T, int DLO, int DHI, int W = 1>
__global__ __launch_bounds__(kBlock) void check_group_kernel(const CheckArgs a, const GroupClass *__restrict__ cls, int n_classes)
{
    constexpr size_t F = 64 * V;
    constexpr int SUB = V / W;
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    int c = 0;
    while (c + 1 < n_classes && gp.block >= cls[c + 1].block_begin) ++c;
    const int wave = (gp.block - cls[c].block_begin) * kWavesPerBlock + wave_id_in_block();
    const int sub = wave % SUB;
    const int r_begin = (wave / SUB) * a.rows_per_wave;
    const int r_end = min(r_begin + a.rows_per_wave, cls[c].count);
    const unsigned lane_off = (unsigned)sub * 64 * W + (unsigned)lane * W;
    const T *Qu = static_cast<const T *>(a.Q) + (size_t)tile * (size_t)a.E * F;
    T *Rt = static_cast<T *>(a.R) + (size_t)tile * (size_t)a.E * F + lane_off;
    const T *Ct = (ALGO == kAlgoMS && a.first_chan)
                      ? static_cast<const T *>(a.first_chan) + (size_t)tile * (size_t)a.N * F + lane_off : nullptr;
    CheckDispatch<ALGO, V, T, DHI, DLO, W>::run(cls[c].degree, Qu, lane_off, Rt, cls[c].ids, r_begin, r_end, Ct, a.edge_col, a.qpos);
}

/* checkResult, decodeCL.c:88-108, on the bit masks: one thread per row XORs the
 * 64-frame words of its columns (an L2-resident gather), a wave ORs its rows and
 * issues one atomic per word.  Runs after every variable-node round. */
struct SyndromeArgs {
    const int32_t *__restrict__ row_ptr;  /* [M+1] */
    const int32_t *__restrict__ edge_col; /* [E]   */
    const uint64_t *__restrict__ hard;    /* [T][N][V] */
    uint64_t *__restrict__ fail;          /* [T][V] */
    const uint64_t *__restrict__ done;    /* [T][V] */
    int32_t M, N;
    int32_t tiles;                        /* > 0: XCD-aware 1-D grid (see syndrome_kernel) */
    int32_t row_blocks;
    TailRef tail;
};

template <int V> __global__ __launch_bounds__(kBlock) void syndrome_kernel(const SyndromeArgs a)
{
    /* A tile's bit masks (N*V*8 B, 2 MB at V = 4) are re-read ~d_c times: keep them in ONE
     * XCD's L2.  Blocks are dealt round-robin over the 8 XCDs, so blocks with equal
     * blockIdx.x % 8 share an L2: block b serves tile (b % 8) + 8 * ((b / 8) / row_blocks).
     * (Speed only; any placement gives the same result.) */
    int tile, rb;
    if (a.tiles > 0) {
        const int xcd = blockIdx.x & 7, j = blockIdx.x >> 3;
        tile = xcd + 8 * (j / a.row_blocks);
        rb = j % a.row_blocks;
        if (tile >= a.tiles) return;
    } else {
        tile = blockIdx.y;
        rb = blockIdx.x;
    }
    tile = tile_select<V>(a.tail, a.done, tile);
    if (tile < 0) return;
    const int m = rb * kBlock + threadIdx.x;
    const uint64_t *hard_t = a.hard + (size_t)tile * (size_t)a.N * V;
    uint64_t s[V];
#pragma unroll
    for (int v = 0; v < V; ++v) s[v] = 0;
    if (m < a.M) {
        for (int p = a.row_ptr[m]; p < a.row_ptr[m + 1]; ++p) {
            const int c = a.edge_col[p];
#pragma unroll
            for (int v = 0; v < V; ++v) s[v] ^= hard_t[(size_t)c * V + v];
        }
    }
    /* OR over the wave, then over the block's 4 waves (LDS), then at most one atomic per
     * block and word -- and none when the bits are already set: every block of a tile ORs into
     * the same word, and same-address device atomics serialise chip-wide (this was 150 us per
     * round at B = 4096).  The plain pre-read may be stale; stale only costs a redundant atomic. */
    __shared__ uint64_t part[kWavesPerBlock][V];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        uint64_t x = s[v];
#pragma unroll
        for (int off = 32; off >= 1; off >>= 1) {
            const uint32_t lo = __shfl_xor((uint32_t)x, off);
            const uint32_t hi = __shfl_xor((uint32_t)(x >> 32), off);
            x |= ((uint64_t)hi << 32) | lo;
        }
        if ((threadIdx.x & 63) == 0) part[threadIdx.x >> 6][v] = x;
    }
    __syncthreads();
    if (threadIdx.x < V) {
        uint64_t x = 0;
#pragma unroll
        for (int w = 0; w < kWavesPerBlock; ++w) x |= part[w][threadIdx.x];
        unsigned long long *dst = reinterpret_cast<unsigned long long *>(&a.fail[(size_t)tile * V + threadIdx.x]);
        const uint64_t seen = __hip_atomic_load(dst, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (x & ~seen) atomicOr(dst, (unsigned long long)x);
    }
}

/* ========================================================================= */
/*                              variable node                                 */
/* ========================================================================= */

struct VarArgs {
    const void *__restrict__ R;           /* [T][E][F] (float or fp16) */
    void *__restrict__ Q;                 /* [T][E][F] */
    const void *__restrict__ chan;        /* [T][N][F] SP: exp(scale*y); MS: y */
    uint64_t *hard;                       /* [T][N][V] read-modify-write */
    const uint64_t *__restrict__ done;    /* [T][V] */
    const int32_t *__restrict__ cls_col;  /* [n_cols] column ids of this degree class */
    const int32_t *__restrict__ cls_edge; /* [n_cols][D] edge ids, ascending */
    int64_t E;
    int32_t N;
    int32_t n_cols;
    int32_t cols_per_wave;
    int32_t write_q;                      /* 0 on the last round (MyLdpc.cpp:1035-1040) */
    int32_t degree;                       /* generic kernel only */
    TailRef tail;
    int32_t tiles_first = 0;              /* grid is (tiles, blocks): grid_pos() */
    /* first Q slot of this class (CheckArgs::qpos: Q is stored in the order its writers produce it): the class owns D
     * streams of n_cols slots, column ci writes message k to slot q_base + k * n_cols + ci -- the waves at work at any
     * moment write D moving fronts (tools/gather_probe.hip: 6079 GB/s, against 5540 with one 8-KiB run per column and
     * 5065 with the messages scattered to their edge ids).  -1: slot = edge id. */
    int64_t q_base = -1;
};

/* Sum-product variable node: hardDecision (decodeCL.c:72-82) and refreshQ
 * (:52-61) share the prefix products.  d[k] is the stored r0-r1 of edge k;
 * r0 = (1+d)/2, r1 = (1-d)/2 (:39-40; the halving is exact).  t = exp(scale*y),
 * priors t/(1+t) and 1/(1+t) (:9-11).  Returns q0-q1 per edge in outq and the
 * full products in full0/full1. */
template <int D, int V>
__device__ __forceinline__ void var_sp(const float (&t)[V], const float (&d)[D][V],
                                       float (&outq)[D][V], float (&full0)[V], float (&full1)[V])
{
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const float den = 1.0f + t[v];
        float pre0 = t[v] / den;
        float pre1 = 1.0f / den;
        float r0[D], r1[D];
#pragma unroll
        for (int k = 0; k < D; ++k) {
            r0[k] = (1.0f + d[k][v]) * 0.5f;
            r1[k] = (1.0f - d[k][v]) * 0.5f;
        }
#pragma unroll
        for (int k = 0; k < D; ++k) {
            float t0 = pre0, t1 = pre1;
#pragma unroll
            for (int j = k + 1; j < D; ++j) { t0 *= r0[j]; t1 *= r1[j]; }
            const float s = t0 + t1;
            const float q0 = t0 / s;
            const float q1 = t1 / s;
            outq[k][v] = q0 - q1;
            pre0 *= r0[k];
            pre1 *= r1[k];
        }
        full0[v] = pre0;
        full1[v] = pre1;
    }
}

/* ---- column-local fusion ------------------------------------------------------------
 * A degree-2 column whose two checks are CONSECUTIVE rows handled by the same wave (the
 * staircase parity columns of an IRA / DVB-S2 code: column K+m sits in rows m and m+1) never
 * needs its check->variable messages in HBM: the wave that has just produced both of them
 * holds everything the variable node needs.  It applies the variable-node update right there
 * (same fp32 operations in the same order as var_kernel, so results stay bit-identical),
 * writes the two NEW variable->check messages into the slots whose old values it has already
 * consumed (only these two rows ever read them), and sets the column's hard bit.  Per such
 * column this saves one 4-byte write and one 4-byte read per edge and frame
 * (DVB-S2 1/2, 16 rows per wave: 12.5 % of all traffic; measured -13.5 % step time).  Columns on a wave's chunk
 * boundary, and all other columns, go through var_kernel as before. */
struct LinkArgs {
    const int32_t *__restrict__ link_col;  /* [n_rows] column linking list row i and i+1, or -1 */
    const int32_t *__restrict__ link_pos;  /* [n_rows] ka | kb << 8: edge position in row i / row i+1 */
    const void *__restrict__ chan;         /* [T][N][F] */
    void *Qw;                              /* = Q, for the in-place writes */
    uint64_t *hard;                        /* [T][N][V] */
    int32_t N;
    int32_t write_q;
    int32_t store_all;                     /* debug taps: also store the fused columns' R */
    /* the few rows of OTHER degrees that are not worth a launch of their own (an IRA code's first row):
     * blocks from link_blocks on take one each per wave, run-time loops */
    const int32_t *__restrict__ extra_e0;  /* [n_extra] first edge id */
    const int32_t *__restrict__ extra_deg; /* [n_extra] */
    int32_t n_extra;
    int32_t link_blocks;                   /* blocks of the linked rows proper */
    /* guided chunks: the first n_big row chunks of a tile hold rows_per_wave rows, the rest small_rows.
     * With a grid of (tiles, blocks) -- grid_pos() -- the launch ENDS with the short
     * chunks of all tiles and its last waves are short ones -- with equal chunks the CUs drain over a whole
     * chunk's time (16 rows: 176 us of a 1.39 ms launch, half of it lost). */
    int32_t n_big, small_rows;
};

/* rows [*rb, *re) of chunk c of a class of n_rows rows (host: fusion lists, grid size; device: the kernels) */
__host__ __device__ inline void link_chunk_rows(int c, int rows_per_wave, int n_big, int small_rows, int n_rows,
                                               int *rb, int *re)
{
    int b, e;
    if (c < n_big) { b = c * rows_per_wave; e = b + rows_per_wave; }
    else { b = n_big * rows_per_wave + (c - n_big) * small_rows; e = b + small_rows; }
    *rb = b < n_rows ? b : n_rows;
    *re = e < n_rows ? e : n_rows;
}
/* number of chunks */
__host__ __device__ inline int link_chunk_count(int rows_per_wave, int n_big, int small_rows, int n_rows)
{
    const int rest = n_rows - n_big * rows_per_wave;
    return n_big + (rest > 0 ? (rest + small_rows - 1) / small_rows : 0);
}

/* the trailing blocks of a linked check launch: one left-over row per wave, V values per lane */
template <int ALGO, int V, typename T>
__device__ __forceinline__ void link_extra_rows(const CheckArgs &a, const LinkArgs &g, int tile, int block)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const int w = (block - g.link_blocks) * kWavesPerBlock + wave_id_in_block();
    if (w >= g.n_extra) return;
    const T *Qu = static_cast<const T *>(a.Q) + (size_t)tile * (size_t)a.E * F;
    T *Rt = static_cast<T *>(a.R) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    check_row_generic<ALGO, V, T>(Qu, (unsigned)lane * V, Rt, g.extra_e0[w], g.extra_deg[w], a.qpos);
}

/* build-time experiment hook: -DLDPC_LINK_WIDE_WAVES=n asks the compiler for n waves per SIMD */
#ifdef LDPC_LINK_WIDE_WAVES
#define LDPC_LINK_WIDE_ATTR __attribute__((amdgpu_waves_per_eu(LDPC_LINK_WIDE_WAVES, LDPC_LINK_WIDE_WAVES)))
#else
#define LDPC_LINK_WIDE_ATTR
#endif
template <int ALGO, int D, int V, typename T>
__global__ __launch_bounds__(kBlock) LDPC_LINK_WIDE_ATTR void check_link_kernel(const CheckArgs a, const LinkArgs g)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    if (gp.block >= g.link_blocks) { link_extra_rows<ALGO, V, T>(a, g, tile, gp.block); return; }
    const int wave = gp.block * kWavesPerBlock + wave_id_in_block();
    int r_begin, r_end;
    link_chunk_rows(wave, a.rows_per_wave, g.n_big, g.small_rows, a.n_rows, &r_begin, &r_end);
    const size_t lane_off = (size_t)lane * V;
    /* Q: wave-uniform row addresses plus a 32-bit lane offset (scalar base + vector offset addressing: the slots of a
     * row are unrelated, one 64-bit vector address each would cost 2 registers per message in flight) */
    const unsigned lane_q = (unsigned)lane * V;
    const T *Qt = static_cast<const T *>(a.Q) + (size_t)tile * (size_t)a.E * F;
    T *Qwt = static_cast<T *>(g.Qw) + (size_t)tile * (size_t)a.E * F;
    T *Rt = static_cast<T *>(a.R) + (size_t)tile * (size_t)a.E * F + lane_off;
    const T *chan_t = static_cast<const T *>(g.chan) + (size_t)tile * (size_t)g.N * F + lane_off;
    uint64_t *hard_t = g.hard + (size_t)tile * (size_t)g.N * V;
    uint64_t frozen[V];
#pragma unroll
    for (int v = 0; v < V; ++v) frozen[v] = a.done[(size_t)tile * V + v];

    int pend_col = -1, pend_edge = 0, pend_kb = 0;   /* column opened by the previous row */
    float pend_r[V];
#pragma unroll
    for (int v = 0; v < V; ++v) pend_r[v] = 0.0f;

    /* software pipeline: row r+1's messages are requested before row r is worked on */
    float x[D][V];
    int qc[D], qn[D];                /* Q slots (CheckArgs::qpos) of this row's and of the next row's messages */
#pragma unroll
    for (int k = 0; k < D; ++k) qc[k] = qn[k] = 0;
    if (r_begin < r_end) {
        row_slots<D>(a.qpos, a.cls_e0[r_begin], lane, qc);
#pragma unroll
        for (int k = 0; k < D; ++k) vload<V>(x[k], Qt + (size_t)qc[k] * F + lane_q);
    }
    for (int r = r_begin; r < r_end; ++r) {
        const int e0 = a.cls_e0[r];
        const int next_col = (r + 1 < r_end) ? g.link_col[r] : -1;
        const int pos = g.link_pos[r];
        const int ka = next_col >= 0 ? (pos & 255) : -1;      /* this row's edge into next_col */
        const int kb = pend_col >= 0 ? pend_kb : -1;          /* this row's edge into pend_col */
        /* the next row's slots: asked for here, ahead of the row's arithmetic */
        if (r + 1 < r_end) row_slots<D>(a.qpos, a.cls_e0[r + 1], lane, qn);
        /* everything the pending column needs besides this row's result: request it now */
        float ch[V];
        uint64_t old_w[V];
        if (pend_col >= 0) {
            vload<V>(ch, chan_t + (size_t)pend_col * F);
#pragma unroll
            for (int v = 0; v < V; ++v) old_w[v] = hard_t[(size_t)pend_col * V + v];
        }
        float out[D][V];
        if (ALGO == kAlgoSP) check_sp<D, V>(x, out); else check_ms<D, V>(x, out);
        if (r + 1 < r_end) {
#pragma unroll
            for (int k = 0; k < D; ++k) vload<V>(x[k], Qt + (size_t)qn[k] * F + lane_q);
        }
#pragma unroll
        for (int k = 0; k < D; ++k)
            if ((k != ka && k != kb) || g.store_all) vstore<V>(Rt + (size_t)(e0 + k) * F, out[k]);

        if (pend_col >= 0) {
            /* variable node of pend_col: edges (previous row, ka') then (this row, kb), ascending */
            float rr[2][V], q[2][V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                rr[0][v] = pend_r[v];
                float t = out[0][v];
#pragma unroll
                for (int k = 1; k < D; ++k) t = (k == kb) ? out[k][v] : t;
                rr[1][v] = t;
            }
            uint64_t new_w[V];
            if (ALGO == kAlgoSP) {
                float f0[V], f1[V];
                var_sp<2, V>(ch, rr, q, f0, f1);
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    const bool oldb = (old_w[v] >> lane) & 1ull;
                    const bool b = (f0[v] > f1[v]) ? false : ((f0[v] < f1[v]) ? true : oldb);
                    new_w[v] = __ballot(b);
                }
            } else {
#pragma unroll
                for (int v = 0; v < V; ++v) {
                    float p = ch[v];
                    p += rr[0][v];
                    p += rr[1][v];
                    q[0][v] = p - rr[0][v];
                    q[1][v] = p - rr[1][v];
                    new_w[v] = __ballot(!(p > 0.0f));
                }
            }
            if (lane == 0) {
#pragma unroll
                for (int v = 0; v < V; ++v)
                    hard_t[(size_t)pend_col * V + v] = (old_w[v] & frozen[v]) | (new_w[v] & ~frozen[v]);
            }
            if (g.write_q) {
                int qw1 = qc[0];                              /* slot of this row's edge kb */
#pragma unroll
                for (int k = 1; k < D; ++k) qw1 = (k == kb) ? qc[k] : qw1;
                vstore<V>(Qwt + (size_t)pend_edge * F + lane_q, q[0]);
                vstore<V>(Qwt + (size_t)qw1 * F + lane_q, q[1]);
            }
        }
        pend_col = next_col;
        if (next_col >= 0) {
            pend_edge = qc[0];                                /* SLOT of this row's edge ka: written by the next row */
#pragma unroll
            for (int k = 1; k < D; ++k) pend_edge = (k == ka) ? qc[k] : pend_edge;
            pend_kb = (pos >> 8) & 255;
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float t = out[0][v];
#pragma unroll
                for (int k = 1; k < D; ++k) t = (k == ka) ? out[k][v] : t;
                pend_r[v] = t;
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) qc[k] = qn[k];
    }
}

/* Bits of x at positions 0, V, 2V, ... gathered into the low 64/V bits (V = 1, 2, 4). */
template <int V> __device__ __forceinline__ uint64_t compress_stride(uint64_t x)
{
    if (V == 1) return x;
    if (V == 2) {
        x &= 0x5555555555555555ull;
        x = (x | (x >> 1)) & 0x3333333333333333ull;
        x = (x | (x >> 2)) & 0x0f0f0f0f0f0f0f0full;
        x = (x | (x >> 4)) & 0x00ff00ff00ff00ffull;
        x = (x | (x >> 8)) & 0x0000ffff0000ffffull;
        x = (x | (x >> 16)) & 0x00000000ffffffffull;
        return x;
    }
    x &= 0x1111111111111111ull;
    x = (x | (x >> 3)) & 0x0303030303030303ull;
    x = (x | (x >> 6)) & 0x000f000f000f000full;
    x = (x | (x >> 12)) & 0x000000ff000000ffull;
    x = (x | (x >> 24)) & 0x000000000000ffffull;
    return x;
}

/* check_link_kernel in NARROW waves (1 value per lane, V sub-waves per row chunk): the register
 * footprint of the wide version (114 VGPRs, 4 waves/SIMD) made it the one kernel whose time
 * varied between boxes (1.31-1.55 ms); here 8 waves/SIMD hide the same latency.  Hard bits:
 * sub-wave j holds frames 64j..64j+63 of the tile, i.e. the 64/V-bit field [j*64/V, (j+1)*64/V)
 * of each of the V mask words (bit l of word v = frame V*l+v) -- written as that field alone
 * (16-bit stores at V = 4), no atomics: the fields of different sub-waves are disjoint. */
template <int ALGO, int D, int V, typename T, int W = 1>
__global__ __launch_bounds__(kBlock) void check_link_narrow_kernel(const CheckArgs a, const LinkArgs g)
{
    /* W values per lane: V / W sub-waves cover a row's 64*V-frame segment (W = 1: the narrow kernel;
     * W = 2 at V = 4: 8 bytes per lane, half the registers of the wide kernel at twice its occupancy). */
    constexpr size_t F = 64 * V;
    constexpr int SUBS = V / W;                      /* sub-waves per row chunk */
    constexpr int FB = 64 / SUBS;                    /* bits per field */
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    if (gp.block >= g.link_blocks) { link_extra_rows<ALGO, V, T>(a, g, tile, gp.block); return; }
    const int wave = gp.block * kWavesPerBlock + wave_id_in_block();
    const int sub = wave % SUBS;
    int r_begin, r_end;
    link_chunk_rows(wave / SUBS, a.rows_per_wave, g.n_big, g.small_rows, a.n_rows, &r_begin, &r_end);
    const size_t lane_off = (size_t)sub * 64 * W + (size_t)lane * W;
    const unsigned lane_q = (unsigned)lane_off;      /* Q: wave-uniform row addresses + 32-bit lane offset (check_link_kernel) */
    const T *Qt = static_cast<const T *>(a.Q) + (size_t)tile * (size_t)a.E * F;
    T *Qwt = static_cast<T *>(g.Qw) + (size_t)tile * (size_t)a.E * F;
    T *Rt = static_cast<T *>(a.R) + (size_t)tile * (size_t)a.E * F + lane_off;
    const T *chan_t = static_cast<const T *>(g.chan) + (size_t)tile * (size_t)g.N * F + lane_off;
    /* this lane's value w is frame 64*W*sub + W*lane + w of the tile = V*l' + v'
     *   ->  word v' = (W*lane + w) % V, bit sub*FB + (W*lane + w) / V */
    uint8_t *hard_b = reinterpret_cast<uint8_t *>(g.hard + (size_t)tile * (size_t)g.N * V) + (size_t)sub * (FB / 8);
    /* frozen frames of this sub-wave, per word (lanes 0..V-1 keep the field of word `lane`) */
    const uint64_t frozen_field = (a.done[(size_t)tile * V + (lane < V ? lane : 0)] >> (sub * FB)) &
                                  (FB == 64 ? ~0ull : ((1ull << (FB & 63)) - 1ull));

    auto load_field = [&](int col, int word) -> uint64_t {
        const uint8_t *p = hard_b + ((size_t)col * V + word) * 8;
        if (FB == 64) return *reinterpret_cast<const uint64_t *>(p);
        if (FB == 32) return *reinterpret_cast<const uint32_t *>(p);
        return *reinterpret_cast<const uint16_t *>(p);
    };

    int pend_col = -1, pend_edge = 0, pend_kb = 0;
    float pend_r[W];
#pragma unroll
    for (int w = 0; w < W; ++w) pend_r[w] = 0.0f;
    float x[D][W];
    int qc[D], qn[D];                /* Q slots of this row's and of the next row's messages */
#pragma unroll
    for (int k = 0; k < D; ++k) qc[k] = qn[k] = 0;
    if (r_begin < r_end) {
        row_slots<D>(a.qpos, a.cls_e0[r_begin], lane, qc);
#pragma unroll
        for (int k = 0; k < D; ++k) vload<W>(x[k], Qt + (size_t)qc[k] * F + lane_q);
    }
    for (int r = r_begin; r < r_end; ++r) {
        const int e0 = a.cls_e0[r];
        const int next_col = (r + 1 < r_end) ? g.link_col[r] : -1;
        const int pos = g.link_pos[r];
        const int ka = next_col >= 0 ? (pos & 255) : -1;
        const int kb = pend_col >= 0 ? pend_kb : -1;
        if (r + 1 < r_end) row_slots<D>(a.qpos, a.cls_e0[r + 1], lane, qn);
        float ch[W];
        uint64_t old_mine[W];
#pragma unroll
        for (int w = 0; w < W; ++w) { ch[w] = 0.0f; old_mine[w] = 0; }
        if (pend_col >= 0) {
            vload<W>(ch, chan_t + (size_t)pend_col * F);
#pragma unroll
            for (int w = 0; w < W; ++w) old_mine[w] = load_field(pend_col, (W * lane + w) % V);
        }
        float out[D][W];
        if (ALGO == kAlgoSP) check_sp<D, W>(x, out); else check_ms<D, W>(x, out);
        if (r + 1 < r_end) {
#pragma unroll
            for (int k = 0; k < D; ++k) vload<W>(x[k], Qt + (size_t)qn[k] * F + lane_q);
        }
#pragma unroll
        for (int k = 0; k < D; ++k)
            if ((k != ka && k != kb) || g.store_all) vstore<W>(Rt + (size_t)(e0 + k) * F, out[k]);

        if (pend_col >= 0) {
            float rr[2][W], q[2][W];
#pragma unroll
            for (int w = 0; w < W; ++w) {
                rr[0][w] = pend_r[w];
                float t = out[0][w];
#pragma unroll
                for (int k = 1; k < D; ++k) t = (k == kb) ? out[k][w] : t;
                rr[1][w] = t;
            }
            uint64_t ballot[W];
            if (ALGO == kAlgoSP) {
                float f0[W], f1[W];
                var_sp<2, W>(ch, rr, q, f0, f1);
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    const bool oldb = (old_mine[w] >> ((W * lane + w) / V)) & 1ull;
                    const bool bit = (f0[w] > f1[w]) ? false : ((f0[w] < f1[w]) ? true : oldb);
                    ballot[w] = __ballot(bit);
                }
            } else {
#pragma unroll
                for (int w = 0; w < W; ++w) {
                    float p = ch[w];
                    p += rr[0][w];
                    p += rr[1][w];
                    q[0][w] = p - rr[0][w];
                    q[1][w] = p - rr[1][w];
                    ballot[w] = __ballot(!(p > 0.0f));
                }
            }
            if (lane < V) {
                /* field of word `lane` = W*j + w: value w of the lanes j, j + SUBS, j + 2*SUBS, ... */
                uint64_t bw = ballot[0];
#pragma unroll
                for (int w = 1; w < W; ++w) bw = (lane % W == w) ? ballot[w] : bw;
                const uint64_t nf = compress_stride<SUBS>(bw >> (lane / W));
                const uint64_t of = load_field(pend_col, lane);
                const uint64_t res = (of & frozen_field) | (nf & ~frozen_field);
                uint8_t *p = hard_b + ((size_t)pend_col * V + lane) * 8;
                if (FB == 64) *reinterpret_cast<uint64_t *>(p) = res;
                else if (FB == 32) *reinterpret_cast<uint32_t *>(p) = (uint32_t)res;
                else *reinterpret_cast<uint16_t *>(p) = (uint16_t)res;
            }
            if (g.write_q) {
                int qw1 = qc[0];
#pragma unroll
                for (int k = 1; k < D; ++k) qw1 = (k == kb) ? qc[k] : qw1;
                vstore<W>(Qwt + (size_t)pend_edge * F + lane_q, q[0]);
                vstore<W>(Qwt + (size_t)qw1 * F + lane_q, q[1]);
            }
        }
        pend_col = next_col;
        if (next_col >= 0) {
            pend_edge = qc[0];                                /* SLOT of this row's edge ka */
#pragma unroll
            for (int k = 1; k < D; ++k) pend_edge = (k == ka) ? qc[k] : pend_edge;
            pend_kb = (pos >> 8) & 255;
#pragma unroll
            for (int w = 0; w < W; ++w) {
                float t = out[0][w];
#pragma unroll
                for (int k = 1; k < D; ++k) t = (k == ka) ? out[k][w] : t;
                pend_r[w] = t;
            }
        }
#pragma unroll
        for (int k = 0; k < D; ++k) qc[k] = qn[k];
    }
}

/* The same kernel with the row inputs requested TWO rows ahead (three buffers in rotation): a wave
 * of the kernel above has one row (7 x 256 B) in flight while it works -- 57 KB per CU at 32 waves --
 * and runs at 83-94 % of the box's copy rate depending on the box; this one keeps two. */
template <int ALGO, int D, int V, typename T>
__global__ __launch_bounds__(kBlock) void check_link_narrow2_kernel(const CheckArgs a, const LinkArgs g)
{
    constexpr size_t F = 64 * V;
    constexpr int FB = 64 / V;                       /* bits per field */
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    if (gp.block >= g.link_blocks) { link_extra_rows<ALGO, V, T>(a, g, tile, gp.block); return; }
    const int wave = gp.block * kWavesPerBlock + wave_id_in_block();
    const int sub = wave % V;
    int r_begin, r_end;
    link_chunk_rows(wave / V, a.rows_per_wave, g.n_big, g.small_rows, a.n_rows, &r_begin, &r_end);
    const size_t lane_off = (size_t)sub * 64 + lane;
    const T *Qt = static_cast<const T *>(a.Q) + (size_t)tile * (size_t)a.E * F + lane_off;
    T *Qwt = static_cast<T *>(g.Qw) + (size_t)tile * (size_t)a.E * F + lane_off;
    T *Rt = static_cast<T *>(a.R) + (size_t)tile * (size_t)a.E * F + lane_off;
    const T *chan_t = static_cast<const T *>(g.chan) + (size_t)tile * (size_t)g.N * F + lane_off;
    /* this lane's frame 64*sub + lane = V*l' + v'  ->  word v' = lane % V, bit sub*FB + lane / V */
    const int my_word = lane % V, my_bit = lane / V;
    uint8_t *hard_b = reinterpret_cast<uint8_t *>(g.hard + (size_t)tile * (size_t)g.N * V) + (size_t)sub * (FB / 8);
    /* frozen frames of this sub-wave, per word (lanes 0..V-1 keep the field of word `lane`) */
    const uint64_t frozen_field = (a.done[(size_t)tile * V + (lane < V ? lane : 0)] >> (sub * FB)) &
                                  (FB == 64 ? ~0ull : ((1ull << FB) - 1ull));

    auto load_field = [&](int col, int word) -> uint64_t {
        const uint8_t *p = hard_b + ((size_t)col * V + word) * 8;
        if (V == 1) return *reinterpret_cast<const uint64_t *>(p);
        if (V == 2) return *reinterpret_cast<const uint32_t *>(p);
        return *reinterpret_cast<const uint16_t *>(p);
    };

    int pend_col = -1, pend_edge = 0, pend_kb = 0;
    float pend_r = 0.0f;
    /* three row buffers in rotation: while row r is worked on, rows r+1 and r+2 are in flight */
    float b0[D], b1[D], b2[D];
    auto load_row = [&](float (&dst)[D], int r) {
        const int e = a.cls_e0[r];
#pragma unroll
        for (int k = 0; k < D; ++k) { float t[1]; vload<1>(t, Qt + (size_t)edge_slot(a.qpos, e + k) * F); dst[k] = t[0]; }
    };
    if (r_begin < r_end) load_row(b0, r_begin);
    if (r_begin + 1 < r_end) load_row(b1, r_begin + 1);
    auto step = [&](int r, float (&x)[D], float (&pre)[D]) {
        const int e0 = a.cls_e0[r];
        const int next_col = (r + 1 < r_end) ? g.link_col[r] : -1;
        const int pos = g.link_pos[r];
        const int ka = next_col >= 0 ? (pos & 255) : -1;
        const int kb = pend_col >= 0 ? pend_kb : -1;
        float ch[1] = {0.0f};
        uint64_t old_mine = 0;
        if (pend_col >= 0) {
            vload<1>(ch, chan_t + (size_t)pend_col * F);
            old_mine = load_field(pend_col, my_word);
        }
        if (r + 2 < r_end) load_row(pre, r + 2);
        float xx[D][1], out[D][1];
#pragma unroll
        for (int k = 0; k < D; ++k) xx[k][0] = x[k];
        if (ALGO == kAlgoSP) check_sp<D, 1>(xx, out); else check_ms<D, 1>(xx, out);
#pragma unroll
        for (int k = 0; k < D; ++k)
            if ((k != ka && k != kb) || g.store_all) vstore<1>(Rt + (size_t)(e0 + k) * F, out[k]);

        if (pend_col >= 0) {
            float rr[2][1], q[2][1];
            rr[0][0] = pend_r;
            float t = out[0][0];
#pragma unroll
            for (int k = 1; k < D; ++k) t = (k == kb) ? out[k][0] : t;
            rr[1][0] = t;
            bool bit;
            if (ALGO == kAlgoSP) {
                float f0[1], f1[1];
                var_sp<2, 1>(ch, rr, q, f0, f1);
                const bool oldb = (old_mine >> my_bit) & 1ull;
                bit = (f0[0] > f1[0]) ? false : ((f0[0] < f1[0]) ? true : oldb);
            } else {
                float p = ch[0];
                p += rr[0][0];
                p += rr[1][0];
                q[0][0] = p - rr[0][0];
                q[1][0] = p - rr[1][0];
                bit = !(p > 0.0f);
            }
            const uint64_t ballot = __ballot(bit);
            if (lane < V) {
                const uint64_t nf = compress_stride<V>(ballot >> lane);
                const uint64_t of = load_field(pend_col, lane);
                const uint64_t res = (of & frozen_field) | (nf & ~frozen_field);
                uint8_t *p = hard_b + ((size_t)pend_col * V + lane) * 8;
                if (V == 1) *reinterpret_cast<uint64_t *>(p) = res;
                else if (V == 2) *reinterpret_cast<uint32_t *>(p) = (uint32_t)res;
                else *reinterpret_cast<uint16_t *>(p) = (uint16_t)res;
            }
            if (g.write_q) {
                vstore<1>(Qwt + (size_t)edge_slot(a.qpos, pend_edge) * F, q[0]);
                vstore<1>(Qwt + (size_t)edge_slot(a.qpos, e0 + kb) * F, q[1]);
            }
        }
        pend_col = next_col;
        if (next_col >= 0) {
            pend_edge = e0 + ka;
            pend_kb = (pos >> 8) & 255;
            float t = out[0][0];
#pragma unroll
            for (int k = 1; k < D; ++k) t = (k == ka) ? out[k][0] : t;
            pend_r = t;
        }
    };
    for (int r = r_begin; r < r_end; r += 3) {
        step(r, b0, b2);
        if (r + 1 < r_end) step(r + 1, b1, b0);
        if (r + 2 < r_end) step(r + 2, b2, b1);
    }
}

/* The variable node keeps WIDE waves (V values per lane, whole 64*V-frame segments per
 * wave-instruction): narrow waves as in check_kernel were measured 10 % slower here
 * (1.69 vs 1.53 ms per round at B = 4096), 2 values per lane no faster (1.11 vs 1.12 ms) --
 * the gather prefers fewer, larger requests.  Segments twice as large would not help either:
 * tools/gather_probe.hip moves this exact traffic at 94-95 % of the box's copy rate with 1-, 2- and
 * 4-KiB segments alike (profiles/r02_gather_probe.txt). */
template <int ALGO, int D, int V, typename T>
__device__ __forceinline__ void var_columns(const T *Rt, T *Qt, const T *chan_t, uint64_t *hard_t, const uint64_t (&frozen)[V],
                                            const int32_t *__restrict__ cls_col, const int32_t *__restrict__ cls_edge,
                                            int c_begin, int c_end, int write_q, int lane, int64_t q_base, int n_cls)
{
    constexpr size_t F = 64 * V;
    for (int ci = c_begin; ci < c_end; ++ci) {
        const int n = cls_col[ci];
        int e[D];
#pragma unroll
        for (int k = 0; k < D; ++k) e[k] = cls_edge[(size_t)ci * D + k];
        float ch[V], r[D][V], q[D][V];
        vload<V>(ch, chan_t + (size_t)n * F);
#pragma unroll
        for (int k = 0; k < D; ++k) vload<V>(r[k], Rt + (size_t)e[k] * F);
        uint64_t old_w[V], new_w[V];
#pragma unroll
        for (int v = 0; v < V; ++v) old_w[v] = hard_t[(size_t)n * V + v];

        if (ALGO == kAlgoSP) {
            float f0[V], f1[V];
            var_sp<D, V>(ch, r, q, f0, f1);
#pragma unroll
            for (int v = 0; v < V; ++v) {
                /* decodeCL.c:78-82: ties and NaN keep the previous bit */
                const bool oldb = (old_w[v] >> lane) & 1ull;
                const bool b = (f0[v] > f1[v]) ? false : ((f0[v] < f1[v]) ? true : oldb);
                new_w[v] = __ballot(b);
            }
        } else {
            /* refreshPostPMS decodeCL.c:157-166, refreshQMS :185 */
#pragma unroll
            for (int v = 0; v < V; ++v) {
                float p = ch[v];
#pragma unroll
                for (int k = 0; k < D; ++k) p += r[k][v];
#pragma unroll
                for (int k = 0; k < D; ++k) q[k][v] = p - r[k][v];
                new_w[v] = __ballot(!(p > 0.0f));
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v)
                hard_t[(size_t)n * V + v] = (old_w[v] & frozen[v]) | (new_w[v] & ~frozen[v]);
        }
        if (write_q) {
            if (q_base >= 0) {
                T *Qc = Qt + ((size_t)q_base + (size_t)ci) * F;            /* this column's slot in stream 0 */
#pragma unroll
                for (int k = 0; k < D; ++k) vstore<V>(Qc + (size_t)k * (size_t)n_cls * F, q[k]);
            } else {
#pragma unroll
                for (int k = 0; k < D; ++k) vstore<V>(Qt + (size_t)e[k] * F, q[k]);
            }
        }
    }
}

/* build-time experiment hook: -DLDPC_VAR_WAVES=n asks the compiler for n waves per SIMD */
#ifdef LDPC_VAR_WAVES
#define LDPC_VAR_ATTR __attribute__((amdgpu_waves_per_eu(LDPC_VAR_WAVES, LDPC_VAR_WAVES)))
#else
#define LDPC_VAR_ATTR
#endif
template <int ALGO, int D, int V, typename T>
__global__ __launch_bounds__(kBlock) LDPC_VAR_ATTR void var_kernel(const VarArgs a)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    const int wave = gp.block * kWavesPerBlock + wave_id_in_block();
    const int c_begin = wave * a.cols_per_wave;
    const int c_end = min(c_begin + a.cols_per_wave, a.n_cols);
    const T *Rt = static_cast<const T *>(a.R) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    T *Qt = static_cast<T *>(a.Q) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    const T *chan_t = static_cast<const T *>(a.chan) + (size_t)tile * (size_t)a.N * F + (size_t)lane * V;
    uint64_t *hard_t = a.hard + (size_t)tile * (size_t)a.N * V;
    uint64_t frozen[V];
#pragma unroll
    for (int v = 0; v < V; ++v) frozen[v] = a.done[(size_t)tile * V + v];
    var_columns<ALGO, D, V, T>(Rt, Qt, chan_t, hard_t, frozen, a.cls_col, a.cls_edge, c_begin, c_end, a.write_q, lane, a.q_base, a.n_cols);
}

template <int ALGO, int V, typename T, int D, int DLO> struct VarDispatch {
    static __device__ __forceinline__ void run(int deg, const T *Rt, T *Qt, const T *chan_t, uint64_t *hard_t,
                                               const uint64_t (&frozen)[V], const int32_t *col, const int32_t *edge,
                                               int cb, int ce, int write_q, int lane, int64_t q_base, int n_cls)
    {
        if (deg == D) var_columns<ALGO, D, V, T>(Rt, Qt, chan_t, hard_t, frozen, col, edge, cb, ce, write_q, lane, q_base, n_cls);
        else VarDispatch<ALGO, V, T, D - 1, DLO>::run(deg, Rt, Qt, chan_t, hard_t, frozen, col, edge, cb, ce, write_q, lane, q_base, n_cls);
    }
};
template <int ALGO, int V, typename T, int DLO> struct VarDispatch<ALGO, V, T, DLO, DLO> {
    static __device__ __forceinline__ void run(int, const T *Rt, T *Qt, const T *chan_t, uint64_t *hard_t,
                                               const uint64_t (&frozen)[V], const int32_t *col, const int32_t *edge,
                                               int cb, int ce, int write_q, int lane, int64_t q_base, int n_cls)
    {
        var_columns<ALGO, DLO, V, T>(Rt, Qt, chan_t, hard_t, frozen, col, edge, cb, ce, write_q, lane, q_base, n_cls);
    }
};

/* every column class of the degree bucket DLO..DHI in one launch (GroupClass table above) */
template <int ALGO, int V, typename T, int DLO, int DHI>
__global__ __launch_bounds__(kBlock) void var_group_kernel(const VarArgs a, const GroupClass *__restrict__ cls, int n_classes)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    int c = 0;
    while (c + 1 < n_classes && gp.block >= cls[c + 1].block_begin) ++c;
    const int wave = (gp.block - cls[c].block_begin) * kWavesPerBlock + wave_id_in_block();
    const int c_begin = wave * a.cols_per_wave;
    const int c_end = min(c_begin + a.cols_per_wave, cls[c].count);
    const T *Rt = static_cast<const T *>(a.R) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    T *Qt = static_cast<T *>(a.Q) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    const T *chan_t = static_cast<const T *>(a.chan) + (size_t)tile * (size_t)a.N * F + (size_t)lane * V;
    uint64_t *hard_t = a.hard + (size_t)tile * (size_t)a.N * V;
    uint64_t frozen[V];
#pragma unroll
    for (int v = 0; v < V; ++v) frozen[v] = a.done[(size_t)tile * V + v];
    VarDispatch<ALGO, V, T, DHI, DLO>::run(cls[c].degree, Rt, Qt, chan_t, hard_t, frozen, cls[c].ids, cls[c].edges,
                                           c_begin, c_end, a.write_q, lane, cls[c].q_base, cls[c].count);
}

template <int ALGO, int V, typename T>
__global__ __launch_bounds__(kBlock) void var_kernel_generic(const VarArgs a)
{
    constexpr size_t F = 64 * V;
    const int lane = threadIdx.x & 63;
    const GridPos gp = grid_pos(a.tiles_first);
    const int tile = tile_select<V>(a.tail, a.done, gp.row);
    if (tile < 0) return;
    const int wave = gp.block * kWavesPerBlock + wave_id_in_block();
    const int c_begin = wave * a.cols_per_wave;
    const int c_end = min(c_begin + a.cols_per_wave, a.n_cols);
    const int D = a.degree;
    const T *Rt = static_cast<const T *>(a.R) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    T *Qt = static_cast<T *>(a.Q) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
    const T *chan_t = static_cast<const T *>(a.chan) + (size_t)tile * (size_t)a.N * F + (size_t)lane * V;
    uint64_t *hard_t = a.hard + (size_t)tile * (size_t)a.N * V;
    uint64_t frozen[V];
#pragma unroll
    for (int v = 0; v < V; ++v) frozen[v] = a.done[(size_t)tile * V + v];

    for (int ci = c_begin; ci < c_end; ++ci) {
        const int n = a.cls_col[ci];
        const int32_t *e = a.cls_edge + (size_t)ci * D;
        auto qs = [&](int k) -> size_t { return a.q_base >= 0 ? (size_t)a.q_base + (size_t)k * a.n_cols + ci : (size_t)e[k]; };
        float ch[V];
        vload<V>(ch, chan_t + (size_t)n * F);
        uint64_t old_w[V], new_w[V];
#pragma unroll
        for (int v = 0; v < V; ++v) old_w[v] = hard_t[(size_t)n * V + v];
        if (ALGO == kAlgoSP) {
            float p0[V], p1[V];
#pragma unroll
            for (int v = 0; v < V; ++v) {
                const float den = 1.0f + ch[v];
                p0[v] = ch[v] / den;
                p1[v] = 1.0f / den;
            }
            for (int k = 0; k <= D; ++k) {     /* k == D: the full product (hard decision) */
                float t0[V], t1[V];
#pragma unroll
                for (int v = 0; v < V; ++v) { t0[v] = p0[v]; t1[v] = p1[v]; }
                for (int j = 0; j < D; ++j) {
                    if (j == k) continue;
                    float dj[V];
                    vload<V>(dj, Rt + (size_t)e[j] * F);
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        t0[v] *= (1.0f + dj[v]) * 0.5f;
                        t1[v] *= (1.0f - dj[v]) * 0.5f;
                    }
                }
                if (k < D) {
                    if (a.write_q) {
                        float o[V];
#pragma unroll
                        for (int v = 0; v < V; ++v) {
                            const float s = t0[v] + t1[v];
                            o[v] = t0[v] / s - t1[v] / s;
                        }
                        vstore<V>(Qt + qs(k) * F, o);
                    }
                } else {
#pragma unroll
                    for (int v = 0; v < V; ++v) {
                        const bool oldb = (old_w[v] >> lane) & 1ull;
                        const bool b = (t0[v] > t1[v]) ? false : ((t0[v] < t1[v]) ? true : oldb);
                        new_w[v] = __ballot(b);
                    }
                }
            }
        } else {
            float p[V];
#pragma unroll
            for (int v = 0; v < V; ++v) p[v] = ch[v];
            for (int j = 0; j < D; ++j) {
                float rj[V];
                vload<V>(rj, Rt + (size_t)e[j] * F);
#pragma unroll
                for (int v = 0; v < V; ++v) p[v] += rj[v];
            }
#pragma unroll
            for (int v = 0; v < V; ++v) new_w[v] = __ballot(!(p[v] > 0.0f));
            if (a.write_q) {
                for (int j = 0; j < D; ++j) {
                    float rj[V], o[V];
                    vload<V>(rj, Rt + (size_t)e[j] * F);
#pragma unroll
                    for (int v = 0; v < V; ++v) o[v] = p[v] - rj[v];
                    vstore<V>(Qt + qs(j) * F, o);
                }
            }
        }
        if (lane == 0) {
#pragma unroll
            for (int v = 0; v < V; ++v)
                hard_t[(size_t)n * V + v] = (old_w[v] & frozen[v]) | (new_w[v] & ~frozen[v]);
        }
    }
}

/* ========================================================================= */
/*                        init / bookkeeping / pack                           */
/* ========================================================================= */

struct InitArgs {
    const float *__restrict__ llr;        /* [frames][N] frame-major (reference layout) */
    void *__restrict__ chan;              /* [T][N][F] (float or fp16) */
    void *__restrict__ Q;                 /* [T][E][F] */
    uint64_t *__restrict__ hard;          /* [T][N][V] */
    const int32_t *__restrict__ col_ptr;  /* [N+1] */
    const int32_t *__restrict__ col_edge; /* [E] */
    int64_t E;
    int64_t frames;
    int32_t N;
    float llr_scale;
};

/* decodeInit (decodeCL.c:3-22) / decodeInitMS (:113-124) + the transpose from the
 * reference's frame-major input to the tile layout.  One block = 32 columns x one
 * tile; the patch crosses LDS so both the read (along n) and the writes (along
 * frames) are contiguous.  Frames past `frames` are filled with y = +1. */
constexpr int kInitCols = 32;

template <int ALGO, int V, typename T>
__global__ __launch_bounds__(kBlock) void init_kernel(const InitArgs a)
{
    constexpr int F = 64 * V;
    constexpr int LD = F + 1;             /* +1 float: conflict-free column writes */
    __shared__ float patch[kInitCols * LD];
    const int tile = blockIdx.y;
    const int n0 = blockIdx.x * kInitCols;
    if ((a.N & 3) == 0 && n0 + kInitCols <= a.N && (reinterpret_cast<uintptr_t>(a.llr) & 15) == 0) {
        /* 16-byte loads, all of a thread's F/32 requests in flight: 8 threads span the 32 columns
         * of a frame, 32 frames per pass */
        const int c4 = (threadIdx.x & 7) * 4, f0 = threadIdx.x >> 3;
        float4 vals[F / 32];
#pragma unroll
        for (int i = 0; i < F / 32; ++i) {
            const int64_t frame = (int64_t)tile * F + f0 + 32 * i;
            vals[i] = float4{1.0f, 1.0f, 1.0f, 1.0f};
            if (frame < a.frames) vals[i] = *reinterpret_cast<const float4 *>(a.llr + (size_t)frame * a.N + n0 + c4);
        }
#pragma unroll
        for (int i = 0; i < F / 32; ++i) {
            const int f = f0 + 32 * i;
            patch[(c4 + 0) * LD + f] = vals[i].x;
            patch[(c4 + 1) * LD + f] = vals[i].y;
            patch[(c4 + 2) * LD + f] = vals[i].z;
            patch[(c4 + 3) * LD + f] = vals[i].w;
        }
    } else {
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
        float ch[V], q[V];
#pragma unroll
        for (int v = 0; v < V; ++v) {
            const float y = patch[c * LD + lane * V + v];
            if (ALGO == kAlgoSP) {
                const float t = ldpc_expf(a.llr_scale * y);         /* decodeCL.c:9 */
                ch[v] = t;
                q[v] = t / (1.0f + t) - 1.0f / (1.0f + t);          /* :10-11, as q0-q1 */
            } else {
                ch[v] = y;                                          /* fp16 storage rounds it here */
                q[v] = y;                                           /* :121 */
            }
        }
        vstore<V>(static_cast<T *>(a.chan) + ((size_t)tile * a.N + n) * F + (size_t)lane * V, ch);
        if (a.Q) {      /* nullptr: round 1's check kernels read the channel values themselves (CheckArgs::first_chan) */
            /* the column's Q slots: one load by the wave's first lanes, a broadcast per store (a load per store
             * serialised the stores behind the index latency) */
            const int p0 = a.col_ptr[n], deg = a.col_ptr[n + 1] - p0;
            T *Qt = static_cast<T *>(a.Q) + (size_t)tile * (size_t)a.E * F + (size_t)lane * V;
            for (int k0 = 0; k0 < deg; k0 += 64) {
                const int m = min(64, deg - k0);
                const int mine = a.col_edge[p0 + k0 + (lane < m ? lane : m - 1)];
                for (int k = 0; k < m; ++k)
                    vstore<V>(Qt + (size_t)__builtin_amdgcn_readlane(mine, k) * F, q);
            }
        }
        if (lane < V) a.hard[((size_t)tile * a.N + n) * V + lane] = 0;
    }
}

struct StateArgs {
    uint64_t *__restrict__ done;        /* [T][V] */
    const uint64_t *__restrict__ fail;  /* [T][V] syndrome of round `iter` */
    int32_t *__restrict__ iters;        /* [T][F] */
    int32_t *__restrict__ active;       /* [1] += number of frames still running */
    int64_t frames;
    int32_t iter;                       /* round whose syndrome `fail` holds; 0 = initialise */
    int32_t max_iter;
    int32_t freeze;                     /* early_term */
    TailRef tail;
    int32_t *__restrict__ running;      /* [max_iter + 2] or nullptr: running[iter] += frames still running */
    int32_t *__restrict__ tile_rounds;  /* [1] or nullptr: += 1 per tile that still had a running frame when round `iter` began,
                                           i.e. whose kernels did not leave at entry (what the round's traffic is priced at) */
};

/* isDones bookkeeping (decodeCL.c:48-49, checkDones :296-300): a frame whose
 * syndrome is clean after round `iter` is frozen with iters = iter.
 * iter == 0 initialises: padding frames are born frozen, iters = max_iter. */
template <int V> __global__ void state_kernel(const StateArgs a)
{
    constexpr int F = 64 * V;
    int tile = blockIdx.x;
    if (a.tail.state && a.iter > 0 && a.tail.state[0]) {      /* handed over: only the overflow tiles live */
        if (tile >= a.tail.n_tiles) return;
        tile = a.tail.base_tile + tile;
    }
    const int lane = threadIdx.x;      /* 64 threads */
    int n_active = 0;
    bool worked = false;               /* some frame of the tile was running when this round began */
#pragma unroll
    for (int v = 0; v < V; ++v) {
        const int64_t frame = (int64_t)tile * F + (int64_t)lane * V + v;
        const bool valid = frame < a.frames;
        uint64_t d;
        if (a.iter == 0) {
            d = __ballot(!valid);
            a.iters[(size_t)tile * F + lane * V + v] = a.max_iter;
        } else {
            const uint64_t old = a.done[(size_t)tile * V + v];
            worked = worked || old != ~0ull;
            const uint64_t clean = ~a.fail[(size_t)tile * V + v];
            const uint64_t newly = clean & ~old;
            if ((newly >> lane) & 1ull) a.iters[(size_t)tile * F + lane * V + v] = a.iter;
            d = a.freeze ? (old | clean) : old;
        }
        if (lane == 0) a.done[(size_t)tile * V + v] = d;
        n_active += __popcll(~d);
    }
    if (lane == 0 && n_active && a.active) atomicAdd(a.active, n_active);
    if (lane == 0 && n_active && a.running) atomicAdd(&a.running[a.iter], n_active);
    if (lane == 0 && worked && a.tile_rounds) atomicAdd(a.tile_rounds, 1);
}

/* ---- tail compaction (early termination): when only a few frames of a large batch are still
 * running, their state moves into ONE 64-frame tile of a small child decoder, which finishes
 * them; the parent's tiles stop being launched.  Frames are independent, so nothing changes for
 * any frame except where its state lives.
 * compact_list_kernel: map[slot] = frame index of every running frame (slot order is arbitrary). */
template <int V> __global__ __launch_bounds__(kBlock) void compact_list_kernel(const uint64_t *__restrict__ done, int64_t frames,
                                                                               int32_t *__restrict__ map, int32_t *__restrict__ count,
                                                                               int32_t capacity)
{
    constexpr int F = 64 * V;
    const int64_t f = (int64_t)blockIdx.x * kBlock + threadIdx.x;
    if (f >= frames) return;
    const int64_t tile = f / F;
    const int fi = (int)(f % F);
    if ((done[tile * V + fi % V] >> (fi / V)) & 1ull) return;
    const int slot = atomicAdd(count, 1);
    if (slot < capacity) map[slot] = (int32_t)f;
}

/* The child's tiles hold cf = 64 * cv frames (cv = 1, or 4 for the 1024-frame child of large batches); slot j of the
 * hand-over is frame j % cf of child tile j / cf, i.e. element j % cf of a row's segment and bit (j % cf) / cv of mask
 * word (j % cf) % cv. */
__host__ __device__ inline size_t child_elem(int j, int cf, int64_t rows, int64_t i) { return ((size_t)(j / cf) * rows + i) * cf + (j % cf); }

/* dst[i][j] = src[tile(map[j])][i][position of map[j]] for j < count (0 beyond): the per-edge /
 * per-column values of the running frames, gathered into the child's tiles.  One wave per row i and 64 slots. */
template <int V, typename T>
__global__ __launch_bounds__(kBlock) void compact_gather_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                                const int32_t *__restrict__ map, int32_t count, int64_t rows, int cf,
                                                                const int32_t *__restrict__ src_row = nullptr,
                                                                const int32_t *__restrict__ dst_row = nullptr)
{
    /* src_row / dst_row: where row i lives in the parent's / the child's array (the Q arrays: CheckArgs::qpos of either
     * decoder -- their column-fused edges differ with the tile size); nullptr: row i */
    constexpr int F = 64 * V;
    const int j = threadIdx.x & 63, jg = blockIdx.y * 64 + j;               /* slot jg */
    const int64_t i = (int64_t)blockIdx.x * kWavesPerBlock + wave_id_in_block();
    if (i >= rows) return;
    const int64_t is = src_row ? src_row[i] : i, id = dst_row ? dst_row[i] : i;
    T val = (T)0;
    if (jg < count) {
        const int64_t f = map[jg];
        val = src[((f / F) * rows + is) * F + (f % F)];
    }
    dst[child_elem(jg, cf, rows, id)] = val;
}

/* The same gather for MANY frames (hundreds, sitting in every tile of the batch): one thread per element as
 * above touches a 64-byte sector for every 2- or 4-byte value, i.e. reads the whole array through the L2 in
 * sector-sized requests.  Here a block takes one row i, streams that row's segments of all `tiles` parent
 * tiles into LDS with coalesced 16-byte loads, chunk by chunk, and picks the running frames' values from
 * there.  HBM traffic = one pass over the parent array (a third of a round) however many frames move. */
constexpr int kGatherChunk = 4096;      /* elements staged at a time (16 KB of fp32) */
/* rows a block walks, the next one's loads in flight: 2-byte rows (8 KB for 4096 frames) left the memory pipe idle with one row
 * per block (3.6 -> 4.9 TB/s with eight); 4-byte rows have enough in flight with one row per block and LOSE with eight (650 -> 880 us
 * for the headline code's 3.7 GB: the block's serial stage / pick phases then stand in the way) */
template <typename T> constexpr int gather_rows_per_block() { return sizeof(T) == 2 ? 8 : 1; }
template <int V, typename T>
__global__ __launch_bounds__(kBlock) void compact_gather_rows_kernel(const T *__restrict__ src, T *__restrict__ dst,
                                                                     const int32_t *__restrict__ map, int32_t count,
                                                                     int64_t rows, int32_t tiles, int cf,
                                                                     const int32_t *__restrict__ src_row = nullptr,
                                                                     const int32_t *__restrict__ dst_row = nullptr)
{
    constexpr int F = 64 * V;
    constexpr int TPC = kGatherChunk / F;               /* parent tiles per chunk */
    constexpr int VEC = 16 / (int)sizeof(T);            /* 16 bytes per lane (a tile's row segment is F * sizeof(T) >= 128 bytes, 16-byte aligned) */
    constexpr int NV = kGatherChunk / (kBlock * VEC);   /* 16-byte vectors a thread stages per chunk */
    __shared__ __attribute__((aligned(16))) T stage[kGatherChunk];
    const int cslots = ((count + cf - 1) / cf) * cf;    /* child slots in use (whole child tiles) */
    constexpr int RPB = gather_rows_per_block<T>();
    const int64_t r0 = (int64_t)blockIdx.x * RPB;
    const int64_t r1 = r0 + RPB < rows ? r0 + RPB : rows;
    if (tiles <= TPC) {
        /* the usual case, a row's segments of all tiles fit one chunk: the block walks its rows with the NEXT row's
         * loads in flight while this row's values are picked from LDS (one row per block left the memory pipe idle
         * during the pick: 3.6 TB/s on fp16 rows of 8 KB) */
        vf4 pre[NV];
        auto issue = [&](int64_t r) {
            const int64_t i = src_row ? src_row[r] : r;
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int k = (v * kBlock + (int)threadIdx.x) * VEC;
                if (k < tiles * F) pre[v] = ld_stream(reinterpret_cast<const vf4 *>(src + ((size_t)(k / F) * rows + i) * F + (k % F)));
            }
        };
        if (r0 < r1) issue(r0);
        /* this thread's slots j = threadIdx.x + u * kBlock: where they come from and where they go is the same for every
         * row (child_elem's run-time divisions, per element, were a quarter of a millisecond of integer arithmetic) */
        constexpr int NS = (2 * kCompactCapacity + kBlock - 1) / kBlock;      /* slots per thread: 1024 frames at most */
        int from[NS];
        size_t to[NS];
#pragma unroll
        for (int u = 0; u < NS; ++u) {
            const int j = (int)threadIdx.x + u * kBlock;
            from[u] = j < count ? map[j] : -1;
            to[u] = j < cslots ? child_elem(j, cf, rows, 0) : 0;
        }
        for (int64_t r = r0; r < r1; ++r) {
#pragma unroll
            for (int v = 0; v < NV; ++v) {
                const int k = (v * kBlock + (int)threadIdx.x) * VEC;
                if (k < tiles * F) *reinterpret_cast<vf4 *>(&stage[k]) = pre[v];
            }
            __syncthreads();
            if (r + 1 < r1) issue(r + 1);
            const int64_t id = dst_row ? dst_row[r] : r;
            T *drow = dst + (size_t)id * cf;                 /* child_elem(j, cf, rows, id) = child_elem(j, cf, rows, 0) + id * cf */
#pragma unroll
            for (int u = 0; u < NS; ++u)
                if ((int)threadIdx.x + u * kBlock < cslots) drow[to[u]] = from[u] >= 0 ? stage[from[u]] : (T)0;
            __syncthreads();
        }
        return;
    }
    for (int64_t r = r0; r < r1; ++r) {
        const int64_t i = src_row ? src_row[r] : r;         /* the row in the parent ... */
        const int64_t id = dst_row ? dst_row[r] : r;        /* ... and in the child */
        for (int t0 = 0; t0 < tiles; t0 += TPC) {
            const int nt = min(TPC, tiles - t0);
            for (int k = threadIdx.x * VEC; k < nt * F; k += kBlock * VEC)
                *reinterpret_cast<vf4 *>(&stage[k]) =
                    ld_stream(reinterpret_cast<const vf4 *>(src + ((size_t)(t0 + k / F) * rows + i) * F + (k % F)));
            __syncthreads();
            for (int j = threadIdx.x; j < cslots; j += kBlock) {
                if (j < count) {
                    const int f = map[j] - t0 * F;
                    if (f >= 0 && f < nt * F) dst[child_elem(j, cf, rows, id)] = stage[f];
                } else if (t0 == 0) {
                    dst[child_elem(j, cf, rows, id)] = (T)0;
                }
            }
            __syncthreads();
        }
    }
}

/* hard bits, parent -> child (sum-product only: its decision rule can keep the previous bit): child bit of slot j <- parent
 * bit of frame map[j].  One wave per column n and group g of 64 slots.  With a child of cv frames per lane the group's slots
 * are frames (g % cv) * 64 ... + 63 of child tile g / cv, i.e. the 64 / cv-bit field number g % cv of each of its cv mask
 * words (bit l of word v = frame cv * l + v) -- written as that field alone, no atomics (compress_stride: the bits of the
 * lanes v, v + cv, ... of the ballot).  The way back is compact_hard_back_kernel. */
template <int V> __global__ __launch_bounds__(kBlock) void compact_hard_kernel(const uint64_t *__restrict__ parent, uint64_t *__restrict__ child,
                                                                               const int32_t *__restrict__ map, int32_t count, int32_t N,
                                                                               int cv)
{
    constexpr int F = 64 * V;
    const int j = threadIdx.x & 63, g = blockIdx.y, jg = g * 64 + j;
    const int64_t n = (int64_t)blockIdx.x * kWavesPerBlock + wave_id_in_block();
    if (n >= N) return;
    const int64_t f = jg < count ? map[jg] : 0;
    const int fi = (int)(f % F);
    const uint64_t word = parent[((f / F) * N + n) * V + fi % V];
    const bool bit = jg < count && ((word >> (fi / V)) & 1ull);
    const uint64_t b = __ballot(bit);
    uint64_t *cw = child + ((size_t)(g / cv) * N + n) * cv;              /* the cv words of child tile g / cv, column n */
    if (cv == 1) {
        if (j == 0) cw[0] = b;
    } else if (cv == 2) {
        if (j < 2) reinterpret_cast<uint32_t *>(cw + j)[g % 2] = (uint32_t)compress_stride<2>(b >> j);
    } else {
        if (j < 4) reinterpret_cast<uint16_t *>(cw + j)[g % 4] = (uint16_t)compress_stride<4>(b >> j);
    }
}

/* The same gather for parents of up to kGatherParentWords mask words per column (4096 frames at V = 4) and up to
 * 2 * kCompactCapacity slots: a block takes 64 columns, stages their parent words (contiguous per tile) and the decoded slot
 * table in LDS, and each thread assembles whole child words from there.  One wave per column and 64 slots (above) reads one
 * scattered 8-byte word per lane through the L2: 66 M requests, 150-200 us for the 1024-frame child of the headline code. */
constexpr int kGatherParentWords = 64;
template <int V> __global__ __launch_bounds__(kBlock) void compact_hard_lds_kernel(const uint64_t *__restrict__ parent, uint64_t *__restrict__ child,
                                                                                   const int32_t *__restrict__ map, int32_t count, int32_t N,
                                                                                   int cv, int ptiles, int cwords)
{
    constexpr int F = 64 * V;
    constexpr int COLS = 64, TPC = kBlock / COLS;           /* columns per block, threads per column (one wave each) */
    __shared__ uint64_t pw[kGatherParentWords][COLS];
    __shared__ int32_t sbit[2 * kCompactCapacity];          /* slot j -> (parent word of the column) << 8 | bit */
    const int n0 = blockIdx.x * COLS;
    for (int idx = threadIdx.x; idx < ptiles * COLS * V; idx += kBlock) {
        const int tile = idx / (COLS * V), rem = idx % (COLS * V), c = rem / V, v = rem % V;
        pw[tile * V + v][c] = n0 + c < N ? parent[((size_t)tile * N + n0) * V + rem] : 0ull;
    }
    for (int j = threadIdx.x; j < count; j += kBlock) {
        const int f = map[j], fi = f % F;
        sbit[j] = (((f / F) * V + fi % V) << 8) | (fi / V);
    }
    __syncthreads();
    const int c = threadIdx.x % COLS, q = threadIdx.x / COLS, n = n0 + c;
    if (n >= N) return;
    const int cf = 64 * cv;
    for (int cwd = q; cwd < cwords; cwd += TPC) {
        const int ctile = cwd / cv, cword = cwd % cv;       /* per child word, not per bit */
        uint64_t val = 0;
        for (int b = 0; b < 64; ++b) {
            const int j = ctile * cf + b * cv + cword;      /* the slot whose bit is bit b of this word */
            if (j >= count) break;
            const int sb = sbit[j];
            val |= ((pw[sb >> 8][c] >> (sb & 255)) & 1ull) << b;
        }
        child[((size_t)ctile * N + n) * cv + cword] = val;
    }
}

/* The way back (for any number of frames; one atomic per bit, as rounds 2 and early 3 did it, is 44 M device
 * atomics for 681 frames of the rate-9/10 code): every parent word is rewritten by ONE thread that looks up, for
 * the bits of its frames that were handed over, the child's bit: inv[frame] = where that bit sits in the child,
 * moved[(tile, v)] = which bits of that word moved (both filled by compact_inverse_kernel). */
template <int V> __global__ void compact_inverse_kernel(const int32_t *__restrict__ map, int32_t count, int32_t *__restrict__ inv,
                                                        unsigned long long *__restrict__ moved, int cv)
{
    /* inv[f] = where slot j's bit sits in the child, decoded once here: (child word of the column) << 8 | bit
     * (compact_hard_back_kernel did the four run-time divisions per moved frame and column: 79 us of integer arithmetic) */
    constexpr int F = 64 * V;
    const int j = blockIdx.x * blockDim.x + threadIdx.x;
    if (j >= count) return;
    const int f = map[j], fi = f % F;
    const int cf = 64 * cv, cfi = j % cf;
    inv[f] = (((j / cf) * cv + cfi % cv) << 8) | (cfi / cv);
    atomicOr(&moved[(size_t)(f / F) * V + fi % V], 1ull << (fi / V));
}

constexpr int kBackWords = 16;          /* child mask words per column a block keeps: 1024 frames */
template <int V> __global__ __launch_bounds__(kBlock) void compact_hard_back_kernel(uint64_t *__restrict__ parent, const uint64_t *__restrict__ child,
                                                                                    const int32_t *__restrict__ inv,
                                                                                    const unsigned long long *__restrict__ moved, int32_t N,
                                                                                    int cv, int cwords)
{
    /* cwords = child tiles in use x cv <= kBackWords: the column's child words go to LDS once (thread-private
     * entries, read back by a run-time index) -- one load per moved frame instead read 8 of every 32 fetched bytes,
     * 1.4 GB through the L2 for 681 frames: 79 us */
    constexpr int F = 64 * V;
    __shared__ uint64_t cw[kBackWords][kBlock];
    __shared__ int32_t sinv[F];             /* the tile's slots: a global load per moved frame put its latency into every trip */
    const int n = blockIdx.x * kBlock + threadIdx.x;
    const int tile = blockIdx.y;
    bool any = false;
#pragma unroll
    for (int v = 0; v < V; ++v) any = any || moved[(size_t)tile * V + v] != 0;
    if (!any) return;                       /* block-uniform */
    for (int k = threadIdx.x; k < F; k += kBlock) sinv[k] = inv[tile * F + k];
    __syncthreads();
    if (n >= N) return;
    for (int w = 0; w < cwords; ++w) cw[w][threadIdx.x] = child[((size_t)(w / cv) * N + n) * cv + w % cv];
#pragma unroll
    for (int v = 0; v < V; ++v) {
        uint64_t m = moved[(size_t)tile * V + v];
        if (!m) continue;
        uint64_t w = parent[((size_t)tile * N + n) * V + v];
        while (m) {
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            const int j = sinv[l * V + v];                  /* child word << 8 | bit (compact_inverse_kernel) */
            const uint64_t bit = (cw[j >> 8][threadIdx.x] >> (j & 255)) & 1ull;
            w = (w & ~(1ull << l)) | (bit << l);
        }
        parent[((size_t)tile * N + n) * V + v] = w;
    }
}

/* iteration counts and converged flags back; one thread per slot */
template <int V> __global__ void compact_finish_kernel(uint64_t *__restrict__ parent_done, int32_t *__restrict__ parent_iters,
                                                       const uint64_t *__restrict__ child_done, const int32_t *__restrict__ child_iters,
                                                       const int32_t *__restrict__ map, int32_t count, int cv)
{
    constexpr int F = 64 * V;
    const int cf = 64 * cv;
    const int jg = blockIdx.x * 64 + threadIdx.x;
    if (jg >= count) return;
    const int64_t f = map[jg];
    const int fi = (int)(f % F), cfi = jg % cf;
    parent_iters[f] = child_iters[jg];
    if ((child_done[(size_t)(jg / cf) * cv + cfi % cv] >> (cfi / cv)) & 1ull)
        atomicOr(reinterpret_cast<unsigned long long *>(parent_done) + (f / F) * V + fi % V, 1ull << (fi / V));
}

/* the child's bookkeeping when it takes over `count` running frames: slots beyond are padding.  One block of 64
 * lanes per child tile. */
template <int kUnused = 0>       /* a template only so that every translation unit may include this header */
__global__ void compact_child_state_kernel(uint64_t *__restrict__ done, int32_t *__restrict__ iters, int32_t count, int32_t max_iter, int cv)
{
    const int l = threadIdx.x, tile = blockIdx.x, cf = 64 * cv;
    for (int v = 0; v < cv; ++v) {
        const int slot = tile * cf + l * cv + v;
        iters[slot] = max_iter;
        const uint64_t pad = __ballot(slot >= count);
        if (l == 0) done[(size_t)tile * cv + v] = pad;
    }
}

/* ---- device-side tail: the same hand-over decided and carried out without the host ---------------
 * tail_gather_kernel runs after the state update of a round.  Every block reads the number of
 * frames still running after that round; unless 0 < running <= threshold (and at most a quarter
 * of the batch) nothing happens.  Otherwise each block lists the running frames (the same list in
 * every block: ascending mask word, ascending bit), and the blocks share the rows of Q, of the
 * channel array and of the hard-bit masks: the running frames' values go to consecutive slots of
 * the overflow tiles (same layout, same V).  The block that finishes last publishes the list,
 * marks every batch tile finished and the overflow slots in use, and sets state[0]: the next
 * launch's blocks follow it (tile_select).  tail_scatter_kernel brings bits, iteration counts and
 * converged flags back before packing. */
struct TailArgs {
    int32_t *state;                     /* [0] handed, [1] count, [2] round, [3] ticket */
    int32_t *map;                       /* [capacity] frame index of each overflow slot */
    const int32_t *running;             /* [max_iter + 2] frames still running after round i */
    uint64_t *done;                     /* [T + TO][V] */
    int32_t *iters;                     /* [T + TO][F] */
    void *Q;                            /* [T + TO][E][F] */
    void *chan;                         /* [T + TO][N][F] */
    uint64_t *hard;                     /* [T + TO][N][V] */
    int64_t E;
    int64_t frames;
    int32_t N;
    int32_t tiles;                      /* batch tiles in use by this call */
    int32_t base_tile;                  /* first overflow tile */
    int32_t capacity;                   /* overflow slots */
    int32_t threshold;
    int32_t iter;
    int32_t max_iter;
};

constexpr int kTailMaxWords = 4096;     /* mask words a block can list: 262144 frames */

template <int V, typename T>
__global__ __launch_bounds__(kBlock) void tail_gather_kernel(const TailArgs a)
{
    constexpr int F = 64 * V;
    __shared__ int32_t s_map[kCompactCapacity];
    __shared__ int32_t s_scan[kBlock];
    const int running = a.running[a.iter];
    if (a.state[0] || running <= 0 || running > a.threshold || (int64_t)running * 4 > a.frames) return;   /* uniform */
    const int words = a.tiles * V;
    /* list the running frames: thread t owns words t, t + 256, ... (ascending); exclusive scan of the
     * per-thread counts over consecutive word ranges keeps the order ascending in (word, bit) */
    const int per = (words + kBlock - 1) / kBlock;
    const int w0 = min((int)threadIdx.x * per, words), w1 = min(w0 + per, words);
    int mine = 0;
    for (int w = w0; w < w1; ++w) mine += __popcll(~a.done[w]);
    s_scan[threadIdx.x] = mine;
    __syncthreads();
    for (int off = 1; off < kBlock; off <<= 1) {
        const int v = threadIdx.x >= off ? s_scan[threadIdx.x - off] : 0;
        __syncthreads();
        s_scan[threadIdx.x] += v;
        __syncthreads();
    }
    int slot = s_scan[threadIdx.x] - mine;
    const int count = s_scan[kBlock - 1];
    for (int w = w0; w < w1; ++w) {
        uint64_t m = ~a.done[w];
        const int tile = w / V, v = w % V;
        while (m) {
            const int l = __ffsll((unsigned long long)m) - 1;
            m &= m - 1;
            if (slot < a.capacity) s_map[slot] = tile * F + l * V + v;      /* bit l of word v = frame V*l+v */
            ++slot;
        }
    }
    __syncthreads();
    if (count > a.capacity || count != running) return;      /* cannot happen: both come from the same masks */
    const int ctiles = (count + F - 1) / F;
    /* rows of Q (E), of the channel array (N) and of the hard masks (N), dealt over the blocks */
    const int64_t total_rows = a.E + 2 * (int64_t)a.N;
    for (int64_t row = blockIdx.x; row < total_rows; row += gridDim.x) {
        if (row < a.E + a.N) {
            const bool isq = row < a.E;
            const int64_t i = isq ? row : row - a.E, rows = isq ? a.E : a.N;
            T *arr = static_cast<T *>(isq ? a.Q : a.chan);
            for (int j = threadIdx.x; j < ctiles * F; j += kBlock) {
                T val = (T)0;
                if (j < count) {
                    const int64_t f = s_map[j];
                    val = arr[((f / F) * rows + i) * F + (f % F)];
                }
                arr[((size_t)(a.base_tile + j / F) * rows + i) * F + (j % F)] = val;
            }
        } else {
            const int64_t n = row - a.E - a.N;
            /* wave-wise: lane l of pass (ct, v) holds slot ct*F + V*l + v = bit l of word v */
            const int lane = threadIdx.x & 63;
            for (int unit = threadIdx.x >> 6; unit < ctiles * V; unit += kWavesPerBlock) {
                const int ct = unit / V, v = unit % V;
                const int j = ct * F + V * lane + v;
                bool bit = false;
                if (j < count) {
                    const int64_t f = s_map[j];
                    const int fi = (int)(f % F);
                    bit = (a.hard[((f / F) * a.N + n) * V + fi % V] >> (fi / V)) & 1ull;
                }
                const uint64_t w = __ballot(bit);
                if (lane == 0) a.hard[((size_t)(a.base_tile + ct) * a.N + n) * V + v] = w;
            }
        }
    }
    /* the last block to finish commits the hand-over */
    __threadfence();
    __shared__ int s_last;
    if (threadIdx.x == 0) s_last = (atomicAdd(&a.state[3], 1) == (int)gridDim.x - 1);
    __syncthreads();
    if (!s_last) return;
    __threadfence();
    for (int j = threadIdx.x; j < count; j += kBlock) a.map[j] = s_map[j];
    for (int w = threadIdx.x; w < words; w += kBlock) a.done[w] = ~0ull;
    const int TOv = (a.capacity / F) * V;
    for (int w = threadIdx.x; w < TOv; w += kBlock) {
        /* overflow word (ct, v): bit l in use iff slot ct*F + V*l + v < count */
        const int ct = w / V, v = w % V;
        uint64_t used = 0;
        for (int l = 0; l < 64; ++l)
            if (ct * F + V * l + v < count) used |= 1ull << l;
        a.done[(size_t)a.base_tile * V + w] = ~used;
    }
    for (int j = threadIdx.x; j < a.capacity; j += kBlock) a.iters[(size_t)a.base_tile * F + j] = a.max_iter;
    __threadfence();
    if (threadIdx.x == 0) { a.state[1] = count; a.state[2] = a.iter; a.state[0] = 1; }
}

/* bits, iteration counts and converged flags of the handed-over frames back into the batch's tiles */
template <int V> __global__ __launch_bounds__(kBlock) void tail_scatter_kernel(const TailArgs a)
{
    constexpr int F = 64 * V;
    if (!a.state[0]) return;
    const int count = a.state[1];
    const int64_t n0 = (int64_t)blockIdx.x * kWavesPerBlock + wave_id_in_block();
    const int lane = threadIdx.x & 63;
    if (blockIdx.x == 0) {
        for (int j = threadIdx.x; j < count; j += kBlock) {
            const int64_t f = a.map[j];
            const int fi = (int)(f % F), oj = j % F;
            a.iters[f] = a.iters[(size_t)a.base_tile * F + j];
            const bool conv = (a.done[(size_t)(a.base_tile + j / F) * V + oj % V] >> (oj / V)) & 1ull;
            /* the batch's word says "finished" for every frame since the hand-over: clear it for a
             * frame that never reached a clean syndrome */
            if (!conv) atomicAnd(reinterpret_cast<unsigned long long *>(a.done) + (f / F) * V + fi % V, ~(1ull << (fi / V)));
        }
    }
    for (int64_t n = n0; n < a.N; n += (int64_t)gridDim.x * kWavesPerBlock) {
        for (int j = lane; j < count; j += 64) {
            const int64_t f = a.map[j];
            const int fi = (int)(f % F), oj = j % F;
            const bool bit = (a.hard[((size_t)(a.base_tile + j / F) * a.N + n) * V + oj % V] >> (oj / V)) & 1ull;
            unsigned long long *word = reinterpret_cast<unsigned long long *>(a.hard) + ((f / F) * a.N + n) * V + fi % V;
            if (bit) atomicOr(word, 1ull << (fi / V));
            else atomicAnd(word, ~(1ull << (fi / V)));
        }
    }
}

struct PackArgs {
    const uint64_t *__restrict__ hard;  /* [T][N][V] */
    uint8_t *__restrict__ out;
    const int32_t *__restrict__ iters_tile; /* [T][F] */
    int32_t *__restrict__ iters_out;    /* [frames] or nullptr */
    int64_t frames;
    int64_t out_bytes;
    int32_t N, K;
    int32_t pack_mode;
};

/* toChar, decodeCL.c:188-199: byte j of frame b = hard bits 8j..8j+7, LSB first,
 * at (b*K)/8 + j.  pack_mode 1 = decodeCPU's bit packing at bit b*K+i
 * (MyLdpc.cpp:765-774).
 * Bytes mode: a 64 x 64 bit transpose per wave.  One wave = 64 consecutive output bytes (lanes along
 * j) of the 64 frames of one tile slice v: each lane fetches its 8 hard words ONCE (64 frames' bits
 * of columns 8j..8j+7) and emits one byte per frame, 64 contiguous bytes per store.  (One thread
 * per output byte, the first version, re-read every hard word once per frame of the tile: 0.41 ms
 * for the 30 MB of a 4096-frame rate-9/10 batch.)  Grid: pack_grid<V>(). */
template <int V> inline dim3 pack_grid(int32_t K, int tiles)
{
    const int chunks = K / 8 > 0 ? (K / 8 + 63) / 64 : 1;
    return dim3((unsigned)((chunks * V + kWavesPerBlock - 1) / kWavesPerBlock), (unsigned)tiles);
}

template <int V> __global__ __launch_bounds__(kBlock) void pack_kernel(const PackArgs a)
{
    constexpr int F = 64 * V;
    if (a.pack_mode == 0) {
        const int tile = blockIdx.y;
        const int kb = a.K / 8;
        if (blockIdx.x == 0 && a.iters_out) {
            for (int f = threadIdx.x; f < F; f += kBlock) {
                const int64_t frame = (int64_t)tile * F + f;        /* tile-major index == frame index */
                if (frame < a.frames) a.iters_out[frame] = a.iters_tile[frame];
            }
        }
        const int unit = (int)blockIdx.x * kWavesPerBlock + wave_id_in_block();
        const int v = unit % V;
        const int j = (unit / V) * 64 + (threadIdx.x & 63);
        if (j >= kb) return;
        const uint64_t *h = a.hard + ((size_t)tile * a.N + (size_t)j * 8) * V + v;
        uint64_t w[8];
#pragma unroll
        for (int b = 0; b < 8; ++b) w[b] = h[(size_t)b * V];
        for (int l = 0; l < 64; ++l) {
            const int64_t frame = (int64_t)tile * F + (int64_t)l * V + v;
            if (frame >= a.frames) break;
            unsigned byte = 0;
#pragma unroll
            for (int b = 0; b < 8; ++b) byte |= (unsigned)((w[b] >> l) & 1ull) << b;
            const int64_t off = frame * (int64_t)a.K / 8 + j;
            if (off < a.out_bytes) a.out[off] = (uint8_t)byte;
        }
    } else {
        const int64_t ob = (int64_t)blockIdx.x * kBlock + threadIdx.x;
        if (ob < a.frames && a.iters_out) a.iters_out[ob] = a.iters_tile[ob];
        if (ob >= a.out_bytes) return;
        unsigned byte = 0;
        for (int b = 0; b < 8; ++b) {
            const int64_t bit = ob * 8 + b;
            const int64_t frame = bit / a.K;
            if (frame >= a.frames) break;
            const int i = (int)(bit % a.K);
            const int64_t tile = frame / F;
            const int fi = (int)(frame % F);
            const uint64_t w = a.hard[((size_t)tile * a.N + i) * V + (fi % V)];
            byte |= (unsigned)((w >> (fi / V)) & 1ull) << b;
        }
        a.out[ob] = (uint8_t)byte;
    }
}

}  // namespace ldpc
