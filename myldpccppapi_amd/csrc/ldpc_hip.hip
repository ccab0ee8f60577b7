/*
 * ldpc_hip.hip -- host side of libldpc_hip.so (C ABI in include/ldpc_hip.h).
 *
 * Owns the HBM-resident decoder state, builds the per-degree work lists from
 * the edge list, and drives the flooding rounds the way the reference's host
 * loops do (decodeOnceSP MyLdpc.cpp:977-1059, decodeOnceMS :786-848) -- minus
 * the per-kernel queue.finish() and the blocking flags read-back every
 * iteration (:1024-1034): frames freeze on the device (state_kernel) and the
 * host only polls an "anything still running" word every poll_interval rounds.
 *
 * Round i (1-based), for every tile of F = 64*V frames:
 *   check_i    : R_i = check(Q_{i-1})
 *   var_i      : bits_i = hard(R_i) (frozen frames keep theirs); Q_i = var(R_i)
 *   syndrome_i : fail_i = any parity check of bits_i odd          } early_term only
 *   state_i    : frames with clean bits_i freeze, iters = i        } (and after the last round)
 *   tail_i     : (asynchronous callers) hand the last running frames over to the overflow tiles
 * then pack.
 *
 * Also here: decoders over several devices (ldpc_decoder_create_multi: one host thread per device
 * range), the host-buffer path and its staging rule (decode_host), the creation-time choice of the
 * column-fused check kernel's form (calibrate_link) and the launch plan (plan_launches).  The
 * kernels themselves are instantiated in other translation units: flood_sp / flood_ms / flood_ms16
 * (flood_tables.hpp) and engine_ldsp / engine_fused / engine_layered (engines.hpp).
 */
#include <hip/hip_runtime.h>

#include <algorithm>
#include <chrono>
#include <cstdarg>
#include <cstdio>
#include <cstdlib>
#include <cstring>
#include <map>
#include <memory>
#include <new>
#include <string>
#include <thread>
#include <vector>

#include "../../include/ldpc_hip.h"
#include "flood_kernels.hpp"
#include "flood_tables.hpp"
#include "layered_kernels.hpp"
#include "fused_kernels.hpp"
#include "ldsp_kernels.hpp"
#include "engines.hpp"
#include "channel_kernels.hpp"
#include "tune.hpp"
#include "host_stage.hpp"

#ifndef LDPC_IDLE_FAT
#define LDPC_IDLE_FAT 8
#endif

namespace {

thread_local std::string g_err;

int fail(int code, const char *fmt, ...)
{
    char buf[512];
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(buf, sizeof buf, fmt, ap);
    va_end(ap);
    g_err = buf;
    return code;
}

#define HIP_TRY(expr)                                                                  \
    do {                                                                               \
        hipError_t e_ = (expr);                                                        \
        if (e_ != hipSuccess)                                                          \
            return fail(LDPC_ERR_HIP, "%s failed: %s (%s:%d)", #expr, hipGetErrorString(e_), \
                        __FILE__, __LINE__);                                           \
    } while (0)

template <typename T> struct DevBuf {
    T *p = nullptr;
    size_t n = 0;
    DevBuf() = default;
    DevBuf(const DevBuf &) = delete;
    DevBuf &operator=(const DevBuf &) = delete;
    DevBuf(DevBuf &&o) noexcept : p(o.p), n(o.n) { o.p = nullptr; o.n = 0; }
    DevBuf &operator=(DevBuf &&o) noexcept
    {
        if (this != &o) { release(); p = o.p; n = o.n; o.p = nullptr; o.n = 0; }
        return *this;
    }
    hipError_t alloc(size_t count)
    {
        release();
        n = count;
        if (!count) return hipSuccess;
        return hipMalloc((void **)&p, count * sizeof(T));
    }
    hipError_t upload(const std::vector<T> &h)
    {
        hipError_t e = alloc(h.size());
        if (e != hipSuccess || h.empty()) return e;
        return hipMemcpy(p, h.data(), h.size() * sizeof(T), hipMemcpyHostToDevice);
    }
    void release()
    {
        if (p) (void)hipFree(p);
        p = nullptr;
        n = 0;
    }
    ~DevBuf() { release(); }
};

}  // namespace

/* ------------------------------------------------------------------ graph */

struct ldpc_graph {
    int32_t M = 0, N = 0;
    int64_t E = 0;
    std::vector<int32_t> rows, cols;        /* [E] hRows, hCols           */
    std::vector<int32_t> row_ptr;           /* [M+1] hRowRange            */
    std::vector<int32_t> col_ptr, col_edge; /* CSC, edges ascending       */
    int32_t max_row_deg = 0, max_col_deg = 0;
};

/* ---------------------------------------------------------------- decoder */

namespace {

using ldpc::CheckFn;
using ldpc::VarFn;
using ldpc::LinkFn;
using ldpc::CheckGroupFn;
using ldpc::VarGroupFn;
using ldpc::InitFn;
using ldpc::kVarBuckets;
using ldpc::kCheckBuckets;
using ldpc::kVarBucketLo;
using ldpc::kVarBucketHi;
using ldpc::kCheckBucketLo;
using ldpc::kCheckBucketHi;

struct RowClass {
    int degree = 0;
    int count = 0;
    DevBuf<int32_t> e0;
    std::vector<int32_t> h_e0;
    /* column-local fusion (check_link_kernel): per list row, the degree-2 column shared with
     * the next list row when both fall in one wave's chunk of link_rpw rows */
    DevBuf<int32_t> link_col, link_pos;
    int linked = 0;          /* number of fused columns */
    /* guided chunks (flood_kernels.hpp: LinkArgs): n_big chunks of link_rpw rows, then chunks of small_rows */
    int n_big = 0, small_rows = 1;
};
struct ColClass {
    int degree = 0;
    int count = 0;
    DevBuf<int32_t> col, edge;
    int64_t q_base = -1;     /* first Q slot of the class when Q is stored in writer order (VarArgs::q_base) */
};

/* the classes of one degree bucket that share a launch */
struct ClassGroup {
    int bucket = 0, lo = 0, hi = 0;
    int blocks = 0;                       /* gridDim.x */
    int blocks_fat = 0;                   /* ... with kIdleFat times the rows / columns per wave */
    std::vector<int> members;             /* indices into row_classes / col_classes */
    DevBuf<ldpc::GroupClass> table, table_fat;
};

/* rounds that are probably idle (the idle hint of run_flooding) launch this many times the rows / columns per wave */
constexpr int kIdleFat = LDPC_IDLE_FAT;

struct TimedSpan {
    hipEvent_t a, b;
    int kind;       /* 0 check, 1 var, 2 layer, 3 other, 4 check with column-local fusion, 5 check group, 6 var group */
    int degree;     /* groups: the bucket's highest degree */
    int64_t bytes;  /* algorithmic bytes of the launch in the two-kernel formulation (16 E + 4 N in total) */
    int64_t moved;  /* bytes this kernel's own loads and stores move (less when columns are fused in) */
    int lo;         /* groups: the bucket's lowest degree */
};

}  // namespace

struct ldpc_decoder {
    ldpc_decoder_config cfg{};
    int32_t M = 0, N = 0;
    int64_t E = 0;
    int V = 1, F = 64, T = 0; /* frames per lane, per tile, tiles at max_batch */
    std::vector<int32_t> h_col_ptr, h_col_edge, h_rows, h_cols;

    DevBuf<int32_t> row_ptr, edge_col, col_ptr, col_edge;
    /* Q in writer order (CheckArgs::qpos): slot of every edge, and col_edge with slots in place of edge ids (init_kernel) */
    DevBuf<int32_t> qpos, col_qedge;
    std::vector<int32_t> h_qpos;
    DevBuf<uint8_t> chan, Q, R;         /* message arrays: msg_size bytes per element */
    int msg_size = 4;                   /* 4 = fp32, 2 = fp16 (LDPC_MSG_F16) */
    InitFn init_fn = nullptr;
    DevBuf<uint64_t> hard, failw, done;
    DevBuf<int32_t> iters, active;
    std::vector<RowClass> row_classes;
    std::vector<ColClass> col_classes;
    /* launch plan of a round (plan_launches): classes that share a launch, classes launched alone,
     * and the left-over rows a linked check launch takes along */
    std::vector<ClassGroup> check_groups, var_groups;
    std::vector<int> check_solo, var_solo;
    CheckGroupFn check_group_fn[kCheckBuckets] = {};
    int check_group_width = 1;          /* values per lane of the group check kernels (2 for fp16 messages at V = 4) */
    VarGroupFn var_group_fn[kVarBuckets] = {};
    DevBuf<int32_t> extra_e0, extra_deg;
    int n_extra = 0;
    int64_t extra_edges = 0;
    CheckFn check_fn[ldpc::kMaxUnrolledCheckDegreeMS + 1] = {};      /* narrow waves (1 value per lane) */
    CheckFn check_fn_wide[ldpc::kMaxUnrolledCheckDegreeMS + 1] = {}; /* V values per lane */
    int max_check_unrolled = ldpc::kMaxUnrolledDegree;
    LinkFn link_fn[ldpc::kMaxUnrolledDegree + 1] = {};        /* wide waves */
    LinkFn link_narrow_fn[ldpc::kMaxUnrolledDegree + 1] = {}; /* narrow waves */
    LinkFn link_deep_fn[ldpc::kMaxUnrolledDegree + 1] = {};   /* narrow waves, inputs two rows ahead */
    LinkFn link_half_fn[ldpc::kMaxUnrolledDegree + 1] = {};   /* 2 values per lane (V = 4) */
    int tune_link_deep = 0;
    ldpc::Tune tune;                    /* cfg.tune_* unpacked (tune.hpp) */
    int tune_link_narrow = 1;           /* linked check kernel: 0 wide (V values per lane), 1 narrow (1), 2 half (2) */
    bool link_calibrated = false;       /* chosen by timing the candidates at creation */
    float link_cal_ms[3] = {0, 0, 0};   /* what the calibration measured per launch: [0] wide, [1] narrow, [2] half */
    /* placement search (cfg.tune_place): the column-fused check kernel's time on each candidate set of arrays */
    int place_candidates = 0, place_kept = 0;
    float place_ms[16] = {};            /* the original pair, then up to 7 fresh R and 7 fresh Q allocations */
    int link_rpw = 16;                  /* rows per wave of the fused check kernel; 0 = fusion off */
    int tune_link_guided = 0;           /* tri-state: shorter row chunks at the end of the fused check launch */
    int tune_tiles_first = 0;           /* tri-state: flooding launches as (tiles, blocks) grids (flood_grid) */
    int cus = 256;                      /* compute units of the device */
    VarFn var_fn[ldpc::kMaxUnrolledDegree + 1] = {};

    ldpc::LayeredPlan layered;          /* LDPC_ALGO_LAYERED, streaming (one launch per layer) */
    ldpc::FusedPlan fused;              /* LDPC_ALGO_LAYERED, short QC codes: whole decode in LDS */
    bool use_fused = false;
    ldpc::LdspPlan ldsp;                /* LDPC_ALGO_LAYERED, mid-size QC codes: posterior in LDS, check records in cache */
    bool use_ldsp = false;

    /* staging for the host-buffer entry point: three slots, so the H2D copy of group k+1
     * (copy_stream) overlaps the decode of group k (stream) and the copy-out of group k-1 */
    hipStream_t stream = nullptr, copy_stream = nullptr;
    struct HostSlot {
        DevBuf<float> llr;
        DevBuf<uint8_t> out;
        DevBuf<int32_t> iters;
        uint8_t *h_out = nullptr;       /* pinned: D2H completes without blocking the host */
        int32_t *h_iters = nullptr;
        uint8_t *h_head = nullptr;      /* pinned, kStageBytes: a whole small group, or a large group's bytes before
                                           its first page boundary and (last group) after its last one */
        hipEvent_t h2d_done = nullptr, all_done = nullptr;
        bool busy = false;
        int64_t off = 0, n = 0, dst = 0, copy_bytes = 0;
        /* the group's counts for the call's statistics: the decoder's summary words (and those of the decoders its
         * stragglers were handed to), copied out behind the group's decode, before the next group resets them */
        int32_t *h_sum = nullptr;       /* pinned: [4][4] */
        int g_iterations = 0, g_tiles = 0, g_children = 0, g_child_f[3] = {0, 0, 0};
    } slot[3];
    /* counts of the last ldpc_decode() call over ALL its launch groups (ldpc_decoder_stats) */
    struct CallCounts { bool valid = false; int32_t iterations = 0, batch_time = 0; int64_t frames = 0, converged = 0, frame_rounds = 0; } call;
    bool suppress_poll = false;
    int32_t *h_active = nullptr;        /* pinned */
    /* LDPC_HOST_INPUT_STAGED: a ring of pinned chunks the caller's channel values pass through, filled by
     * the stager thread (and its copy helpers) while the calling thread enqueues -- or, with polling,
     * sits in -- the previous group's decode.  Threads and ring are made by the first large host-buffer
     * call and live until the handle is destroyed. */
    struct RingChunk { uint8_t *h = nullptr; hipEvent_t ev = nullptr; bool used = false; };
    std::vector<RingChunk> ring;
    size_t ring_next = 0;
    std::unique_ptr<ldpc::Worker> stager;
    std::vector<std::unique_ptr<ldpc::Worker>> copy_helpers;
    std::vector<ldpc::Job> copy_jobs;   /* one per helper, reused chunk after chunk */
    ldpc::Job stage_job[3];             /* one per slot */
    bool stage_pending[3] = {false, false, false};
    /* LDPC_HOST_INPUT_LOCK_PAGES: blocks of the caller's buffer this handle has page-locked (empty between
     * calls), and blocks it could not release (reported by the call and by ldpc_decoder_destroy) */
    std::vector<void *> locked_blocks, stuck_blocks;
    /* tail compaction (flood_kernels.hpp): a V = 1, one-tile decoder that takes over the last running
     * frames of a polled, early-terminating decode */
    ldpc_decoder *child = nullptr;
    DevBuf<int32_t> cmap;               /* [child_capacity] frame indices handed to the child */
    DevBuf<int32_t> cinv;               /* [max_batch] where a handed-over frame's bit sits in the child (valid where cmoved says so) */
    DevBuf<unsigned long long> cmoved;  /* [T][V] bits of each mask word whose frames were handed over */
    int child_capacity = ldpc::kCompactCapacity;      /* frames the child holds: 512, or 1024 for batches of >= 4096 frames */
    int compact_threshold = ldpc::kCompactCapacity;   /* cfg.tune_compact: 0 = off, else hand over when <= this many frames run */
    bool is_child = false;
    /* device-side tail (flood_kernels.hpp: TailRef, tail_gather_kernel): TO overflow tiles follow the T
     * tiles of max_batch in every array (TA = T + TO allocated) */
    bool tail_enabled = false;
    int TO = 0, TA = 0;
    DevBuf<int32_t> tail_state, tail_map, running;
    /* Rounds beyond the previous call's iteration count are probably idle: they are launched with
     * kIdleFat times as many rows / columns per wave, i.e. that many times fewer workgroups (an idle
     * workgroup costs about a clock of dispatch chip-wide: 312 000 of them per round for the rate-9/10
     * code at 4096 frames).  The count arrives through a pinned word copied at the end of every call;
     * it is read only once that copy has completed.  (Block-strided loops inside the kernels were
     * tried instead and cost 17-27 % at full work: profiles/r02_ab_block_strided_negative.txt.) */
    int32_t *h_summary = nullptr;       /* pinned [2] */
    hipEvent_t ev_summary = nullptr;
    bool summary_pending = false;
    int idle_after = 0;                 /* 0: no hint */

    bool timing = false;                /* the call being enqueued is timed */
    int timing_every = 0;               /* 0 off, k: every k-th device call is timed */
    int64_t timing_calls = 0;
    std::vector<TimedSpan> spans;
    size_t spans_used = 0;
    hipEvent_t ev_begin = nullptr, ev_end = nullptr;
    hipStream_t last_stream = nullptr;
    bool have_last = false;
    int32_t tap_iter = 0;
    int tune_rpw = 0, tune_cpw = 0;     /* rows / columns per wave (0 = automatic) */
    int tune_syn_xcd = 1;               /* 0: plain 2-D syndrome grid */
    int tune_check_wide = 0;            /* 1: check kernels move V floats per lane */
    int32_t last_iterations = 0;
    int64_t last_frames = 0;
    DevBuf<int32_t> summary;            /* [4]: max iters, converged count, tile-rounds that did work (early termination) */
    ldpc_decoder *handed_to = nullptr;  /* the decoder (child, or the child's child) that finished the last call's stragglers */
    bool first_round_from_chan = false; /* this call's round 1 reads q = y from the channel array (min-sum) */
    int32_t last_tiles = 0;

    /* a handle over several devices (ldpc_decoder_create_multi): one single-device decoder per entry
     * of the device list, and one persistent host thread per entry that runs its frame range; this
     * object then owns no device state of its own */
    std::vector<ldpc_decoder *> shards;
    std::vector<std::unique_ptr<ldpc::Worker>> shard_workers;

    ~ldpc_decoder()
    {
        /* threads first: nothing of this handle runs any more when its streams and buffers go */
        for (auto &w : shard_workers) w->stop();
        if (stager) stager->stop();
        for (auto &w : copy_helpers) w->stop();
        for (ldpc_decoder *sh : shards) (void)ldpc_decoder_destroy(sh);
        for (auto &s : spans) { (void)hipEventDestroy(s.a); (void)hipEventDestroy(s.b); }
        if (ev_begin) (void)hipEventDestroy(ev_begin);
        if (ev_end) (void)hipEventDestroy(ev_end);
        if (h_active) (void)hipHostFree(h_active);
        for (auto &rc : ring) {
            if (rc.h) (void)hipHostFree(rc.h);
            if (rc.ev) (void)hipEventDestroy(rc.ev);
        }
        if (h_summary) (void)hipHostFree(h_summary);
        if (ev_summary) (void)hipEventDestroy(ev_summary);
        for (auto &sl : slot) {
            if (sl.h_out) (void)hipHostFree(sl.h_out);
            if (sl.h_iters) (void)hipHostFree(sl.h_iters);
            if (sl.h_head) (void)hipHostFree(sl.h_head);
            if (sl.h_sum) (void)hipHostFree(sl.h_sum);
            if (sl.h2d_done) (void)hipEventDestroy(sl.h2d_done);
            if (sl.all_done) (void)hipEventDestroy(sl.all_done);
        }
        if (copy_stream) (void)hipStreamDestroy(copy_stream);
        if (stream) (void)hipStreamDestroy(stream);
        delete child;
    }
};

namespace {

/* a plain float4 copy (the measurement aid ldpc_hbm_probe_device; also the placement search's second opinion) */
template <bool NT>
__global__ __launch_bounds__(256) void hbm_probe_copy_kernel(const ldpc::vf4 *__restrict__ src, ldpc::vf4 *__restrict__ dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n4; i += stride) {
        ldpc::vf4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + (size_t)k * 256 < n4) v[k] = NT ? __builtin_nontemporal_load(&src[i + (size_t)k * 256]) : src[i + (size_t)k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k)
            if (i + (size_t)k * 256 < n4) {
                if (NT) __builtin_nontemporal_store(v[k], &dst[i + (size_t)k * 256]);
                else dst[i + (size_t)k * 256] = v[k];
            }
    }
}

/* summary[0] = max over frames of iters (the reference's `Time=`), summary[1] =
 * number of frames whose syndrome ended clean. */
template <int V>
__global__ void summary_kernel(const int32_t *iters, const uint64_t *done, const uint64_t *fail,
                               int64_t frames, int32_t freeze, int32_t *summary)
{
    constexpr int F = 64 * V;
    const int64_t f = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (f >= frames) return;
    const int64_t tile = f / F;
    const int fi = (int)(f % F);
    const int l = fi / V, v = fi % V;
    atomicMax(&summary[0], iters[f]);
    const uint64_t ok = freeze ? done[tile * V + v] : ~fail[tile * V + v];
    if ((ok >> l) & 1ull) atomicAdd(&summary[1], 1);
}

int pick_frames_per_lane(const ldpc_decoder_config &cfg, int32_t max_row_deg, int32_t max_col_deg)
{
    if (cfg.frames_per_lane) return cfg.frames_per_lane;
    /* Wide tiles (4 values = 16 B fp32 / 8 B fp16 per lane) once there are enough frames to
     * fill them and the register arrays of the unrolled kernels stay moderate.  Flooding check
     * kernels run in narrow waves, so only the column degree counts there; the layered kernel
     * holds a whole row (P and R) per lane. */
    int deg = cfg.algo == LDPC_ALGO_LAYERED ? max_row_deg : max_col_deg;
    if (cfg.algo == LDPC_ALGO_MS) deg = (deg + 1) / 2;   /* min-sum variable nodes need half the registers */
    if (cfg.max_batch >= 1024 && deg <= 8) return 4;
    if (cfg.max_batch >= 256 && deg <= ldpc::kMaxUnrolledDegree) return 2;
    return 1;
}

hipError_t span_begin(ldpc_decoder *d, hipStream_t s, int kind, int degree = 0, int64_t bytes = 0, int64_t moved = -1,
                      int lo = 0)
{
    if (!d->timing) return hipSuccess;
    if (d->spans_used == d->spans.size()) {
        TimedSpan t{};
        hipError_t e = hipEventCreate(&t.a);
        if (e != hipSuccess) return e;
        e = hipEventCreate(&t.b);
        if (e != hipSuccess) return e;
        d->spans.push_back(t);
    }
    d->spans[d->spans_used].kind = kind;
    d->spans[d->spans_used].degree = degree;
    d->spans[d->spans_used].bytes = bytes;
    d->spans[d->spans_used].moved = moved < 0 ? bytes : moved;
    d->spans[d->spans_used].lo = lo;
    return hipEventRecord(d->spans[d->spans_used].a, s);
}

hipError_t span_end(ldpc_decoder *d, hipStream_t s)
{
    if (!d->timing) return hipSuccess;
    return hipEventRecord(d->spans[d->spans_used++].b, s);
}

/* grid of a flooding launch: (tiles, blocks) -- flood_kernels.hpp: grid_pos() -- when the blocks fit gridDim.y */
static inline dim3 flood_grid(const ldpc_decoder *d, unsigned blocks, unsigned tiles, int32_t *tiles_first, bool linked = false)
{
    /* tune_tiles_first: 0 automatic = the column-fused check launch only (-8 % there; the variable-node
     * launches gather anyway and lose 4 %, the plain check launches of the rate-9/10 code 1.7 %), 1 all,
     * 2 none */
    const bool want = d->tune_tiles_first == 1 || (d->tune_tiles_first == 0 && linked);
    *tiles_first = (blocks <= 65535u && want) ? 1 : 0;
    return *tiles_first ? dim3(tiles, blocks) : dim3(blocks, tiles);
}

template <int V> int run_flooding(ldpc_decoder *d, const float *llr_dev, int64_t frames,
                                  uint8_t *out_dev, int64_t out_bytes, int32_t *iters_dev,
                                  hipStream_t s, int start_round = 1);

/* The check phase of round `it` (R_i = check(Q_{i-1})) over `tiles` tiles: the column-fused launch, the bucket
 * launches and the classes launched alone.  Also what the placement search times. */
template <int V> int enqueue_check_phase(ldpc_decoder *d, hipStream_t s, int tiles, int64_t frames, int it, int max_iter,
                                         bool fat, const ldpc::TailRef &tr)
{
    using namespace ldpc;
    const int64_t msz = d->msg_size;
    for (auto &rc : d->row_classes) {
        if (!rc.linked) continue;
        /* algorithmic bytes: the fused columns' messages and channel values count as in the
         * two-kernel formulation (16 E + 4 N per frame-iteration in total).
         * moved: every Q of the class and the fused columns' channel values in; R of the unfused
         * edges and the fused columns' new Q out.  Rows riding along: Q in, R out. */
        HIP_TRY(span_begin(d, s, 4, rc.degree,
                           (2 * msz * rc.degree * rc.count + msz * 5 * rc.linked + 2 * msz * d->extra_edges) * frames,
                           msz * ((int64_t)rc.degree * rc.count + rc.linked +
                                  ((int64_t)rc.degree * rc.count - 2 * rc.linked) +
                                  (it < max_iter ? 2 * rc.linked : 0) + 2 * d->extra_edges) * frames));
        CheckArgs a{d->Q.p, d->R.p, rc.e0.p, d->done.p, d->E, rc.count, 1, rc.degree, tr};
        a.qpos = d->qpos.p;
        LinkArgs lk{rc.link_col.p, rc.link_pos.p, d->chan.p, d->Q.p, d->hard.p, d->N,
                    (it < max_iter) ? 1 : 0, d->tap_iter ? 1 : 0, d->extra_e0.p, d->extra_deg.p, d->n_extra, 0,
                    rc.n_big, rc.small_rows};
        a.rows_per_wave = d->link_rpw;
        const int variant = (d->tune_link_narrow == 2 && !d->link_half_fn[rc.degree]) ? 1 : d->tune_link_narrow;
        const int waves = link_chunk_count(d->link_rpw, rc.n_big, rc.small_rows, rc.count) * (variant == 1 ? V : variant == 2 ? V / 2 : 1);
        lk.link_blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
        /* tiles vary fastest: the short chunks of all tiles are the launch's last blocks */
        const dim3 grid = flood_grid(d, lk.link_blocks + (d->n_extra + kWavesPerBlock - 1) / kWavesPerBlock, tiles, &a.tiles_first, true);
        (variant == 2 ? d->link_half_fn : variant == 1 ? (d->tune_link_deep ? d->link_deep_fn : d->link_narrow_fn) : d->link_fn)
            [rc.degree]<<<grid, kBlock, 0, s>>>(a, lk);
        HIP_TRY(span_end(d, s));
    }
    for (auto &g : d->check_groups) {
        int64_t edges = 0;
        for (int i : g.members) edges += (int64_t)d->row_classes[i].degree * d->row_classes[i].count;
        HIP_TRY(span_begin(d, s, 5, g.hi, 2 * msz * edges * frames, -1, g.lo));
        CheckArgs a{d->Q.p, d->R.p, nullptr, d->done.p, d->E, 0, (d->tune_rpw ? d->tune_rpw : 2) * (fat ? kIdleFat : 1), 0, tr};
        a.qpos = d->qpos.p;
        if (it == 1 && d->first_round_from_chan) { a.first_chan = d->chan.p; a.edge_col = d->edge_col.p; a.N = d->N; }
        const dim3 grid = flood_grid(d, fat ? g.blocks_fat : g.blocks, tiles, &a.tiles_first);
        d->check_group_fn[g.bucket]<<<grid, kBlock, 0, s>>>(a, fat ? g.table_fat.p : g.table.p, (int)g.members.size());
        HIP_TRY(span_end(d, s));
    }
    for (int ci : d->check_solo) {
        RowClass &rc = d->row_classes[ci];
        HIP_TRY(span_begin(d, s, 0, rc.degree, 2 * msz * rc.degree * rc.count * frames));
        CheckArgs a{d->Q.p, d->R.p, rc.e0.p, d->done.p, d->E, rc.count, 1, rc.degree, tr};
        a.qpos = d->qpos.p;
        if (it == 1 && d->first_round_from_chan) { a.first_chan = d->chan.p; a.edge_col = d->edge_col.p; a.N = d->N; }
        const int slotk = rc.degree <= d->max_check_unrolled ? rc.degree : 0;
        const bool narrow = slotk && (!d->tune_check_wide || rc.degree > kMaxUnrolledDegree);
        const int rpw = (d->tune_rpw ? d->tune_rpw : (narrow ? 2 : 1)) * (fat ? kIdleFat : 1);
        a.rows_per_wave = rpw;
        const int waves = ((rc.count + rpw - 1) / rpw) * (narrow ? V : 1);
        const dim3 grid = flood_grid(d, (waves + kWavesPerBlock - 1) / kWavesPerBlock, tiles, &a.tiles_first);
        (narrow ? d->check_fn : d->check_fn_wide)[slotk]<<<grid, kBlock, 0, s>>>(a);
        HIP_TRY(span_end(d, s));
    }
    return LDPC_OK;
}

/* The variable-node phase of round `it`: bits_i = hard(R_i); Q_i = var(R_i) unless this is the last round. */
template <int V> int enqueue_var_phase(ldpc_decoder *d, hipStream_t s, int tiles, int64_t frames, int it, int max_iter,
                                       bool fat, const ldpc::TailRef &tr)
{
    using namespace ldpc;
    const int64_t msz = d->msg_size;
    /* var_i: bits_i = hard(R_i); Q_i = var(R_i) unless this is the last round */
    const int wq = (it < max_iter) ? 1 : 0;
    for (auto &g : d->var_groups) {
        int64_t units = 0;          /* messages read + written + channel values read, per frame */
        for (int i : g.members) units += (int64_t)((wq ? 2 : 1) * d->col_classes[i].degree + 1) * d->col_classes[i].count;
        HIP_TRY(span_begin(d, s, 6, g.hi, msz * units * frames, -1, g.lo));
        VarArgs a{d->R.p, d->Q.p, d->chan.p, d->hard.p, d->done.p, nullptr, nullptr,
                  d->E, d->N, 0, (d->tune_cpw ? d->tune_cpw : 1) * (fat ? kIdleFat : 1), wq, 0, tr};
        const dim3 grid = flood_grid(d, fat ? g.blocks_fat : g.blocks, tiles, &a.tiles_first);
        d->var_group_fn[g.bucket]<<<grid, kBlock, 0, s>>>(a, fat ? g.table_fat.p : g.table.p, (int)g.members.size());
        HIP_TRY(span_end(d, s));
    }
    for (int ci : d->var_solo) {
        ColClass &cc = d->col_classes[ci];
        HIP_TRY(span_begin(d, s, 1, cc.degree, msz * ((wq ? 2 : 1) * cc.degree + 1) * cc.count * frames));
        VarArgs a{d->R.p, d->Q.p, d->chan.p, d->hard.p, d->done.p, cc.col.p, cc.edge.p,
                  d->E, d->N, cc.count, 1, wq, cc.degree, tr};
        const int cpw = (d->tune_cpw ? d->tune_cpw : 1) * (fat ? kIdleFat : 1);
        a.cols_per_wave = cpw;
        a.q_base = cc.q_base;
        const int slotk = cc.degree <= kMaxUnrolledDegree ? cc.degree : 0;
        const int waves = (cc.count + cpw - 1) / cpw;
        const dim3 grid = flood_grid(d, (waves + kWavesPerBlock - 1) / kWavesPerBlock, tiles, &a.tiles_first);
        d->var_fn[slotk]<<<grid, kBlock, 0, s>>>(a);
        HIP_TRY(span_end(d, s));
    }
    return LDPC_OK;
}

/* Hand the `count` frames that are still running after round `it` over to the child decoder, let it
 * finish them (rounds it+1 ...), and bring their bits, iteration counts and converged flags back. */
template <int V> int compact_and_finish(ldpc_decoder *d, int64_t frames, int count, int it, hipStream_t s)
{
    using namespace ldpc;
    /* the smallest decoder of the chain (1024 frames in tiles of 256 -> 512 in tiles of 64 -> one tile of 64) that holds them:
     * a tile of 256 frames for a dozen stragglers would cost four times the traffic per round, eight tiles of 64 with one
     * straggler each eight times that of one tile */
    ldpc_decoder *c = d->child;
    while (c->child && count <= c->child->cfg.max_batch) c = c->child;
    const int cv = c->V, cf = 64 * cv;                      /* its frames per lane and per tile */
    const unsigned ct = (unsigned)((count + cf - 1) / cf);  /* child tiles in use */
    const unsigned cg = ct * (unsigned)cv;                  /* ... in groups of 64 slots */
    if (cg > (unsigned)kBackWords) return fail(LDPC_ERR_STATE, "hand-over of %d frames: more than %d mask words per column", count, kBackWords);
    HIP_TRY(hipMemsetAsync(d->active.p, 0, sizeof(int32_t), s));
    compact_list_kernel<V><<<(unsigned)((frames + kBlock - 1) / kBlock), kBlock, 0, s>>>(d->done.p, frames, d->cmap.p,
                                                                                         d->active.p, d->child_capacity);
    const dim3 ge((unsigned)((d->E + kWavesPerBlock - 1) / kWavesPerBlock), cg);
    const dim3 gn((unsigned)((d->N + kWavesPerBlock - 1) / kWavesPerBlock), cg);
    /* many frames: one coalesced pass over the parent's rows through LDS; few: one sector per value */
    const bool rowwise = count >= 128;
    const int ptiles = (int)((frames + 64 * V - 1) / (64 * V));
    if (d->msg_size == 2) {
        if (rowwise) {
            compact_gather_rows_kernel<V, _Float16><<<(unsigned)((d->E + gather_rows_per_block<_Float16>() - 1) / gather_rows_per_block<_Float16>()), kBlock, 0, s>>>((const _Float16 *)d->Q.p, (_Float16 *)c->Q.p, d->cmap.p, count, d->E, ptiles, cf, d->qpos.p, c->qpos.p);
            compact_gather_rows_kernel<V, _Float16><<<(unsigned)((d->N + gather_rows_per_block<_Float16>() - 1) / gather_rows_per_block<_Float16>()), kBlock, 0, s>>>((const _Float16 *)d->chan.p, (_Float16 *)c->chan.p, d->cmap.p, count, d->N, ptiles, cf);
        } else {
            compact_gather_kernel<V, _Float16><<<ge, kBlock, 0, s>>>((const _Float16 *)d->Q.p, (_Float16 *)c->Q.p, d->cmap.p, count, d->E, cf, d->qpos.p, c->qpos.p);
            compact_gather_kernel<V, _Float16><<<gn, kBlock, 0, s>>>((const _Float16 *)d->chan.p, (_Float16 *)c->chan.p, d->cmap.p, count, d->N, cf);
        }
    } else {
        if (rowwise) {
            compact_gather_rows_kernel<V, float><<<(unsigned)((d->E + gather_rows_per_block<float>() - 1) / gather_rows_per_block<float>()), kBlock, 0, s>>>((const float *)d->Q.p, (float *)c->Q.p, d->cmap.p, count, d->E, ptiles, cf, d->qpos.p, c->qpos.p);
            compact_gather_rows_kernel<V, float><<<(unsigned)((d->N + gather_rows_per_block<float>() - 1) / gather_rows_per_block<float>()), kBlock, 0, s>>>((const float *)d->chan.p, (float *)c->chan.p, d->cmap.p, count, d->N, ptiles, cf);
        } else {
            compact_gather_kernel<V, float><<<ge, kBlock, 0, s>>>((const float *)d->Q.p, (float *)c->Q.p, d->cmap.p, count, d->E, cf, d->qpos.p, c->qpos.p);
            compact_gather_kernel<V, float><<<gn, kBlock, 0, s>>>((const float *)d->chan.p, (float *)c->chan.p, d->cmap.p, count, d->N, cf);
        }
    }
    /* the hard bits travel only where the next decision can depend on the previous one: the sum-product rule keeps the old
     * bit on a tie or a NaN (decodeCL.c:78-82); min-sum decides every bit anew in every round (bit = !(p > 0), :161-165) */
    HIP_TRY(hipMemsetAsync(c->hard.p, 0, (size_t)ct * d->N * cv * sizeof(uint64_t), s));
    if (d->cfg.algo == LDPC_ALGO_SP) {
        if (ptiles * V <= kGatherParentWords && count <= 2 * kCompactCapacity)
            compact_hard_lds_kernel<V><<<(unsigned)((d->N + 63) / 64), kBlock, 0, s>>>(d->hard.p, c->hard.p, d->cmap.p, count, d->N, cv, ptiles, (int)cg);
        else
            compact_hard_kernel<V><<<gn, kBlock, 0, s>>>(d->hard.p, c->hard.p, d->cmap.p, count, d->N, cv);
    }
    compact_child_state_kernel<0><<<ct, 64, 0, s>>>(c->done.p, c->iters.p, count, d->cfg.max_iter, cv);
    HIP_TRY(hipGetLastError());
    c->timing = false;
    c->tap_iter = 0;
    const int rc = cv == 1 ? run_flooding<1>(c, nullptr, count, nullptr, 0, nullptr, s, it + 1)
                           : cv == 2 ? run_flooding<2>(c, nullptr, count, nullptr, 0, nullptr, s, it + 1)
                                     : run_flooding<4>(c, nullptr, count, nullptr, 0, nullptr, s, it + 1);
    if (rc) return rc;
    d->handed_to = c;
    /* the bits back: every parent word collects its moved frames' bits (no atomics; the per-bit atomic scatter of
     * compact_hard_kernel took 53-80 us for a few dozen frames, this takes 10-30) */
    HIP_TRY(hipMemsetAsync(d->cmoved.p, 0, d->cmoved.n * sizeof(unsigned long long), s));
    compact_inverse_kernel<V><<<(unsigned)((count + 255) / 256), 256, 0, s>>>(d->cmap.p, count, d->cinv.p, d->cmoved.p, cv);
    compact_hard_back_kernel<V><<<dim3((unsigned)((d->N + kBlock - 1) / kBlock), (unsigned)ptiles), kBlock, 0, s>>>(
        d->hard.p, c->hard.p, d->cinv.p, d->cmoved.p, d->N, cv, (int)cg);
    compact_finish_kernel<V><<<(unsigned)((count + 63) / 64), 64, 0, s>>>(d->done.p, d->iters.p, c->done.p, c->iters.p, d->cmap.p, count, cv);
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

template <int V> int run_flooding(ldpc_decoder *d, const float *llr_dev, int64_t frames,
                                  uint8_t *out_dev, int64_t out_bytes, int32_t *iters_dev,
                                  hipStream_t s, int start_round)
{
    using namespace ldpc;
    const int F = 64 * V;
    const int tiles = (int)((frames + F - 1) / F);
    const int64_t msz = d->msg_size;
    const int max_iter = d->cfg.max_iter;
    const int rounds = d->tap_iter ? std::min(d->tap_iter, max_iter) : max_iter;
    const bool freeze = d->cfg.early_term != 0;
    const size_t slot = (size_t)d->TA * V;  /* words per fail slot */

    const bool resume = start_round > 1;    /* a child taking over running frames: their state is in place */
    HIP_TRY(hipMemsetAsync(d->failw.p, 0, d->failw.n * sizeof(uint64_t), s));
    HIP_TRY(hipMemsetAsync(d->summary.p, 0, 4 * sizeof(int32_t), s));
    d->handed_to = nullptr;
    /* idle hint from the previous call (asynchronous early termination only) */
    if (!resume && d->summary_pending) {
        if (hipEventQuery(d->ev_summary) == hipSuccess) {
            d->summary_pending = false;
            d->idle_after = (d->h_summary[0] > 0 && d->h_summary[0] < max_iter) ? d->h_summary[0] + 1 : 0;
        } else {
            (void)hipGetLastError();        /* "not ready" is not an error of this call */
        }
    }
    const int idle_after = (freeze && !resume && !d->tap_iter && d->cfg.poll_interval == 0) ? d->idle_after : 0;
    /* device-side tail: only when the call has clearly more tiles than the overflow area */
    const bool use_tail = d->tail_enabled && freeze && !resume && !d->tap_iter && tiles >= 4 * d->TO;
    const TailRef tr{use_tail ? d->tail_state.p : nullptr, d->T, d->TO};
    TailArgs ta{};
    if (use_tail) {
        HIP_TRY(hipMemsetAsync(d->tail_state.p, 0, 4 * sizeof(int32_t), s));
        HIP_TRY(hipMemsetAsync(d->running.p, 0, d->running.n * sizeof(int32_t), s));
        ta = TailArgs{d->tail_state.p, d->tail_map.p, d->running.p, d->done.p, d->iters.p, d->Q.p, d->chan.p, d->hard.p,
                      d->E, frames, d->N, tiles, d->T, d->TO * F, std::min(d->compact_threshold, d->TO * F), 0, max_iter};
    }

    /* min-sum whose check phase is made of the unrolled bucket / single-class kernels only: round 1 reads q = y from the
     * channel array and the input transpose writes no Q (CheckArgs::first_chan); not with a debug tap */
    bool q_less = d->cfg.algo == LDPC_ALGO_MS && !resume && !d->tap_iter && max_iter > 1 && d->n_extra == 0;
    for (auto &rc : d->row_classes) if (rc.linked) q_less = false;
    for (int ci : d->check_solo) if (d->row_classes[ci].degree > d->max_check_unrolled) q_less = false;
    d->first_round_from_chan = q_less;
    if (!resume) {
        HIP_TRY(span_begin(d, s, 3));
        InitArgs a{llr_dev, d->chan.p, q_less ? nullptr : d->Q.p, d->hard.p, d->col_ptr.p, d->col_qedge.p,
                   d->E, frames, d->N, d->cfg.llr_scale};
        dim3 grid((d->N + kInitCols - 1) / kInitCols, tiles);
        d->init_fn<<<grid, kBlock, 0, s>>>(a);
        StateArgs st{d->done.p, nullptr, d->iters.p, nullptr, frames, 0, max_iter, freeze ? 1 : 0};
        /* overflow tiles (and unused tiles in between) are born finished: frames beyond `frames` */
        state_kernel<V><<<use_tail ? d->TA : tiles, 64, 0, s>>>(st);
        HIP_TRY(span_end(d, s));
    }

    int launched = start_round - 1;
    /* host polling: every poll_interval rounds -- and every round once a poll has seen a tenth of the frames
     * finished: from there on the running count falls fast (rate 9/10 at 4096 frames: 4096, 3501, 681, 41
     * frames take part in rounds 4..7), and the round after which a quarter is left is the one to hand over at */
    bool poll_dense = false;
    for (int it = start_round; it <= rounds; ++it) {
        const bool fat = idle_after > 0 && it > idle_after;      /* probably idle: fewer, fatter workgroups */
        /* check_i: R_i = check(Q_{i-1}) */
        {
            const int rcp = enqueue_check_phase<V>(d, s, tiles, frames, it, max_iter, fat, tr);
            if (rcp) return rcp;
        }
        {
            const int rcv = enqueue_var_phase<V>(d, s, tiles, frames, it, max_iter, fat, tr);
            if (rcv) return rcv;
        }
        launched = it;
        /* syndrome of bits_i, then freeze the frames that are clean (iters = i) */
        if (freeze || it == rounds) {
            HIP_TRY(span_begin(d, s, 3));
            uint64_t *fw = d->failw.p + (size_t)it * slot;
            const int rbk = (d->M + kBlock - 1) / kBlock;
            SyndromeArgs sa{d->row_ptr.p, d->edge_col.p, d->hard.p, fw, d->done.p, d->M, d->N,
                            d->tune_syn_xcd ? tiles : 0, rbk, tr};
            dim3 sgrid = d->tune_syn_xcd ? dim3(8 * rbk * ((tiles + 7) / 8)) : dim3(rbk, tiles);
            syndrome_kernel<V><<<sgrid, kBlock, 0, s>>>(sa);
            StateArgs st{d->done.p, fw, d->iters.p, nullptr, frames, it, max_iter, 1, tr, use_tail ? d->running.p : nullptr,
                         d->summary.p + 2};
            const bool poll = freeze && it < rounds && d->cfg.poll_interval > 0 && !d->suppress_poll &&
                              ((it % d->cfg.poll_interval) == 0 || poll_dense);
            if (poll) {
                HIP_TRY(hipMemsetAsync(d->active.p, 0, sizeof(int32_t), s));
                st.active = d->active.p;
            }
            state_kernel<V><<<tiles, 64, 0, s>>>(st);
            if (use_tail && it < rounds) {
                /* hand the last running frames over to the overflow tiles if their number has fallen
                 * below the threshold after this round (decided by the kernel; usually it just returns) */
                ta.iter = it;
                const unsigned tg = (unsigned)std::min<int64_t>(1024, d->E + 2 * (int64_t)d->N);
                if (d->msg_size == 2) tail_gather_kernel<V, _Float16><<<tg, kBlock, 0, s>>>(ta);
                else tail_gather_kernel<V, float><<<tg, kBlock, 0, s>>>(ta);
            }
            HIP_TRY(span_end(d, s));
            if (poll) {
                HIP_TRY(hipMemcpyAsync(d->h_active, d->active.p, sizeof(int32_t),
                                       hipMemcpyDeviceToHost, s));
                HIP_TRY(hipStreamSynchronize(s));
                const int running = *d->h_active;
                if (running == 0) break;        /* every frame frozen: MyLdpc.cpp:1035-1036 */
                /* ... where a round is long enough for a host round trip (about 25 us) not to matter: from 1.5 GB of
                 * message traffic per round (about 0.3 ms) */
                if ((int64_t)running * 10 <= frames * 9 && d->child && tiles > 1 &&
                    (double)d->E * (double)frames * 4.0 * (double)msz > 1.5e9) poll_dense = true;
                if (d->child && running <= d->compact_threshold && (int64_t)running * 4 <= frames && tiles > 1 && !d->tap_iter) {
                    const int rc = compact_and_finish<V>(d, frames, running, it, s);
                    if (rc) return rc;
                    launched = d->handed_to->last_iterations;
                    break;
                }
            }
        }
    }
    d->last_iterations = launched;
    d->last_tiles = tiles;
    if (resume) return LDPC_OK;             /* the parent packs */

    HIP_TRY(span_begin(d, s, 3));
    if (use_tail) tail_scatter_kernel<V><<<2048, kBlock, 0, s>>>(ta);
    {
        PackArgs pa{d->hard.p, out_dev, d->iters.p, iters_dev, frames, out_bytes, d->N, d->cfg.K,
                    d->cfg.pack_mode};
        if (d->cfg.pack_mode == LDPC_PACK_BYTES) {
            if (out_dev && frames) pack_kernel<V><<<pack_grid<V>(d->cfg.K, tiles), kBlock, 0, s>>>(pa);
        } else {
            const int64_t n = std::max<int64_t>(out_bytes, frames);
            dim3 grid((unsigned)((n + kBlock - 1) / kBlock));
            if (out_dev && frames) pack_kernel<V><<<grid, kBlock, 0, s>>>(pa);
        }
        /* after the final state_kernel `done` marks exactly the converged frames */
        summary_kernel<V><<<(unsigned)((frames + 255) / 256), 256, 0, s>>>(
            d->iters.p, d->done.p, d->failw.p, frames, 1, d->summary.p);
        if (!d->is_child && d->cfg.poll_interval == 0 && freeze && !d->summary_pending) {
            /* the next call's idle hint */
            HIP_TRY(hipMemcpyAsync(d->h_summary, d->summary.p, 2 * sizeof(int32_t), hipMemcpyDeviceToHost, s));
            HIP_TRY(hipEventRecord(d->ev_summary, s));
            d->summary_pending = true;
        }
    }
    HIP_TRY(span_end(d, s));
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int build_classes(ldpc_decoder *d, const ldpc_graph *g)
{
    std::map<int, std::vector<int32_t>> rows_by_deg, cols_by_deg, rowids_by_deg;
    for (int32_t m = 0; m < g->M; ++m) {
        const int deg = g->row_ptr[m + 1] - g->row_ptr[m];
        if (deg > 0) { rows_by_deg[deg].push_back(g->row_ptr[m]); rowids_by_deg[deg].push_back(m); }
    }
    /* column-local fusion: degree-2 columns whose checks are consecutive list rows of one wave */
    std::vector<char> fused_col((size_t)g->N, 0);
    d->row_classes.resize(rows_by_deg.size());
    size_t i = 0;
    for (auto &kv : rows_by_deg) {
        RowClass &rc = d->row_classes[i++];
        rc.degree = kv.first;
        rc.count = (int)kv.second.size();
        rc.h_e0 = kv.second;
        HIP_TRY(rc.e0.upload(kv.second));
        const std::vector<int32_t> &ids = rowids_by_deg[kv.first];
        const int rpw = d->link_rpw;
        if (rpw >= 2 && rc.degree >= 2 && rc.degree <= ldpc::kMaxUnrolledDegree) {
            /* Guided chunks: what the chip holds at once (16 waves per CU) is the launch's last
             * generation of waves; that many chunks per tile, at the end of the row list, are cut to
             * a quarter of the rows (>= 2), so that the launch drains over a short chunk's time.  With
             * fewer than 4 tiles everything would be "last generation": equal chunks then. */
            rc.n_big = (rc.count + rpw - 1) / rpw;
            rc.small_rows = rpw;
            const int small = std::max(2, rpw / 4);
            const bool forced = ldpc::tune_forced_on(d->tune_link_guided);         /* tests: also on small launches */
            if ((d->T >= 4 || forced) && small < rpw && !ldpc::tune_forced_off(d->tune_link_guided)) {
                const int64_t last_generation = (int64_t)d->cus * 16 / d->T;       /* chunks per tile */
                int64_t big = (int64_t)rc.n_big - last_generation;
                if (big <= 0 && forced) big = rc.n_big - std::max(1, rc.n_big / 3);
                if (big > 0) { rc.n_big = (int)big; rc.small_rows = small; }
            }
            std::vector<char> chunk_end((size_t)rc.count, 0);
            for (int c = 0, nc = ldpc::link_chunk_count(rpw, rc.n_big, rc.small_rows, rc.count); c < nc; ++c) {
                int rb, re;
                ldpc::link_chunk_rows(c, rpw, rc.n_big, rc.small_rows, rc.count, &rb, &re);
                if (re > rb) chunk_end[(size_t)re - 1] = 1;
            }
            std::vector<int32_t> lcol((size_t)rc.count, -1), lpos((size_t)rc.count, 0);
            for (int idx = 0; idx + 1 < rc.count; ++idx) {
                if (chunk_end[idx]) continue;                  /* next row belongs to another wave */
                const int32_t m = ids[idx], m2 = ids[idx + 1];
                for (int32_t p = g->row_ptr[m]; p < g->row_ptr[m + 1]; ++p) {
                    const int32_t c = g->cols[p];
                    if (g->col_ptr[c + 1] - g->col_ptr[c] != 2 || fused_col[c]) continue;
                    const int32_t ea = g->col_edge[g->col_ptr[c]], eb = g->col_edge[g->col_ptr[c] + 1];
                    if (ea != p || g->rows[eb] != m2) continue;   /* edges ascending: row m first */
                    lcol[idx] = c;
                    lpos[idx] = (p - g->row_ptr[m]) | ((eb - g->row_ptr[m2]) << 8);
                    fused_col[c] = 1;
                    ++rc.linked;
                    break;
                }
            }
            if (rc.linked * 4 >= rc.count) {                    /* worth a specialised kernel */
                HIP_TRY(rc.link_col.upload(lcol));
                HIP_TRY(rc.link_pos.upload(lpos));
            } else {
                for (int idx = 0; idx < rc.count; ++idx)
                    if (lcol[idx] >= 0) fused_col[lcol[idx]] = 0;
                rc.linked = 0;
            }
        }
    }
    for (int32_t n = 0; n < g->N; ++n) {
        const int deg = g->col_ptr[n + 1] - g->col_ptr[n];
        if (!fused_col[n]) cols_by_deg[deg].push_back(n);   /* degree 0: still needs its hard bit */
    }
    d->col_classes.resize(cols_by_deg.size());
    i = 0;
    for (auto &kv : cols_by_deg) {
        ColClass &cc = d->col_classes[i++];
        cc.degree = kv.first;
        cc.count = (int)kv.second.size();
        std::vector<int32_t> edges;
        edges.reserve((size_t)cc.count * std::max(cc.degree, 1));
        for (int32_t n : kv.second)
            for (int32_t p = g->col_ptr[n]; p < g->col_ptr[n + 1]; ++p) edges.push_back(g->col_edge[p]);
        if (edges.empty()) edges.push_back(0);
        HIP_TRY(cc.col.upload(kv.second));
        HIP_TRY(cc.edge.upload(edges));
    }
    {
        /* Q in the order its writers produce it (CheckArgs::qpos): the column classes one after the other, a class of degree D
         * as D streams of its columns' k-th messages -- the variable-node waves at work write D moving fronts --, then the
         * fused columns' edges, which the column-fused check kernel writes row by row, in row order */
        d->h_qpos.assign((size_t)g->E, -1);
        int64_t slot = 0;
        if (d->tune.q_order >= 0)
            for (ColClass &cc : d->col_classes) {
                cc.q_base = slot;
                const std::vector<int32_t> &members = cols_by_deg[cc.degree];
                for (size_t ci = 0; ci < members.size(); ++ci)
                    for (int k = 0; k < cc.degree; ++k)
                        d->h_qpos[(size_t)g->col_edge[(size_t)g->col_ptr[members[ci]] + k]] = (int32_t)(slot + (int64_t)k * cc.count + (int64_t)ci);
                slot += (int64_t)cc.degree * cc.count;
            }
        /* (tune_q_order = -1: every edge in its own slot, as in R) */
        for (int64_t e = 0; e < g->E; ++e)
            if (d->h_qpos[(size_t)e] < 0) d->h_qpos[(size_t)e] = d->tune.q_order >= 0 ? (int32_t)slot++ : (int32_t)e;
        std::vector<int32_t> cq((size_t)g->E);
        for (int64_t p = 0; p < g->E; ++p) cq[(size_t)p] = d->h_qpos[(size_t)g->col_edge[(size_t)p]];
        HIP_TRY(d->qpos.upload(d->h_qpos));
        HIP_TRY(d->col_qedge.upload(cq));
    }
    return LDPC_OK;
}

/* Which classes share a launch (degree buckets), which go alone, and whether the few rows outside a
 * linked class ride along with its launch.  LDPC_TUNE_OFF(LDPC_TUNE_MERGE): one launch per class. */
int plan_launches(ldpc_decoder *d)
{
    using ldpc::GroupClass;
    const int V = d->V;
    const bool merge = ldpc::tune_pick(d->tune.merge, true) && !d->tune_check_wide;
    d->check_groups.clear(); d->var_groups.clear(); d->check_solo.clear(); d->var_solo.clear();
    d->n_extra = 0; d->extra_edges = 0;
    int linked_classes = 0;
    int64_t unlinked_rows = 0;
    for (auto &rc : d->row_classes) { if (rc.linked) ++linked_classes; else unlinked_rows += rc.count; }
    const bool as_extra = merge && linked_classes == 1 && unlinked_rows > 0 && unlinked_rows <= 64;
    std::vector<int> cb[kCheckBuckets], vb[kVarBuckets];
    std::vector<int32_t> xe0, xdeg;
    for (int i = 0; i < (int)d->row_classes.size(); ++i) {
        RowClass &rc = d->row_classes[i];
        if (rc.linked) continue;
        if (as_extra) {
            for (int32_t e : rc.h_e0) { xe0.push_back(e); xdeg.push_back(rc.degree); d->extra_edges += rc.degree; }
            continue;
        }
        int b = -1;
        for (int k = 0; k < kCheckBuckets; ++k)
            if (rc.degree >= kCheckBucketLo[k] && rc.degree <= kCheckBucketHi[k] && rc.degree <= d->max_check_unrolled &&
                d->check_group_fn[k]) b = k;
        if (merge && b >= 0) cb[b].push_back(i); else d->check_solo.push_back(i);
    }
    if (as_extra) {
        d->n_extra = (int)xe0.size();
        HIP_TRY(d->extra_e0.upload(xe0));
        HIP_TRY(d->extra_deg.upload(xdeg));
    }
    for (int i = 0; i < (int)d->col_classes.size(); ++i) {
        const ColClass &cc = d->col_classes[i];
        int b = -1;
        for (int k = 0; k < kVarBuckets; ++k)
            if (cc.degree >= kVarBucketLo[k] && cc.degree <= kVarBucketHi[k] && d->var_group_fn[k]) b = k;
        if (merge && b >= 0) vb[b].push_back(i); else d->var_solo.push_back(i);
    }
    const int rpw = d->tune_rpw ? d->tune_rpw : 2, cpw = d->tune_cpw ? d->tune_cpw : 1;
    auto make = [&](std::vector<ClassGroup> &groups, std::vector<int> &solo, const std::vector<int> &members, int bucket,
                    int lo, int hi, bool rows) -> hipError_t {
        if (members.size() < 2) { for (int i : members) solo.push_back(i); return hipSuccess; }
        groups.emplace_back();
        ClassGroup &g = groups.back();
        g.bucket = bucket; g.lo = lo; g.hi = hi; g.members = members;
        std::vector<GroupClass> tab, tabf;
        constexpr int fatk = kIdleFat;
        for (int i : members) {
            GroupClass gc{};
            if (rows) {
                const RowClass &rc = d->row_classes[i];
                gc.degree = rc.degree; gc.count = rc.count; gc.ids = rc.e0.p; gc.edges = nullptr; gc.q_base = -1;
            } else {
                const ColClass &cc = d->col_classes[i];
                gc.degree = cc.degree; gc.count = cc.count; gc.ids = cc.col.p; gc.edges = cc.edge.p; gc.q_base = cc.q_base;
            }
            auto blocks_of = [&](int per_wave) {
                const int waves = ((gc.count + per_wave - 1) / per_wave) * (rows ? V / d->check_group_width : 1);
                return (waves + ldpc::kWavesPerBlock - 1) / ldpc::kWavesPerBlock;
            };
            gc.block_begin = g.blocks;
            tab.push_back(gc);
            g.blocks += blocks_of(rows ? rpw : cpw);
            gc.block_begin = g.blocks_fat;
            tabf.push_back(gc);
            g.blocks_fat += blocks_of((rows ? rpw : cpw) * fatk);
        }
        const hipError_t e = g.table.upload(tab);
        return e != hipSuccess ? e : g.table_fat.upload(tabf);
    };
    for (int k = 0; k < kCheckBuckets; ++k) HIP_TRY(make(d->check_groups, d->check_solo, cb[k], k, kCheckBucketLo[k], kCheckBucketHi[k], true));
    for (int k = 0; k < kVarBuckets; ++k) HIP_TRY(make(d->var_groups, d->var_solo, vb[k], k, kVarBucketLo[k], kVarBucketHi[k], false));
    return LDPC_OK;
}

/* HBM message arrays, per-degree work lists and kernel tables of a streaming flooding decoder. */
int setup_flooding(ldpc_decoder *d, const ldpc_graph *g, size_t TF)
{
    const ldpc_decoder_config *cfg = &d->cfg;
    d->msg_size = cfg->msg_dtype == LDPC_MSG_F16 ? 2 : 4;
    HIP_TRY(d->chan.alloc(TF * d->N * d->msg_size));
    HIP_TRY(d->Q.alloc(TF * (size_t)d->E * d->msg_size));
    HIP_TRY(d->R.alloc(TF * (size_t)d->E * d->msg_size));
    int rc = build_classes(d, g);
    if (rc) return rc;
    /* the kernels live in flood_sp.hip / flood_ms.hip / flood_ms16.hip (flood_tables.hpp) */
    ldpc::FloodFns fns;
    if (cfg->algo == LDPC_ALGO_SP) ldpc::fill_flood_sp(d->V, &fns);
    else if (cfg->msg_dtype == LDPC_MSG_F16) ldpc::fill_flood_ms16(d->V, &fns);
    else ldpc::fill_flood_ms(d->V, &fns);
    memcpy(d->check_fn, fns.check, sizeof fns.check);
    memcpy(d->check_fn_wide, fns.check_wide, sizeof fns.check_wide);
    memcpy(d->link_fn, fns.link, sizeof fns.link);
    memcpy(d->link_narrow_fn, fns.link_narrow, sizeof fns.link_narrow);
    memcpy(d->link_deep_fn, fns.link_deep, sizeof fns.link_deep);
    memcpy(d->link_half_fn, fns.link_half, sizeof fns.link_half);
    memcpy(d->var_fn, fns.var, sizeof fns.var);
    memcpy(d->check_group_fn, fns.check_group, sizeof fns.check_group);
    d->check_group_width = fns.check_group_width;
    memcpy(d->var_group_fn, fns.var_group, sizeof fns.var_group);
    d->init_fn = fns.init;
    d->max_check_unrolled = fns.max_check_unrolled;
    return plan_launches(d);
}

/* The column-fused check kernel exists in wide waves (V values per lane, 128 VGPRs), in narrow waves (1 value per
 * lane, 46 VGPRs) and, for tiles of 256 frames, with 2 values per lane (68 VGPRs).  Which one is fastest was
 * different from box to box in rounds 1 and 2 (narrow ahead by 2 % on round 1's boxes, wide 6-17 % ahead on round
 * 2's: profiles/r02_ab_link_wide.txt) -- part of which was the placement effect the search below deals with: the
 * forms do not slow down by the same factor on a slow pair of allocations.  Unless the caller fixes the choice
 * (LDPC_TUNE_LINK_NARROW / LINK_HALF), a decoder with more than one frame per lane therefore times the forms on
 * its own arrays when it is created -- interleaved launches, a few milliseconds -- and keeps the fastest (wide
 * also wins at 256 ... 1024 frames: +4 ... +8 % on the whole decode).  The arrays hold zeros, which the first
 * decode overwrites; results do not depend on the choice (the tests run all forms). */
template <int V> int calibrate_link(ldpc_decoder *d)
{
    using namespace ldpc;
    RowClass *rcp = nullptr;
    for (auto &rc : d->row_classes) if (rc.linked) rcp = &rc;
    if (!rcp || !d->link_fn[rcp->degree] || !d->link_narrow_fn[rcp->degree]) return LDPC_OK;
    RowClass &rc = *rcp;
    const int tiles = d->T;
    hipStream_t s = d->stream;
    HIP_TRY(hipMemsetAsync(d->Q.p, 0, d->Q.n, s));
    HIP_TRY(hipMemsetAsync(d->chan.p, 0, d->chan.n, s));
    HIP_TRY(hipMemsetAsync(d->done.p, 0, d->done.n * sizeof(uint64_t), s));
    hipEvent_t ev[2] = {nullptr, nullptr};
    HIP_TRY(hipEventCreate(&ev[0]));
    HIP_TRY(hipEventCreate(&ev[1]));
    float best[3] = {1e30f, 1e30f, 1e30f};
    const int candidates = d->link_half_fn[rc.degree] ? 3 : 2;
    hipError_t err = hipSuccess;
    for (int rep = 0; rep < 4 && err == hipSuccess; ++rep) {
        for (int nar = 0; nar < candidates && err == hipSuccess; ++nar) {
            CheckArgs a{d->Q.p, d->R.p, rc.e0.p, d->done.p, d->E, rc.count, d->link_rpw, rc.degree, TailRef{nullptr, 0, 0}};
            a.qpos = d->qpos.p;
            LinkArgs lk{rc.link_col.p, rc.link_pos.p, d->chan.p, d->Q.p, d->hard.p, d->N, 1, 0, nullptr, nullptr, 0, 0,
                        rc.n_big, rc.small_rows};
            const int waves = link_chunk_count(d->link_rpw, rc.n_big, rc.small_rows, rc.count) * (nar == 1 ? V : nar == 2 ? V / 2 : 1);
            lk.link_blocks = (waves + kWavesPerBlock - 1) / kWavesPerBlock;
            const dim3 grid = flood_grid(d, lk.link_blocks, tiles, &a.tiles_first, true);
            err = hipEventRecord(ev[0], s);
            (nar == 2 ? d->link_half_fn : nar == 1 ? d->link_narrow_fn : d->link_fn)[rc.degree]<<<grid, kBlock, 0, s>>>(a, lk);
            if (err == hipSuccess) err = hipEventRecord(ev[1], s);
            if (err == hipSuccess) err = hipEventSynchronize(ev[1]);
            float ms = 0;
            if (err == hipSuccess) err = hipEventElapsedTime(&ms, ev[0], ev[1]);
            if (err == hipSuccess && rep > 0 && ms < best[nar]) best[nar] = ms;      /* rep 0 warms up */
        }
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    if (err != hipSuccess) return fail(LDPC_ERR_HIP, "link calibration: %s", hipGetErrorString(err));
    HIP_TRY(hipGetLastError());
    int pick = 0;
    for (int k = 0; k < candidates; ++k) { d->link_cal_ms[k] = best[k]; if (best[k] < best[pick]) pick = k; }
    d->tune_link_narrow = pick;
    d->link_calibrated = true;
    return LDPC_OK;
}

/* One message round (check phase + variable-node phase) over all tiles of the decoder on zeroed arrays, best of three
 * timed repetitions (ms). */
template <int V> int time_check_phase(ldpc_decoder *d, float *ms_out)
{
    hipStream_t s = d->stream;
    HIP_TRY(hipMemsetAsync(d->Q.p, 0, d->Q.n, s));
    HIP_TRY(hipMemsetAsync(d->chan.p, 0, d->chan.n, s));
    HIP_TRY(hipMemsetAsync(d->done.p, 0, d->done.n * sizeof(uint64_t), s));
    hipEvent_t ev[2] = {nullptr, nullptr};
    HIP_TRY(hipEventCreate(&ev[0]));
    HIP_TRY(hipEventCreate(&ev[1]));
    float best = 1e30f;
    hipError_t err = hipSuccess;
    int rc = LDPC_OK;
    for (int rep = 0; rep < 4 && err == hipSuccess && rc == LDPC_OK; ++rep) {
        err = hipEventRecord(ev[0], s);
        rc = enqueue_check_phase<V>(d, s, d->T, (int64_t)d->T * d->F, 1, d->cfg.max_iter, false, ldpc::TailRef{nullptr, 0, 0});
        if (rc == LDPC_OK)
            rc = enqueue_var_phase<V>(d, s, d->T, (int64_t)d->T * d->F, 1, d->cfg.max_iter, false, ldpc::TailRef{nullptr, 0, 0});
        if (err == hipSuccess) err = hipEventRecord(ev[1], s);
        if (err == hipSuccess) err = hipEventSynchronize(ev[1]);
        float ms = 0;
        if (err == hipSuccess) err = hipEventElapsedTime(&ms, ev[0], ev[1]);
        if (err == hipSuccess && rep > 0 && ms < best) best = ms;          /* rep 0 warms up */
    }
    (void)hipEventDestroy(ev[0]);
    (void)hipEventDestroy(ev[1]);
    if (rc) return rc;
    if (err != hipSuccess) return fail(LDPC_ERR_HIP, "placement search: %s", hipGetErrorString(err));
    *ms_out = best;
    return LDPC_OK;
}

/* Where the message arrays lie in device memory decides how fast the streaming check kernels run: the same
 * kernel on the same data takes 1.27, 1.35 or 1.53 ms per launch depending on the allocations it works on, for
 * as long as they live (tools/gpu_placement_probe2.py: six decoders alive in one process, each with its own time,
 * round after round; virtual addresses, offsets inside an allocation, clocks, power and temperature do not predict
 * it -- rounds 2 and 3 looked; profiles/r03_placement_search.txt).  This was the "123 ms or 137 ms regime" of the
 * headline step.  It is a property of the PAIR of allocations behind Q and R: with Q fixed some fresh R allocations
 * are fast and some slow, with R fixed the same holds for Q, the channel array does not matter
 * (tools/gpu_array_trials.py), and consecutive allocations tend to share their luck.  So a decoder whose arrays are
 * large does not take its first allocations as they come: holding what it has, it tries up to `tune_place`
 * (default 6) fresh allocations for R, then for Q, times one message round (check + variable-node phase) with each,
 * keeps the fastest and
 * releases the rest at the end; after at least four measurements a stage stops as soon as it has seen the fast speed next to the slow one (a
 * candidate at least 9 % faster than another).  No guarantee: in some processes every pair is slow.  About 10 ms and 4 GB per candidate while the decoder is being created. */
template <int V> int placement_search(ldpc_decoder *d, size_t TF)
{
    const size_t bq = TF * (size_t)d->E * d->msg_size, bc = TF * d->N * d->msg_size;
    const int want = d->tune.place == 0 ? (2 * bq + bc >= ((size_t)256 << 20) ? 6 : 1) : d->tune.place;
    if (want <= 1) return LDPC_OK;
    float best_ms = 0.0f;
    int rc = time_check_phase<V>(d, &best_ms);
    if (rc) return rc;
    std::vector<DevBuf<uint8_t>> held;           /* the allocations that lost: kept alive until the search ends */
    d->place_ms[0] = best_ms;
    d->place_candidates = 1;
    d->place_kept = 0;
    float lo = best_ms, hi = best_ms;
    for (int stage = 0; stage < 2; ++stage) {
        DevBuf<uint8_t> &arr = stage == 0 ? d->R : d->Q;
        for (int c = 1; c < want; ++c) {
            /* three speeds of the check phase occur (about 1 : 0.88 : 0.83, i.e. 1 : 0.93 : 0.895 for the whole round):
             * stop once the fastest of them has been seen next to the slowest -- but not before four measurements: the
             * speeds within the fast class still differ by 2-3 % (14 processes: a search that stopped after 2.63, 2.61,
             * 2.36 ms kept 2.36 where its neighbours found 2.28-2.31) */
            if (lo < 0.91f * hi && best_ms <= lo && d->place_candidates >= 4) break;
            size_t free_b = 0, total_b = 0;
            if (hipMemGetInfo(&free_b, &total_b) != hipSuccess || free_b < 2 * bq + ((size_t)2 << 30)) break;
            /* Device memory comes in two classes that alternate every 16 GiB of (physical) address space, and a read
             * stream and a write stream in DIFFERENT classes do not get in each other's way (tools/offset_map.hip: a copy
             * inside one 40 GiB allocation runs at 6.2 TB/s to a destination less than 16 GiB away, 6.4 beyond, 6.8 at
             * the transition, and back to 6.2 from 32 GiB on; tools/pair_map.hip: separate 4 GiB allocations come in
             * alternating blocks of four).  Physical addresses are not visible, but allocations made one after the other
             * mostly are neighbours: a spacer that brings the distance to the array's partner to about 16 GiB, held while
             * the candidate is allocated, makes the other class likely.  The timing below decides. */
            DevBuf<uint8_t> cand, spacer;
            const size_t period = (size_t)16 << 30;
            if (c == 1 && arr.n < period && free_b > period + 2 * bq + ((size_t)2 << 30)) {
                if (spacer.alloc(period - arr.n) != hipSuccess) (void)hipGetLastError();
            }
            if (cand.alloc(arr.n) != hipSuccess) { (void)hipGetLastError(); break; }
            spacer.release();
            std::swap(arr, cand);                                        /* the candidate is the decoder's array now */
            float ms = 0.0f;
            rc = time_check_phase<V>(d, &ms);
            if (rc) return rc;
            if (d->place_candidates < 16) d->place_ms[d->place_candidates] = ms;
            lo = std::min(lo, ms); hi = std::max(hi, ms);
            if (ms < best_ms) {
                best_ms = ms;
                d->place_kept = d->place_candidates;
            } else {
                std::swap(arr, cand);                                    /* back to the array it had */
            }
            ++d->place_candidates;
            held.push_back(std::move(cand));
        }
    }
    return LDPC_OK;                                                      /* `held` releases the losers here */
}

}  // namespace

/* ================================================================== C ABI */

/* set while a decoder creates its tail-compaction child: the child must not create one of its own */
/* 0 while an ordinary decoder is created, 1 for its hand-over child, 2 for the child's own child */
static thread_local int t_child_depth = 0;

extern "C" {

int ldpc_abi_version(void) { return LDPC_HIP_ABI_VERSION; }

const char *ldpc_last_error(void) { return g_err.c_str(); }

int ldpc_device_count(int *count)
{
    if (!count) return fail(LDPC_ERR_ARG, "count is NULL");
    *count = 0;
    int n = 0;
    hipError_t e = hipGetDeviceCount(&n);
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "hipGetDeviceCount: %s", hipGetErrorString(e));
    *count = n;
    return LDPC_OK;
}

int64_t ldpc_out_bytes(int32_t K, int64_t frames, int32_t pack_mode)
{
    if (frames <= 0 || K <= 0) return 0;
    if (pack_mode == LDPC_PACK_BYTES) return (frames - 1) * (int64_t)K / 8 + K / 8;
    return (frames * (int64_t)K + 7) / 8;
}

int ldpc_graph_create(const int32_t *rows, const int32_t *cols, int64_t E, int32_t M, int32_t N,
                      ldpc_graph **out)
{
    if (!out) return fail(LDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!rows || !cols) return fail(LDPC_ERR_ARG, "rows/cols is NULL");
    if (M <= 0 || N <= 0 || E <= 0) return fail(LDPC_ERR_ARG, "M, N, E must be positive");
    if (E > 0x7fffffffLL) return fail(LDPC_ERR_ARG, "E does not fit int32 edge ids");
    for (int64_t e = 0; e < E; ++e) {
        if (rows[e] < 0 || rows[e] >= M || cols[e] < 0 || cols[e] >= N)
            return fail(LDPC_ERR_ARG, "edge %lld = (%d, %d) outside %d x %d", (long long)e, rows[e],
                        cols[e], M, N);
        if (e && (rows[e] < rows[e - 1] || (rows[e] == rows[e - 1] && cols[e] <= cols[e - 1])))
            return fail(LDPC_ERR_ARG, "edges must be in strictly ascending row-major order (edge %lld)",
                        (long long)e);
    }
    ldpc_graph *g = new (std::nothrow) ldpc_graph;
    if (!g) return fail(LDPC_ERR_NOMEM, "out of memory");
    g->M = M; g->N = N; g->E = E;
    g->rows.assign(rows, rows + E);
    g->cols.assign(cols, cols + E);
    g->row_ptr.assign((size_t)M + 1, 0);
    g->col_ptr.assign((size_t)N + 1, 0);
    for (int64_t e = 0; e < E; ++e) { ++g->row_ptr[rows[e] + 1]; ++g->col_ptr[cols[e] + 1]; }
    for (int32_t m = 0; m < M; ++m) {
        g->max_row_deg = std::max(g->max_row_deg, g->row_ptr[m + 1]);
        g->row_ptr[m + 1] += g->row_ptr[m];
    }
    for (int32_t n = 0; n < N; ++n) {
        g->max_col_deg = std::max(g->max_col_deg, g->col_ptr[n + 1]);
        g->col_ptr[n + 1] += g->col_ptr[n];
    }
    /* column lists in ascending edge id, as the reference's linked lists are
     * appended in edge order (MyLdpc.cpp:209-218) */
    g->col_edge.assign((size_t)E, 0);
    std::vector<int32_t> fill(g->col_ptr.begin(), g->col_ptr.end() - 1);
    for (int64_t e = 0; e < E; ++e) g->col_edge[fill[cols[e]]++] = (int32_t)e;
    *out = g;
    return LDPC_OK;
}

int ldpc_graph_destroy(ldpc_graph *g)
{
    delete g;
    return LDPC_OK;
}

int ldpc_graph_info(const ldpc_graph *g, int32_t *M, int32_t *N, int64_t *E, int32_t *max_row_deg,
                    int32_t *max_col_deg)
{
    if (!g) return fail(LDPC_ERR_ARG, "graph is NULL");
    if (M) *M = g->M;
    if (N) *N = g->N;
    if (E) *E = g->E;
    if (max_row_deg) *max_row_deg = g->max_row_deg;
    if (max_col_deg) *max_col_deg = g->max_col_deg;
    return LDPC_OK;
}

void ldpc_decoder_config_init(ldpc_decoder_config *cfg)
{
    if (!cfg) return;
    memset(cfg, 0, sizeof *cfg);
    cfg->struct_size = sizeof *cfg;
    cfg->max_batch = 1;
    cfg->algo = LDPC_ALGO_SP;
    cfg->msg_dtype = LDPC_MSG_F32;
    cfg->max_iter = 40;       /* MyLdpc.cpp:24 */
    cfg->llr_scale = 8.0f;    /* decodeCL.c:9 */
    cfg->early_term = 1;
    cfg->pack_mode = LDPC_PACK_BYTES;
}

int ldpc_decoder_create(const ldpc_graph *g, const ldpc_decoder_config *cfg, ldpc_decoder **out)
{
    if (!out) return fail(LDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!g || !cfg) return fail(LDPC_ERR_ARG, "graph/config is NULL");
    if (cfg->struct_size != sizeof(ldpc_decoder_config))
        return fail(LDPC_ERR_ARG, "config struct_size %u != %zu (ABI mismatch)", cfg->struct_size,
                    sizeof(ldpc_decoder_config));
    if (cfg->K <= 0 || cfg->K > g->N) return fail(LDPC_ERR_ARG, "K=%d out of range", cfg->K);
    if (cfg->max_batch <= 0) return fail(LDPC_ERR_ARG, "max_batch must be positive");
    if (cfg->max_iter <= 0 || cfg->max_iter > 100000) return fail(LDPC_ERR_ARG, "max_iter out of range");
    if (cfg->algo != LDPC_ALGO_SP && cfg->algo != LDPC_ALGO_MS && cfg->algo != LDPC_ALGO_LAYERED &&
        cfg->algo != LDPC_ALGO_MS_FUSED && cfg->algo != LDPC_ALGO_LAYERED_HOST)
        return fail(LDPC_ERR_ARG, "unknown algo %d", cfg->algo);
    if (cfg->algo == LDPC_ALGO_LAYERED_HOST) {
        for (int32_t m = 1; m < g->M; ++m)
            if (g->row_ptr[m + 1] - g->row_ptr[m] != g->row_ptr[1] - g->row_ptr[0])
                return fail(LDPC_ERR_UNSUPPORTED, "LAYERED_HOST follows the reference's host-layered path, which sizes its "
                            "layers correctly only when every row of H has the same weight (MyLdpc.cpp:907,958)");
        if (g->max_row_deg > ldpc::kMaxUnrolledLayerDegree)
            return fail(LDPC_ERR_UNSUPPORTED, "LAYERED_HOST: row weight %d > %d", g->max_row_deg, ldpc::kMaxUnrolledLayerDegree);
    }
    if (cfg->pack_mode != LDPC_PACK_BYTES && cfg->pack_mode != LDPC_PACK_BITS)
        return fail(LDPC_ERR_ARG, "unknown pack_mode %d", cfg->pack_mode);
    if (cfg->frames_per_lane != 0 && cfg->frames_per_lane != 1 && cfg->frames_per_lane != 2 &&
        cfg->frames_per_lane != 4)
        return fail(LDPC_ERR_ARG, "frames_per_lane must be 0, 1, 2 or 4");
    if (cfg->msg_dtype != LDPC_MSG_F32 && cfg->msg_dtype != LDPC_MSG_F16)
        return fail(LDPC_ERR_ARG, "unknown msg_dtype %d", cfg->msg_dtype);
    if (cfg->msg_dtype == LDPC_MSG_F16 && cfg->algo != LDPC_ALGO_MS)
        return fail(LDPC_ERR_UNSUPPORTED, "fp16 messages are built for flooding min-sum only "
                    "(the probability-domain SP needs fp32 range; layered: not yet)");
    if (!ldpc::tune_valid(*cfg)) return fail(LDPC_ERR_ARG, "tuning / streams config fields out of range");

    int ndev = 0;
    HIP_TRY(hipGetDeviceCount(&ndev));
    if (cfg->device < 0 || cfg->device >= ndev)
        return fail(LDPC_ERR_HIP, "device %d not present (%d HIP devices)", cfg->device, ndev);
    HIP_TRY(hipSetDevice(cfg->device));

    ldpc_decoder *d = new (std::nothrow) ldpc_decoder;
    if (!d) return fail(LDPC_ERR_NOMEM, "out of memory");
    std::unique_ptr<ldpc_decoder> guard(d);
    d->cfg = *cfg;
    d->M = g->M; d->N = g->N; d->E = g->E;
    d->h_col_ptr = g->col_ptr; d->h_col_edge = g->col_edge; d->h_rows = g->rows; d->h_cols = g->cols;
    const ldpc::Tune tune = d->tune = ldpc::tune_from_config(*cfg);
    d->tune_rpw = tune.rows_per_wave;
    d->tune_cpw = tune.cols_per_wave;
    d->tune_syn_xcd = ldpc::tune_pick(tune.syn_xcd, true);
    d->tune_check_wide = ldpc::tune_pick(tune.check_wide, false);
    d->tune_link_narrow = ldpc::tune_pick(tune.link_half, false) ? 2 : (ldpc::tune_pick(tune.link_narrow, true) ? 1 : 0);
    d->tune_link_deep = ldpc::tune_pick(tune.link_deep, false);
    if (tune.link_rows) d->link_rpw = tune.link_rows < 0 ? 0 : tune.link_rows;
    d->tune_link_guided = tune.link_guided;
    d->tune_tiles_first = tune.tiles_first;
    HIP_TRY(hipDeviceGetAttribute(&d->cus, hipDeviceAttributeMultiprocessorCount, cfg->device));
    d->V = pick_frames_per_lane(*cfg, g->max_row_deg, g->max_col_deg);
    d->F = 64 * d->V;
    d->T = (cfg->max_batch + d->F - 1) / d->F;
    /* a single tile is latency-bound (one wave walks its rows one after the other): shorter row
     * chunks per wave, 4.1 -> 3.4 ms for one 50-iteration decode of the (64800, 32400) code */
    if (d->T == 1 && !tune.link_rows) d->link_rpw = 4;

    HIP_TRY(hipStreamCreateWithFlags(&d->stream, hipStreamNonBlocking));
    HIP_TRY(hipEventCreate(&d->ev_begin));
    HIP_TRY(hipEventCreate(&d->ev_end));
    HIP_TRY(hipHostMalloc((void **)&d->h_active, sizeof(int32_t), hipHostMallocDefault));
    HIP_TRY(hipHostMalloc((void **)&d->h_summary, 2 * sizeof(int32_t), hipHostMallocDefault));
    HIP_TRY(hipEventCreateWithFlags(&d->ev_summary, hipEventDisableTiming));
    HIP_TRY(d->row_ptr.upload(g->row_ptr));
    HIP_TRY(d->edge_col.upload(g->cols));
    HIP_TRY(d->col_ptr.upload(g->col_ptr));
    HIP_TRY(d->col_edge.upload(g->col_edge));
    /* Device-side tail: for asynchronous callers (poll_interval == 0) of the streaming flooding kernels
     * with early termination, when the batch has clearly more tiles than the overflow area.
     * LDPC_TUNE_OFF(LDPC_TUNE_DEVICE_TAIL) switches it off. */
    d->TO = (ldpc::kCompactCapacity + d->F - 1) / d->F;
    d->tail_enabled = cfg->early_term && cfg->poll_interval == 0 && ldpc::tune_pick(tune.device_tail, true) &&
                      (cfg->algo == LDPC_ALGO_SP || cfg->algo == LDPC_ALGO_MS) && d->T >= 4 * d->TO && t_child_depth == 0;
    if (!d->tail_enabled) d->TO = 0;
    d->TA = d->T + d->TO;
    const size_t TF = (size_t)d->TA * d->F;
    HIP_TRY(d->hard.alloc((size_t)d->TA * d->N * d->V));
    HIP_TRY(d->failw.alloc((size_t)(cfg->max_iter + 2) * d->TA * d->V));
    HIP_TRY(d->done.alloc((size_t)d->TA * d->V));
    HIP_TRY(d->iters.alloc(TF));
    HIP_TRY(d->active.alloc(1));
    HIP_TRY(d->summary.alloc(4));

    if (cfg->algo == LDPC_ALGO_MS_FUSED) {
        if (cfg->pack_mode != LDPC_PACK_BYTES && cfg->K % 8)
            return fail(LDPC_ERR_UNSUPPORTED, "MS_FUSED packs whole bytes per frame only");
        /* flood_ldsp_kernel<.., CHAIN = false> (posteriors in LDS, 16-byte check records) wherever it fits
         * with two workgroups per CU, else / with LDPC_TUNE_OFF(LDPC_TUNE_LDSP) the LDS-resident fused_flood_kernel */
        if (!ldpc::tune_forced_off(tune.ldsp)) {
            HIP_TRY(ldpc::engine_ldsp_plan_create(&d->ldsp, g->M, g->N, g->E, g->row_ptr, g->cols, cfg->layer_rows, cfg->K,
                                           cfg->max_batch, cfg->device, tune, /*flood=*/2));
            if (d->ldsp.eligible && (ldpc::tune_forced_on(tune.ldsp) || d->ldsp.lds_bytes <= 80 * 1024)) d->use_ldsp = true;
            else ldpc::ldsp_plan_destroy(&d->ldsp);
        }
        if (!d->use_ldsp) {
            HIP_TRY(ldpc::fused_plan_create(&d->fused, g->M, g->N, g->E, g->row_ptr, g->cols, cfg->layer_rows));
            if (!d->fused.eligible)
                return fail(LDPC_ERR_UNSUPPORTED, "MS_FUSED needs a quasi-cyclic H (circulant size = layer_rows) whose "
                            "posteriors fit in LDS");
        }
        d->use_fused = true;
    } else if (cfg->algo == LDPC_ALGO_LAYERED_HOST) {
        d->layered.host_arith = 1;
        int rc = ldpc::layered_plan_create(&d->layered, g->M, g->N, g->E, g->row_ptr, g->cols, cfg->layer_rows, d->T, d->V);
        if (rc == -1) return fail(LDPC_ERR_ARG, "layer_rows=%d must divide M=%d and rows of a layer "
                                  "must not share a column", cfg->layer_rows, g->M);
        if (rc) return fail(LDPC_ERR_HIP, "layered plan allocation failed: %s", hipGetErrorString(hipGetLastError()));
    } else if (cfg->algo == LDPC_ALGO_LAYERED) {
        /* short quasi-cyclic codes decode entirely in LDS, one launch (fused_kernels.hpp);
         * LDPC_TUNE_OFF(LDPC_TUNE_FUSED) keeps the streaming kernels (same results, bit for bit) */
        if (!ldpc::tune_forced_off(tune.fused) && (cfg->pack_mode == LDPC_PACK_BYTES || cfg->K % 8 == 0)) {
            HIP_TRY(ldpc::fused_plan_create(&d->fused, g->M, g->N, g->E, g->row_ptr, g->cols, cfg->layer_rows));
            d->use_fused = d->fused.eligible;
            /* layered_ldsp_kernel (posterior in LDS, 16-byte check records in cache) is the default for
             * every QC code it fits: larger codes cannot use the fully LDS-resident kernel at all, and on
             * short ones it is 1.2-2.9x faster (exact-width rows, bit-level sign algebra: 800 against 280 G
             * edge updates/s; circulants of <= 32 rows run several frames per wave in both).
             * LDPC_TUNE_OFF(LDPC_TUNE_LDSP) forbids it (the LDS-resident kernel is then used where it applies). */
            if (!ldpc::tune_forced_off(tune.ldsp)) {
                HIP_TRY(ldpc::engine_ldsp_plan_create(&d->ldsp, g->M, g->N, g->E, g->row_ptr, g->cols, cfg->layer_rows,
                                               cfg->K, cfg->max_batch, cfg->device, tune, /*flood=*/0));
                if (d->ldsp.eligible) d->use_fused = d->use_ldsp = true;
            }
        }
        /* the streaming plan (and its P / R arrays in HBM) only when no LDS-resident kernel applies;
         * a detected QC structure already implies that rows of a layer share no column */
        int rc = d->use_fused ? 0 : ldpc::layered_plan_create(&d->layered, g->M, g->N, g->E, g->row_ptr, g->cols,
                                                              cfg->layer_rows, d->T, d->V);
        if (rc == -1) return fail(LDPC_ERR_ARG, "layer_rows=%d must divide M=%d and rows of a layer "
                                  "must not share a column", cfg->layer_rows, g->M);
        if (rc) return fail(LDPC_ERR_HIP, "layered plan allocation failed: %s",
                            hipGetErrorString(hipGetLastError()));
    } else {
        /* flooding min-sum on a short quasi-cyclic code (layer_rows = circulant size given): the
         * same arithmetic in one LDS-resident launch (fused_flood_kernel<.., CHAIN>) */
        /* Measured (tools/gpu_short.py): for the flooding schedules the streaming kernels win at
         * full work once the batch is large (HBM-bound, 1.3-1.9x), the fused kernels win on latency
         * and for small batches; crossover near max_batch * E = 2^23 edge-frames.
         * LDPC_TUNE_ON(LDPC_TUNE_FUSED) forces the fused kernels, LDPC_TUNE_OFF the streaming ones. */
        if (cfg->msg_dtype == LDPC_MSG_F32 && cfg->layer_rows > 0 && cfg->frames_per_lane == 0 &&
            (cfg->pack_mode == LDPC_PACK_BYTES || cfg->K % 8 == 0)) {
            const bool small = (int64_t)cfg->max_batch * g->E <= (int64_t)1 << 23;
            if (ldpc::tune_pick(tune.fused, small)) {
                HIP_TRY(ldpc::fused_plan_create(&d->fused, g->M, g->N, g->E, g->row_ptr, g->cols, cfg->layer_rows));
                d->use_fused = d->fused.eligible && (cfg->algo == LDPC_ALGO_MS || d->fused.eligible_sp);
            }
            /* min-sum: posteriors in LDS (two images), one 16-byte record per check row (flood_ldsp_kernel).
             * Default wherever it fits with >= 2 workgroups per CU: 2-2.9x the LDS-resident kernel and the
             * streaming kernels on the 802.16e codes at any batch size ((2304, 1152): 9.9 / 4.0 / 2.9 Gbit/s
             * at 16 384 frames, 2.6 / 0.9 / 1.3 at full work), 2.4 / 2.0 / 1.7 / 1.2x the streaming kernels on
             * BG1-profile codes at Z = 64 / 128 / 256 / 384.  LDPC_TUNE_ON / OFF(LDPC_TUNE_LDSP) forces / forbids it. */
            if (cfg->algo == LDPC_ALGO_MS && !ldpc::tune_forced_off(tune.fused) && !ldpc::tune_forced_off(tune.ldsp)) {
                HIP_TRY(ldpc::engine_ldsp_plan_create(&d->ldsp, g->M, g->N, g->E, g->row_ptr, g->cols, cfg->layer_rows,
                                               cfg->K, cfg->max_batch, cfg->device, tune, /*flood=*/1));
                if (d->ldsp.eligible && (ldpc::tune_forced_on(tune.ldsp) || d->ldsp.lds_bytes <= 80 * 1024))
                    d->use_fused = d->use_ldsp = true;
                else
                    ldpc::ldsp_plan_destroy(&d->ldsp);
            }
        }
        if (!d->use_fused) {
            int rc = setup_flooding(d, g, TF);
            if (rc) return rc;
            /* narrow or wide column-fused check kernel: measured here unless the caller says which */
            if (tune.link_narrow == 0 && tune.link_half == 0 && !tune.link_deep && d->V >= 2 && t_child_depth == 0) {
                rc = d->V == 1 ? calibrate_link<1>(d) : d->V == 2 ? calibrate_link<2>(d) : calibrate_link<4>(d);
                if (rc) return rc;
            }
            if (t_child_depth == 0) {
                rc = d->V == 1 ? placement_search<1>(d, TF) : d->V == 2 ? placement_search<2>(d, TF) : placement_search<4>(d, TF);
                if (rc) return rc;
            }
            if (d->tail_enabled) {
                HIP_TRY(d->tail_state.alloc(4));
                HIP_TRY(d->tail_map.alloc((size_t)d->TO * d->F));
                HIP_TRY(d->running.alloc((size_t)cfg->max_iter + 2));
            }
            /* tail compaction: with host polling on, the last <= 512 running frames of a batch of several
             * tiles are finished by a small (8 x 64 frames) child decoder (cfg.tune_compact = -1: off, n: threshold) */
            /* the child takes over once at most a quarter of the batch still runs: 1024 frames for the 4096-frame
             * batches of the benchmark configurations (rate 9/10, fp16: 681 frames still run after round 5 of 8 and
             * sit in all 16 tiles; a 512-frame child had to wait for round 6), 512 otherwise */
            if (t_child_depth == 0) d->child_capacity = cfg->max_batch >= 4096 ? 2 * ldpc::kCompactCapacity : ldpc::kCompactCapacity;
            else d->child_capacity = cfg->max_batch > ldpc::kCompactCapacity ? ldpc::kCompactCapacity : ldpc::kLastCapacity;
            d->compact_threshold = d->child_capacity;
            if (tune.compact) d->compact_threshold = tune.compact < 0 ? 0 : std::min(d->child_capacity, tune.compact);
            /* the 1024-frame child has a 512-frame child of its own (rate 9/10: of the 681 frames handed over after round 5
             * only 41 still run after round 6, spread over the child's three tiles of 256), and that one a single tile of 64
             * frames (sum-product at 5 dB: a handful of frames in 4096 run all 50 rounds, one in each of its tiles) */
            const bool may_have_child = t_child_depth == 0 || cfg->max_batch > ldpc::kLastCapacity;
            if (cfg->early_term && cfg->poll_interval > 0 && d->T > 1 && d->compact_threshold > 0 && may_have_child) {
                ldpc_decoder_config cc = *cfg;
                cc.max_batch = d->child_capacity;
                /* tiles of 64 frames for the 512-frame child (the last few stragglers of a batch); tiles of 256 for the
                 * 1024-frame child, which takes over hundreds of frames: dense tiles, 8- / 16-byte accesses */
                cc.frames_per_lane = (d->child_capacity > ldpc::kCompactCapacity && d->V == 4) ? 4 : 1;
                cc.layer_rows = 0;                 /* streaming kernels, same arithmetic */
                cc.tune_compact = d->child_capacity > ldpc::kLastCapacity ? 0 : -1;     /* all but the last hand over once more */
                ++t_child_depth;
                rc = ldpc_decoder_create(g, &cc, &d->child);
                --t_child_depth;
                if (rc) return rc;
                d->child->is_child = true;
                HIP_TRY(hipSetDevice(cfg->device));
                HIP_TRY(d->cmap.alloc((size_t)d->child_capacity));
                HIP_TRY(d->cinv.alloc((size_t)d->T * d->F));
                HIP_TRY(d->cmoved.alloc((size_t)d->T * d->V));
            }
        }
    }
    *out = guard.release();
    return LDPC_OK;
}

/* Waits for the handle's OWN work: its two streams and, through the end-of-decode event, the
 * caller's stream of the last ldpc_decode_device call -- not for the device, which other handles
 * and the caller's other streams keep using. */
static void wait_for_own_work(ldpc_decoder *d)
{
    if (d->child) wait_for_own_work(d->child);
    (void)hipSetDevice(d->cfg.device);
    if (d->have_last && d->ev_end) (void)hipEventSynchronize(d->ev_end);
    if (d->stream) (void)hipStreamSynchronize(d->stream);
    if (d->copy_stream) (void)hipStreamSynchronize(d->copy_stream);
}

int ldpc_decoder_destroy(ldpc_decoder *d)
{
    if (!d) return LDPC_OK;
    /* blocks of caller memory an LDPC_HOST_INPUT_LOCK_PAGES call could not release: said loudly, here too */
    size_t stuck = d->stuck_blocks.size() + d->locked_blocks.size();
    for (ldpc_decoder *sh : d->shards) stuck += sh->stuck_blocks.size() + sh->locked_blocks.size();
    if (d->shards.empty()) {
        wait_for_own_work(d);
        ldpc::layered_plan_destroy(&d->layered);
        ldpc::fused_plan_destroy(&d->fused);
        ldpc::ldsp_plan_destroy(&d->ldsp);
    }
    delete d;            /* joins the handle's threads; a multi-device handle destroys its per-device decoders here */
    if (stuck) {
        fprintf(stderr, "ldpc_decoder_destroy: %zu page-locked block(s) of caller memory were never released\n", stuck);
        return fail(LDPC_ERR_STATE, "%zu page-locked block(s) of caller memory could not be released "
                    "(hipHostUnregister failed in an earlier ldpc_decode)", stuck);
    }
    return LDPC_OK;
}

int ldpc_shard_range(int64_t frames, int32_t part, int32_t parts, int32_t unit, int64_t *lo, int64_t *hi)
{
    if (!lo || !hi) return fail(LDPC_ERR_ARG, "lo/hi is NULL");
    if (frames < 0 || parts <= 0 || part < 0 || part >= parts || unit <= 0)
        return fail(LDPC_ERR_ARG, "shard_range(frames=%lld, part=%d, parts=%d, unit=%d)", (long long)frames, part, parts, unit);
    const int64_t nu = (frames + unit - 1) / unit, base = nu / parts, rem = nu % parts;
    const int64_t lo_u = part * base + std::min<int64_t>(part, rem);
    const int64_t hi_u = lo_u + base + (part < rem ? 1 : 0);
    *lo = std::min(lo_u * unit, frames);
    *hi = std::min(hi_u * unit, frames);
    return LDPC_OK;
}

int ldpc_decoder_create_multi(const ldpc_graph *g, const ldpc_decoder_config *cfg, const int32_t *devices,
                              int32_t n_devices, ldpc_decoder **out)
{
    if (!out) return fail(LDPC_ERR_ARG, "out is NULL");
    *out = nullptr;
    if (!g || !cfg) return fail(LDPC_ERR_ARG, "graph/config is NULL");
    if (!devices || n_devices <= 0 || n_devices > 64) return fail(LDPC_ERR_ARG, "devices[] must hold 1..64 ordinals");
    ldpc_decoder *grp = new (std::nothrow) ldpc_decoder;
    if (!grp) return fail(LDPC_ERR_NOMEM, "out of memory");
    std::unique_ptr<ldpc_decoder> guard(grp);
    for (int32_t i = 0; i < n_devices; ++i) {
        ldpc_decoder_config c = *cfg;
        c.device = devices[i];
        ldpc_decoder *sh = nullptr;
        const int rc = ldpc_decoder_create(g, &c, &sh);
        if (rc) return rc;                       /* the guard destroys the shards made so far */
        grp->shards.push_back(sh);
        /* the host thread that runs this device's frame range in every ldpc_decode of the handle */
        grp->shard_workers.emplace_back(new (std::nothrow) ldpc::Worker([] { return g_err; }));
        if (!grp->shard_workers.back() || !grp->shard_workers.back()->start())
            return fail(LDPC_ERR_NOMEM, "cannot start the host thread of device-list entry %d", i);
    }
    grp->cfg = *cfg;
    grp->cfg.device = devices[0];
    grp->M = g->M; grp->N = g->N; grp->E = g->E;
    *out = guard.release();
    return LDPC_OK;
}

int ldpc_decode_device(ldpc_decoder *d, const float *llr_dev, int64_t frames, uint8_t *out_dev,
                       int64_t out_bytes, int32_t *iters_dev, void *stream)
{
    if (!d) return fail(LDPC_ERR_ARG, "decoder is NULL");
    if (!d->shards.empty())
        return fail(LDPC_ERR_STATE, "a multi-device handle decodes host buffers only (ldpc_decode): device "
                    "pointers belong to one device");
    if (frames < 0 || frames > d->cfg.max_batch)
        return fail(LDPC_ERR_ARG, "frames=%lld outside [0, max_batch=%d]", (long long)frames,
                    d->cfg.max_batch);
    if (frames == 0) { d->last_frames = 0; d->have_last = false; return LDPC_OK; }
    if (!llr_dev) return fail(LDPC_ERR_ARG, "llr is NULL");
    const int64_t need = ldpc_out_bytes(d->cfg.K, frames, d->cfg.pack_mode);
    if (out_dev && out_bytes < need && d->cfg.pack_mode == LDPC_PACK_BITS)
        return fail(LDPC_ERR_ARG, "out_bytes=%lld < %lld", (long long)out_bytes, (long long)need);
    HIP_TRY(hipSetDevice(d->cfg.device));
    hipStream_t s = (hipStream_t)stream;
    d->timing = d->timing_every > 0 && (d->timing_calls++ % d->timing_every) == 0;
    d->last_stream = s;
    d->last_frames = frames;
    d->call.valid = false;          /* ldpc_decode() sets it again once all its groups are in */
    HIP_TRY(hipEventRecord(d->ev_begin, s));
    /* gaps between frames (K % 8 != 0, decodeCL.c:191-192 leaves them alone) read as 0 */
    if (out_dev) HIP_TRY(hipMemsetAsync(out_dev, 0, (size_t)std::min(out_bytes, need), s));
    int rc;
    if (d->use_fused) {
        ldpc::FusedRun run{llr_dev, frames, out_dev, std::min(out_bytes, need), iters_dev, d->cfg.K,
                           d->cfg.max_iter, d->tap_iter, d->cfg.early_term, d->summary.p,
                           d->cfg.algo == LDPC_ALGO_MS_FUSED ? 1 : (d->cfg.algo == LDPC_ALGO_MS ? 2 : (d->cfg.algo == LDPC_ALGO_SP ? 3 : 0)),
                           d->cfg.llr_scale, ldpc::tune_pick(d->tune.fused_loop, false) ? 1 : 0,
                           ldpc::tune_pick(d->tune.fused_pack, true) ? 0 : 1};
        hipError_t e = span_begin(d, s, 2, 0, (int64_t)frames * (4 * d->N + d->cfg.K / 8));
        if (e == hipSuccess)
            e = d->use_ldsp ? ldpc::engine_ldsp_run(&d->ldsp, run, s, &d->last_iterations)
                            : ldpc::engine_fused_run(&d->fused, run, s, &d->last_iterations);
        if (e == hipSuccess) e = span_end(d, s);
        rc = (e == hipSuccess) ? LDPC_OK : fail(LDPC_ERR_HIP, "fused decode: %s", hipGetErrorString(e));
    } else if (d->cfg.algo == LDPC_ALGO_LAYERED || d->cfg.algo == LDPC_ALGO_LAYERED_HOST) {
        ldpc::LayeredRun run;
        run.span_begin = [](void *c, hipStream_t st, int kind, int deg, int64_t bytes) {
            return span_begin((ldpc_decoder *)c, st, kind, deg, bytes);
        };
        run.span_end = [](void *c, hipStream_t st) { return span_end((ldpc_decoder *)c, st); };
        run.span_ctx = d;
        run.llr_dev = llr_dev; run.frames = frames; run.out_dev = out_dev;
        run.out_bytes = std::min(out_bytes, need); run.iters_dev = iters_dev;
        run.K = d->cfg.K; run.max_iter = d->cfg.max_iter; run.tap_iter = d->tap_iter;
        run.early_term = d->cfg.early_term; run.pack_mode = d->cfg.pack_mode;
        run.hard = d->hard.p; run.failw = d->failw.p; run.done = d->done.p; run.iters = d->iters.p;
        run.row_ptr = d->row_ptr.p; run.edge_col = d->edge_col.p; run.summary = d->summary.p;
        hipError_t e = ldpc::engine_layered_run(&d->layered, run, s, &d->last_iterations);
        rc = (e == hipSuccess) ? LDPC_OK
                               : fail(LDPC_ERR_HIP, "layered decode: %s", hipGetErrorString(e));
    } else if (d->V == 1) rc = run_flooding<1>(d, llr_dev, frames, out_dev, std::min(out_bytes, need), iters_dev, s);
    else if (d->V == 2) rc = run_flooding<2>(d, llr_dev, frames, out_dev, std::min(out_bytes, need), iters_dev, s);
    else rc = run_flooding<4>(d, llr_dev, frames, out_dev, std::min(out_bytes, need), iters_dev, s);
    if (rc) return rc;
    HIP_TRY(hipEventRecord(d->ev_end, s));
    d->have_last = true;
    return LDPC_OK;
}

/* ------------------------------------------------------------ host-buffer path
 * ldpc_decode: the reference's Coder::decode signature (MyLdpc.cpp:571-618; its copies are the blocking
 * enqueueWriteBuffer / enqueueReadBuffer of :796 and :988).  What moves the caller's channel values is
 * chosen per call (enum ldpc_host_input, include/ldpc_hip.h):
 *   direct -- the caller has page-locked the buffer itself: plain asynchronous copies;
 *   staged -- the default: worker threads of the handle copy each launch group through a ring of this
 *             library's own pinned chunks; the HIP runtime never sees the caller's pointer;
 *   lock   -- opt-in: whole pages strictly inside the call's byte range are page-locked for the call
 *             and read in place by the copy engine (host_stage.hpp: plan_group_blocks, PageLockRegistry).
 * Groups of up to kStageBytes are copied by the calling thread into the slot's pinned scratch in the
 * last two modes (one frame of the (648, 324) code is 2.6 KB: no thread hop on the latency path). */
namespace {

constexpr size_t kStageBytes = (size_t)4 << 20;
constexpr size_t kRingChunk = (size_t)8 << 20;
constexpr int kRingChunks = 4;

enum InputMode { kInputDirect = 0, kInputStaged = 1, kInputLock = 2 };

InputMode resolve_input_mode(const ldpc_decoder_config &cfg, const void *p, size_t bytes)
{
    if (ldpc::PageLockRegistry::instance().caller_locked(p, (const uint8_t *)p + bytes - 1)) return kInputDirect;
    return cfg.host_input == LDPC_HOST_INPUT_LOCK_PAGES ? kInputLock : kInputStaged;
}

/* the pinned ring, the stager thread and its copy helpers: made once, by the first call that needs them */
int ensure_stager(ldpc_decoder *d)
{
    if (d->stager) return LDPC_OK;
    if (d->ring.empty()) d->ring.resize(kRingChunks);
    for (auto &c : d->ring) {
        if (!c.h) HIP_TRY(hipHostMalloc((void **)&c.h, kRingChunk, hipHostMallocDefault));
        if (!c.ev) HIP_TRY(hipEventCreateWithFlags(&c.ev, hipEventDisableTiming));
    }
    const int threads = d->cfg.host_copy_threads > 0 ? d->cfg.host_copy_threads : 4;
    while ((int)d->copy_helpers.size() < threads - 1) {
        std::unique_ptr<ldpc::Worker> w(new (std::nothrow) ldpc::Worker([] { return g_err; }));
        if (!w || !w->start()) return fail(LDPC_ERR_NOMEM, "cannot start a copy thread");
        d->copy_helpers.push_back(std::move(w));
    }
    d->copy_jobs.resize(d->copy_helpers.size());
    std::unique_ptr<ldpc::Worker> st(new (std::nothrow) ldpc::Worker([] { return g_err; }));
    if (!st || !st->start()) return fail(LDPC_ERR_NOMEM, "cannot start the staging thread");
    d->stager = std::move(st);
    return LDPC_OK;
}

/* n bytes into a pinned chunk, the helpers taking equal page-aligned parts */
void ring_fill(ldpc_decoder *d, uint8_t *dst, const uint8_t *src, size_t n)
{
    const size_t parts = d->copy_helpers.size() + 1;
    if (parts == 1 || n < ((size_t)1 << 20)) { memcpy(dst, src, n); return; }
    const size_t per = ((n + parts - 1) / parts + 4095) & ~(size_t)4095;
    size_t used = 0;
    for (size_t i = 0; i < d->copy_helpers.size(); ++i) {
        const size_t lo = (i + 1) * per;
        if (lo >= n) break;
        const size_t len = std::min(per, n - lo);
        d->copy_jobs[i].fn = [dst, src, lo, len]() -> int { memcpy(dst + lo, src + lo, len); return 0; };
        d->copy_helpers[i]->submit(&d->copy_jobs[i]);
        ++used;
    }
    memcpy(dst, src, std::min(per, n));
    for (size_t i = 0; i < used; ++i) (void)d->copy_helpers[i]->wait(&d->copy_jobs[i]);
}

/* `bytes` from pageable memory to the device through the ring, on the copy stream.  One thread at a
 * time per decoder (the stager thread; in lock mode the calling thread, for a block that could not be
 * locked).  A chunk is reused once the copy that read it has completed. */
hipError_t staged_copy(ldpc_decoder *d, uint8_t *dst, const uint8_t *src, size_t bytes)
{
    for (size_t o = 0; o < bytes; o += kRingChunk) {
        const size_t n = std::min(kRingChunk, bytes - o);
        auto &c = d->ring[d->ring_next++ % d->ring.size()];
        hipError_t e = c.used ? hipEventSynchronize(c.ev) : hipSuccess;
        if (e != hipSuccess) return e;
        ring_fill(d, c.h, src + o, n);
        e = hipMemcpyAsync(dst + o, c.h, n, hipMemcpyHostToDevice, d->copy_stream);
        if (e == hipSuccess) e = hipEventRecord(c.ev, d->copy_stream);
        if (e != hipSuccess) return e;
        c.used = true;
    }
    return hipSuccess;
}

/* ldpc_decode on ONE device; `mode` was decided once per ldpc_decode call, before any thread of a device
 * list has touched the buffer. */
int decode_host(ldpc_decoder *d, const float *llr_host, int64_t frames, uint8_t *out_host,
                int64_t out_bytes, int32_t *iters, InputMode mode)
{
    const int64_t total = ldpc_out_bytes(d->cfg.K, frames, d->cfg.pack_mode);
    HIP_TRY(hipSetDevice(d->cfg.device));
    int64_t B = d->cfg.max_batch;
    /* A large call that is ONE launch group is cut into two: the second half's channel values travel while the first
     * half is decoded (one group exposes its whole copy: 21 ms of PCIe in front of a 94 ms decode for 4096 frames of the
     * headline code; half batches decode at 0.99 of the full batch's rate).  Only where the grouping cannot be seen in
     * the output: K a multiple of 8 (MyLdpc.cpp:577-616 starts every group at byte off*K/8). */
    if (frames <= B && frames >= 2048 && d->cfg.K % 8 == 0 &&
        (size_t)frames * d->N * sizeof(float) >= ((size_t)256 << 20))
        B = ((frames + 1) / 2 + 255) / 256 * 256;
    if (d->cfg.pack_mode == LDPC_PACK_BITS && (d->cfg.K % 8) && frames > B)
        return fail(LDPC_ERR_UNSUPPORTED, "bit-packed output with K %% 8 != 0 cannot be chunked: "
                    "raise max_batch to cover all %lld frames", (long long)frames);
    const int64_t Bmax = d->cfg.max_batch;      /* the slots hold a full group whatever this call's groups are */
    const int64_t stage_out = ldpc_out_bytes(d->cfg.K, Bmax, d->cfg.pack_mode) + 8;
    /* more than one group: three staging slots, so that group k+1's channel values are copied in while
     * group k is decoded (the host may block in group k's early-termination polls) and group k-1's
     * results are copied out */
    const int nslots = frames > B ? 3 : 1;
    const int64_t ngroups = (frames + B - 1) / B;
    if (!d->copy_stream) HIP_TRY(hipStreamCreateWithFlags(&d->copy_stream, hipStreamNonBlocking));
    for (int i = 0; i < nslots; ++i) {
        auto &sl = d->slot[i];
        if (!sl.llr.p) HIP_TRY(sl.llr.alloc((size_t)Bmax * d->N));
        if (!sl.out.p) HIP_TRY(sl.out.alloc((size_t)stage_out));
        if (!sl.iters.p) HIP_TRY(sl.iters.alloc((size_t)Bmax));
        if (!sl.h_out) HIP_TRY(hipHostMalloc((void **)&sl.h_out, (size_t)stage_out, hipHostMallocDefault));
        if (!sl.h_iters) HIP_TRY(hipHostMalloc((void **)&sl.h_iters, (size_t)Bmax * sizeof(int32_t), hipHostMallocDefault));
        if (!sl.h_head) HIP_TRY(hipHostMalloc((void **)&sl.h_head, kStageBytes, hipHostMallocDefault));
        if (!sl.h_sum) HIP_TRY(hipHostMalloc((void **)&sl.h_sum, 16 * sizeof(int32_t), hipHostMallocDefault));
        if (!sl.h2d_done) HIP_TRY(hipEventCreateWithFlags(&sl.h2d_done, hipEventDisableTiming));
        if (!sl.all_done) HIP_TRY(hipEventCreateWithFlags(&sl.all_done, hipEventDisableTiming));
    }
    if (mode != kInputDirect && (size_t)std::min(B, frames) * d->N * sizeof(float) > kStageBytes) {
        const int rs = ensure_stager(d);
        if (rs) return rs;
    }
    ldpc::PageLockRegistry &registry = ldpc::PageLockRegistry::instance();
    /* a finished group's bytes go from the pinned slot to the caller's buffers */
    ldpc_decoder::CallCounts counts;
    const bool flooding_counts = !d->use_fused && d->cfg.algo != LDPC_ALGO_LAYERED && d->cfg.algo != LDPC_ALGO_LAYERED_HOST;
    auto drain = [&](ldpc_decoder::HostSlot &sl) -> int {
        if (!sl.busy) return LDPC_OK;
        sl.busy = false;
        HIP_TRY(hipEventSynchronize(sl.all_done));
        if (sl.copy_bytes > 0) memcpy(out_host + sl.dst, sl.h_out, (size_t)sl.copy_bytes);
        if (iters) memcpy(iters + sl.off, sl.h_iters, (size_t)sl.n * sizeof(int32_t));
        /* the call's counts: sums over its groups, maxima for the iteration numbers (as ldpc_decoder_stats forms them) */
        counts.frames += sl.n;
        counts.converged += sl.h_sum[1];
        counts.batch_time = std::max(counts.batch_time, sl.h_sum[0]);
        counts.iterations = std::max(counts.iterations, sl.g_iterations);
        if (flooding_counts) {
            counts.frame_rounds += d->cfg.early_term ? (int64_t)sl.h_sum[2] * d->F : (int64_t)sl.g_iterations * sl.g_tiles * d->F;
            for (int c = 0; c < sl.g_children; ++c) counts.frame_rounds += (int64_t)sl.h_sum[4 * (c + 1) + 2] * sl.g_child_f[c];
        }
        return LDPC_OK;
    };
    int rc = LDPC_OK;
#ifdef LDPC_TRACE_HOST
    const auto t_start = std::chrono::steady_clock::now();
#define LDPC_STAMP(what, kk)                                                                             \
    fprintf(stderr, "[ldpc_decode] %8.2f ms  group %lld  %s\n",                                          \
            std::chrono::duration<double, std::milli>(std::chrono::steady_clock::now() - t_start).count(), \
            (long long)(kk), what)
#else
#define LDPC_STAMP(what, kk) ((void)0)
#endif
    /* Coder::decode, MyLdpc.cpp:577-616: groups of batchSize frames, last one short.
     * stage_in(k): group k's channel values -> slot k % nslots, on the copy stream (a copy from pageable
     * memory handed to the runtime as it is would wait for the device's other work -- measured: 151 ms
     * behind a 140 ms decode instead of 19 ms -- so the input never travels that way). */
    auto stage_in = [&](int64_t kk) -> int {
        const int si = (int)(kk % nslots);
        auto &sl = d->slot[si];
        const int r1 = drain(sl);                /* the slot's previous tenant (group kk - nslots) */
        if (r1) return r1;
        const int64_t off = kk * B, n = std::min(B, frames - off);
        const uint8_t *src = reinterpret_cast<const uint8_t *>(llr_host + (size_t)off * d->N);
        const size_t bytes = (size_t)n * d->N * sizeof(float);
        uint8_t *dst = reinterpret_cast<uint8_t *>(sl.llr.p);
        hipError_t e = hipSuccess;
        if (mode == kInputDirect) {
            e = hipMemcpyAsync(dst, src, bytes, hipMemcpyDefault, d->copy_stream);
        } else if (bytes <= kStageBytes) {
            memcpy(sl.h_head, src, bytes);
            e = hipMemcpyAsync(dst, sl.h_head, bytes, hipMemcpyHostToDevice, d->copy_stream);
        } else if (mode == kInputStaged) {
            ldpc::Job &job = d->stage_job[si];
            hipEvent_t done = sl.h2d_done;
            job.fn = [d, dst, src, bytes, done]() -> int {
                hipError_t je = hipSetDevice(d->cfg.device);
                if (je == hipSuccess) je = staged_copy(d, dst, src, bytes);
                if (je == hipSuccess) je = hipEventRecord(done, d->copy_stream);
                return je == hipSuccess ? LDPC_OK
                                        : fail(LDPC_ERR_HIP, "staging through the pinned ring: %s", hipGetErrorString(je));
            };
            d->stager->submit(&job);
            d->stage_pending[si] = true;
            LDPC_STAMP("staging submitted", kk);
            return LDPC_OK;                      /* the job records h2d_done */
        } else {
            const ldpc::GroupBlocks gb = ldpc::plan_group_blocks((uintptr_t)llr_host, frames, d->N, B, kk);
            bool locked = false;
            if (!gb.whole_by_cpu) {
                bool overlap = false;
                locked = registry.lock((void *)gb.b0, (size_t)(gb.b1 - gb.b0), &overlap) == hipSuccess;
                if (locked) d->locked_blocks.push_back((void *)gb.b0);
            }
            if (!locked) {
                e = staged_copy(d, dst, src, bytes);      /* somebody else holds these pages: stage */
            } else {
                const size_t head = (size_t)(gb.b0 - gb.s0), body = (size_t)(gb.body_end - gb.b0),
                             tail = (size_t)(gb.s1 - gb.body_end);
                if (head) {
                    memcpy(sl.h_head, src, head);
                    e = hipMemcpyAsync(dst, sl.h_head, head, hipMemcpyHostToDevice, d->copy_stream);
                }
                if (e == hipSuccess)
                    e = hipMemcpyAsync(dst + head, (const void *)gb.b0, body, hipMemcpyHostToDevice, d->copy_stream);
                if (e == hipSuccess && tail) {
                    memcpy(sl.h_head + ldpc::kPage, (const void *)gb.body_end, tail);
                    e = hipMemcpyAsync(dst + head + body, sl.h_head + ldpc::kPage, tail, hipMemcpyHostToDevice, d->copy_stream);
                }
            }
        }
        if (e == hipSuccess) e = hipEventRecord(sl.h2d_done, d->copy_stream);
        if (e != hipSuccess) return fail(LDPC_ERR_HIP, "host-to-device staging: %s", hipGetErrorString(e));
        LDPC_STAMP("H2D enqueued", kk);
        return LDPC_OK;
    };
    /* the staging job of slot si has run: its copies and h2d_done are on the copy stream */
    auto staged_ready = [&](int si) -> int {
        if (!d->stage_pending[si]) return LDPC_OK;
        d->stage_pending[si] = false;
        const int r = d->stager->wait(&d->stage_job[si]);
        if (r) g_err = d->stage_job[si].err;
        return r;
    };
    rc = stage_in(0);
    for (int64_t k = 0; k < ngroups && rc == LDPC_OK; ++k) {
        const int si = (int)(k % nslots);
        auto &sl = d->slot[si];
        const int64_t off = k * B, n = std::min(B, frames - off);
        if (k + 1 < ngroups && (rc = stage_in(k + 1))) break;   /* runs beside this group's decode */
        if ((rc = staged_ready(si))) break;
        hipError_t e = hipStreamWaitEvent(d->stream, sl.h2d_done, 0);
        if (e != hipSuccess) { rc = fail(LDPC_ERR_HIP, "host-to-device staging: %s", hipGetErrorString(e)); break; }
        const int64_t chunk_bytes = ldpc_out_bytes(d->cfg.K, n, d->cfg.pack_mode);
        rc = ldpc_decode_device(d, sl.llr.p, n, sl.out.p, chunk_bytes, iters ? sl.iters.p : nullptr, d->stream);
        if (rc) break;
        LDPC_STAMP("decode enqueued", k);
        /* byte offset of this group's first frame: (off*K)/8 in both packings */
        sl.off = off; sl.n = n;
        sl.dst = off * (int64_t)d->cfg.K / 8;
        sl.copy_bytes = std::max<int64_t>(0, std::min(std::min(out_bytes, total) - sl.dst, chunk_bytes));
        if (sl.copy_bytes > 0)
            e = hipMemcpyAsync(sl.h_out, sl.out.p, (size_t)sl.copy_bytes, hipMemcpyDeviceToHost, d->stream);
        if (e == hipSuccess && iters)
            e = hipMemcpyAsync(sl.h_iters, sl.iters.p, (size_t)n * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream);
        sl.g_iterations = d->last_iterations; sl.g_tiles = d->last_tiles; sl.g_children = 0;
        if (e == hipSuccess) e = hipMemcpyAsync(sl.h_sum, d->summary.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream);
        for (const ldpc_decoder *p = d->handed_to; p && sl.g_children < 3 && e == hipSuccess; p = p->handed_to) {
            sl.g_child_f[sl.g_children] = p->F;
            e = hipMemcpyAsync(sl.h_sum + 4 * (sl.g_children + 1), p->summary.p, 4 * sizeof(int32_t), hipMemcpyDeviceToHost, d->stream);
            ++sl.g_children;
        }
        if (e == hipSuccess) e = hipEventRecord(sl.all_done, d->stream);
        if (e != hipSuccess) { rc = fail(LDPC_ERR_HIP, "device-to-host staging: %s", hipGetErrorString(e)); break; }
        sl.busy = true;
    }
#undef LDPC_STAMP
    std::string first_error = rc ? g_err : std::string();
    /* every exit path: no staging job still reads the caller's buffer, nothing of this call is in flight */
    for (int i = 0; i < nslots; ++i) {
        const int r2 = staged_ready(i);
        if (rc == LDPC_OK && r2) { rc = r2; first_error = g_err; }
    }
    for (int i = 0; i < nslots; ++i) {          /* oldest first: slot (ngroups % nslots) was filled earliest */
        const int r2 = drain(d->slot[(ngroups + i) % nslots]);
        if (rc == LDPC_OK && r2) { rc = r2; first_error = g_err; }
    }
    hipError_t es = hipStreamSynchronize(d->copy_stream);
    const hipError_t es2 = hipStreamSynchronize(d->stream);
    if (es == hipSuccess) es = es2;
    if (es != hipSuccess && rc == LDPC_OK) {
        rc = fail(LDPC_ERR_HIP, "ldpc_decode: draining the streams: %s", hipGetErrorString(es));
        first_error = g_err;
    }
    /* lock mode: the pages go back to the caller; a block that cannot be released stays on record */
    for (void *p : d->locked_blocks) {
        const hipError_t eu = registry.unlock(p);
        if (eu == hipSuccess) continue;
        d->stuck_blocks.push_back(p);
        if (rc == LDPC_OK) {
            rc = fail(LDPC_ERR_HIP, "hipHostUnregister(%p) failed: %s -- the block stays page-locked and on this "
                      "library's record", p, hipGetErrorString(eu));
            first_error = g_err;
        }
    }
    d->locked_blocks.clear();
    if (!first_error.empty()) g_err = first_error;
    counts.valid = rc == LDPC_OK && ngroups > 1;       /* one group: the decoder's own record is the call's */
    d->call = counts;
    return rc;
}

/* Shard boundaries that keep a multi-device result byte-identical to the single-device one: with
 * K % 8 != 0 a frame's first byte is (frame*K)/8 with the division applied per launch group
 * (MyLdpc.cpp:577-616 passes &srcCode[off*K/8]), so ranges must start where frame*K is a multiple
 * of 8 -- and on a group boundary once the stream is longer than one group. */
int32_t shard_unit(const ldpc_decoder_config &cfg, int64_t frames)
{
    if (cfg.K % 8 == 0) return 1;
    int64_t u = 8;
    while (u > 1 && ((u / 2) * (int64_t)cfg.K) % 8 == 0) u /= 2;
    if (frames > cfg.max_batch) {
        int64_t a = u, b = cfg.max_batch;
        while (b) { const int64_t t = a % b; a = b; b = t; }
        u = u / a * cfg.max_batch;               /* lcm(u, max_batch) */
    }
    return (int32_t)std::min<int64_t>(u, 0x7fffffff);
}

}  // namespace

int ldpc_decode(ldpc_decoder *d, const float *llr_host, int64_t frames, uint8_t *out_host,
                int64_t out_bytes, int32_t *iters)
{
    if (!d) return fail(LDPC_ERR_ARG, "decoder is NULL");
    if (frames < 0) return fail(LDPC_ERR_ARG, "frames < 0");
    if (frames == 0) return LDPC_OK;
    if (!llr_host || !out_host) return fail(LDPC_ERR_ARG, "llr/out is NULL");
    if (out_bytes < 0) return fail(LDPC_ERR_ARG, "out_bytes < 0");
    /* asked once, before any thread touches the buffer */
    const InputMode mode = resolve_input_mode(d->cfg, llr_host, (size_t)frames * d->N * sizeof(float));
    if (d->shards.empty()) return decode_host(d, llr_host, frames, out_host, out_bytes, iters, mode);

    /* several devices: each entry's own host thread decodes a contiguous frame range */
    const int n = (int)d->shards.size();
    const int32_t unit = shard_unit(d->cfg, frames);
    std::vector<int64_t> lo((size_t)n), hi((size_t)n);
    for (int i = 0; i < n; ++i) {
        const int rc = ldpc_shard_range(frames, i, n, unit, &lo[i], &hi[i]);
        if (rc) return rc;
    }
    std::vector<ldpc::Job> jobs((size_t)n);
    for (int i = 0; i < n; ++i) {
        ldpc_decoder *sh = d->shards[i];
        sh->have_last = false;
        if (hi[i] <= lo[i]) continue;
        const int64_t base = lo[i] * (int64_t)d->cfg.K / 8;      /* exact: lo is a multiple of the unit */
        const int64_t room = std::max<int64_t>(0, out_bytes - base);
        const int64_t lo_i = lo[i], cnt = hi[i] - lo[i];
        const int64_t obytes = std::min(room, ldpc_out_bytes(d->cfg.K, cnt, d->cfg.pack_mode));
        const int32_t N = d->N;
        jobs[i].fn = [sh, llr_host, out_host, iters, lo_i, cnt, base, obytes, N, mode]() -> int {
            return decode_host(sh, llr_host + (size_t)lo_i * N, cnt, out_host + base, obytes,
                               iters ? iters + lo_i : nullptr, mode);
        };
        d->shard_workers[i]->submit(&jobs[i]);
    }
    int rc = LDPC_OK;
    for (int i = 0; i < n; ++i) {
        if (hi[i] <= lo[i]) continue;
        const int r = d->shard_workers[i]->wait(&jobs[i]);     /* all of them, also after a failure */
        if (r && rc == LDPC_OK) { rc = r; g_err = jobs[i].err; }
    }
    if (rc) return rc;
    d->have_last = true;
    d->last_frames = frames;
    return LDPC_OK;
}

int ldpc_host_block_plan(uint64_t base, int64_t frames, int32_t N, int32_t max_batch, int64_t group, uint64_t out[6])
{
    if (!out) return fail(LDPC_ERR_ARG, "out is NULL");
    if (frames <= 0 || N <= 0 || max_batch <= 0 || group < 0 || group * (int64_t)max_batch >= frames)
        return fail(LDPC_ERR_ARG, "block_plan(frames=%lld, N=%d, max_batch=%d, group=%lld)", (long long)frames, N, max_batch,
                    (long long)group);
    const ldpc::GroupBlocks g = ldpc::plan_group_blocks((uintptr_t)base, frames, N, max_batch, group);
    out[0] = g.s0; out[1] = g.s1; out[2] = g.b0; out[3] = g.b1; out[4] = g.body_end; out[5] = g.whole_by_cpu ? 1 : 0;
    return LDPC_OK;
}

int ldpc_host_locked_ranges(int64_t *live, int64_t *stale)
{
    if (live) *live = (int64_t)ldpc::PageLockRegistry::instance().live_count();
    if (stale) *stale = (int64_t)ldpc::PageLockRegistry::instance().stale_count();
    return LDPC_OK;
}

int ldpc_decoder_set_timing(ldpc_decoder *d, int enable)
{
    if (!d) return fail(LDPC_ERR_ARG, "decoder is NULL");
    if (!d->shards.empty()) {
        for (ldpc_decoder *sh : d->shards) {
            const int rc = ldpc_decoder_set_timing(sh, enable);
            if (rc) return rc;
        }
        return LDPC_OK;
    }
    if (d->have_last) {   /* events of earlier calls may still be pending */
        HIP_TRY(hipSetDevice(d->cfg.device));
        HIP_TRY(hipEventSynchronize(d->ev_end));
    }
    d->timing_every = enable > 0 ? enable : 0;
    d->timing_calls = 0;
    d->timing = false;
    d->spans_used = 0;
    return LDPC_OK;
}

int ldpc_decoder_stats(ldpc_decoder *d, ldpc_decode_stats *st)
{
    if (!d || !st) return fail(LDPC_ERR_ARG, "decoder/stats is NULL");
    memset(st, 0, sizeof *st);
    if (!d->have_last) return fail(LDPC_ERR_STATE, "no decode call to report on");
    if (!d->shards.empty()) {
        /* the devices ran side by side: counts add up, times and iteration numbers take the maximum.
         * Each device's counts cover all launch groups of its range, its times the last group. */
        for (ldpc_decoder *sh : d->shards) {
            if (!sh->have_last) continue;
            ldpc_decode_stats one;
            const int rc = ldpc_decoder_stats(sh, &one);
            if (rc) return rc;
            st->iterations_launched = std::max(st->iterations_launched, one.iterations_launched);
            st->batch_time = std::max(st->batch_time, one.batch_time);
            st->frames += one.frames;
            st->frames_converged += one.frames_converged;
            st->ms_total = std::max(st->ms_total, one.ms_total);
            st->ms_check += one.ms_check; st->ms_var += one.ms_var; st->ms_other += one.ms_other;
            st->launches_check += one.launches_check; st->launches_var += one.launches_var;
            st->frame_rounds += one.frame_rounds;
        }
        return LDPC_OK;
    }
    HIP_TRY(hipSetDevice(d->cfg.device));
    HIP_TRY(hipEventSynchronize(d->ev_end));
    st->iterations_launched = d->last_iterations;
    st->frames = d->last_frames;
    HIP_TRY(hipEventElapsedTime(&st->ms_total, d->ev_begin, d->ev_end));
    int32_t summary[4] = {0, 0, 0, 0};
    HIP_TRY(hipMemcpy(summary, d->summary.p, sizeof summary, hipMemcpyDeviceToHost));
    st->batch_time = summary[0];
    st->frames_converged = summary[1];
    /* frame-rounds the message kernels really worked on (tiles that were finished when a round began leave
     * at kernel entry): counted on the device with early termination, all launched rounds without; the
     * one-launch kernels (frames leave individually inside the launch) report 0 */
    st->frame_rounds = 0;
    if (!d->use_fused && d->cfg.algo != LDPC_ALGO_LAYERED && d->cfg.algo != LDPC_ALGO_LAYERED_HOST) {
        st->frame_rounds = d->cfg.early_term ? (int64_t)summary[2] * d->F
                                             : (int64_t)d->last_iterations * d->last_tiles * d->F;
        for (const ldpc_decoder *p = d->handed_to; p; p = p->handed_to) {      /* the child, and whom it handed over to */
            int32_t cs[4] = {0, 0, 0, 0};
            HIP_TRY(hipMemcpy(cs, p->summary.p, sizeof cs, hipMemcpyDeviceToHost));
            st->frame_rounds += (int64_t)cs[2] * p->F;
        }
    }
    if (d->call.valid) {            /* a host-buffer call of several launch groups: its counts cover all of them */
        st->iterations_launched = d->call.iterations;
        st->batch_time = d->call.batch_time;
        st->frames = d->call.frames;
        st->frames_converged = d->call.converged;
        st->frame_rounds = d->call.frame_rounds;
    }
    for (size_t i = 0; i < d->spans_used; ++i) {
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, d->spans[i].a, d->spans[i].b));
        const int kind = d->spans[i].kind;
        if (kind == 0 || kind == 4 || kind == 5) { st->ms_check += ms; ++st->launches_check; }
        else if (kind == 1 || kind == 2 || kind == 6) { st->ms_var += ms; ++st->launches_var; }
        else st->ms_other += ms;
    }
    return LDPC_OK;
}

int ldpc_decoder_kernel_times(ldpc_decoder *d, ldpc_kernel_time *out, int32_t capacity, int32_t *count)
{
    if (!d || !out || !count || capacity <= 0) return fail(LDPC_ERR_ARG, "bad arguments");
    *count = 0;
    if (!d->shards.empty()) return ldpc_decoder_kernel_times(d->shards[0], out, capacity, count);
    if (!d->have_last) return fail(LDPC_ERR_STATE, "no decode call to report on");
    HIP_TRY(hipSetDevice(d->cfg.device));
    HIP_TRY(hipEventSynchronize(d->ev_end));
    const char *phase_name[] = {"check_kernel", "var_kernel", "layer_kernel", "other",
                                d->tune_link_narrow == 2 ? "check_link_half_kernel"
                                : d->tune_link_narrow ? "check_link_narrow_kernel" : "check_link_kernel"};
    static const char *algo_name_f32[] = {"sp", "ms", "layered", "ms_fused", "layered_host"};
    static const char *algo_name_f16[] = {"sp16", "ms16", "layered16", "ms_fused16", "layered_host16"};
    const char **algo_name = d->msg_size == 2 ? algo_name_f16 : algo_name_f32;
    for (size_t i = 0; i < d->spans_used; ++i) {
        const TimedSpan &sp = d->spans[i];
        float ms = 0;
        HIP_TRY(hipEventElapsedTime(&ms, sp.a, sp.b));
        const int phase = (sp.kind == 4 || sp.kind == 5) ? 0 : (sp.kind == 6 ? 1 : sp.kind);   /* check / variable node */
        char name[64];
        if (sp.kind == 3) snprintf(name, sizeof name, "other");
        else if (d->use_fused && d->use_ldsp)   /* whole decode in one launch; [persistent grid x workgroup size, frames per workgroup] */
            snprintf(name, sizeof name, "%s[%dx%d,%d]", d->cfg.algo == LDPC_ALGO_LAYERED ? "layered_ldsp_kernel" : "flood_ldsp_kernel",
                     d->ldsp.grid, d->ldsp.block, d->ldsp.wg_frames);
        else if (d->use_fused)      /* bytes = channel values in + packed bits out */
            snprintf(name, sizeof name, "%s", d->cfg.algo == LDPC_ALGO_SP ? "fused_sp_kernel"
                     : d->cfg.algo == LDPC_ALGO_LAYERED ? "fused_layered_kernel" : "fused_flood_kernel");
        else if (sp.kind == 5 || sp.kind == 6)
            snprintf(name, sizeof name, "%s<%s,%d-%d,%d>", sp.kind == 5 ? "check_group_kernel" : "var_group_kernel",
                     algo_name[d->cfg.algo], sp.lo, sp.degree, d->V);
        else snprintf(name, sizeof name, "%s<%s,%d,%d>", phase_name[sp.kind], algo_name[d->cfg.algo], sp.degree, d->V);
        int k = 0;
        for (; k < *count; ++k)
            if (!strcmp(out[k].name, name)) break;
        if (k == *count) {
            if (*count == capacity) continue;
            ++*count;
            memset(&out[k], 0, sizeof out[k]);
            out[k].phase = phase;
            out[k].degree = sp.degree;
            memcpy(out[k].name, name, sizeof name);
        }
        ++out[k].launches;
        out[k].ms_total += ms;
        out[k].bytes_total += sp.bytes;
        out[k].bytes_moved += sp.moved;
    }
    return LDPC_OK;
}

int ldpc_decoder_link_form(ldpc_decoder *d, int32_t *form, int32_t *calibrated, float ms[3])
{
    if (!d) return fail(LDPC_ERR_ARG, "decoder is NULL");
    if (!d->shards.empty()) return ldpc_decoder_link_form(d->shards[0], form, calibrated, ms);
    bool linked = false;
    for (auto &rc : d->row_classes) linked = linked || rc.linked;
    if (form) *form = linked ? d->tune_link_narrow : -1;
    if (calibrated) *calibrated = d->link_calibrated ? 1 : 0;
    if (ms) for (int k = 0; k < 3; ++k) ms[k] = d->link_calibrated && d->link_cal_ms[k] < 1e29f ? d->link_cal_ms[k] : 0.0f;
    return LDPC_OK;
}

int ldpc_decoder_placement(ldpc_decoder *d, int32_t *candidates, int32_t *kept, float ms[16])
{
    if (!d) return fail(LDPC_ERR_ARG, "decoder is NULL");
    if (!d->shards.empty()) return ldpc_decoder_placement(d->shards[0], candidates, kept, ms);
    if (candidates) *candidates = d->place_candidates;
    if (kept) *kept = d->place_kept;
    if (ms) for (int k = 0; k < 16; ++k) ms[k] = k < d->place_candidates ? d->place_ms[k] : 0.0f;
    return LDPC_OK;
}

int ldpc_decoder_array_addresses(ldpc_decoder *d, uint64_t out[4])
{
    if (!d || !out) return fail(LDPC_ERR_ARG, "decoder/out is NULL");
    if (!d->shards.empty()) return ldpc_decoder_array_addresses(d->shards[0], out);
    out[0] = (uint64_t)(uintptr_t)d->Q.p; out[1] = (uint64_t)(uintptr_t)d->R.p;
    out[2] = (uint64_t)(uintptr_t)d->chan.p; out[3] = (uint64_t)(uintptr_t)d->hard.p;
    return LDPC_OK;
}

int ldpc_decoder_set_tap(ldpc_decoder *d, int32_t iter)
{
    if (!d) return fail(LDPC_ERR_ARG, "decoder is NULL");
    if (iter < 0) return fail(LDPC_ERR_ARG, "iter < 0");
    if (!d->shards.empty()) return fail(LDPC_ERR_STATE, "debug taps need a single-device handle");
    d->tap_iter = iter;
    return LDPC_OK;
}

int ldpc_decoder_dump(ldpc_decoder *d, int32_t which, float *host_out, int64_t count)
{
    if (!d || !host_out) return fail(LDPC_ERR_ARG, "decoder/host_out is NULL");
    if (!d->shards.empty()) return fail(LDPC_ERR_STATE, "debug taps need a single-device handle");
    if (!d->have_last) return fail(LDPC_ERR_STATE, "no decode call to dump");
    HIP_TRY(hipSetDevice(d->cfg.device));
    wait_for_own_work(d);
    const int64_t frames = d->last_frames;
    const int V = d->V, F = d->F;
    const int tiles = (int)((frames + F - 1) / F);
    if (d->use_fused && d->cfg.algo == LDPC_ALGO_SP) {
        if (!d->fused.dump_p) return fail(LDPC_ERR_STATE, "fused dump needs set_tap() before the decode");
        const int64_t per = (which == 0 || which == 1) ? d->E : d->N;
        if (which < 0 || which > 3 || count != frames * per) return fail(LDPC_ERR_ARG, "bad `which`/count");
        if (which == 3) {
            std::vector<uint8_t> b((size_t)count);
            HIP_TRY(hipMemcpy(b.data(), d->fused.dump_b, (size_t)count, hipMemcpyDeviceToHost));
            for (int64_t i = 0; i < count; ++i) host_out[i] = (float)b[i];
            return LDPC_OK;
        }
        const float *src = which == 0 ? d->fused.dump_r : (which == 1 ? d->fused.dump_q : d->fused.dump_p);
        HIP_TRY(hipMemcpy(host_out, src, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
        return LDPC_OK;
    }
    if (d->use_fused) {
        const float *dump_r = d->use_ldsp ? d->ldsp.dump_r : d->fused.dump_r;
        const float *dump_p = d->use_ldsp ? d->ldsp.dump_p : d->fused.dump_p;
        const float *src = which == 0 ? dump_r : (which == 2 ? dump_p : nullptr);
        const int64_t per = which == 0 ? d->E : d->N;
        if (which == 3) {           /* hard bits = P < 0 */
            if (!dump_p || count != frames * d->N) return fail(LDPC_ERR_ARG, "fused dump needs set_tap() and count = frames*N");
            HIP_TRY(hipMemcpy(host_out, dump_p, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
            const bool notpos = d->cfg.algo == LDPC_ALGO_MS;      /* MS chain: bit = !(p > 0) */
            for (int64_t i = 0; i < count; ++i)
                host_out[i] = (notpos ? !(host_out[i] > 0.0f) : (host_out[i] < 0.0f)) ? 1.0f : 0.0f;
            return LDPC_OK;
        }
        if (!src || count != frames * per) return fail(LDPC_ERR_ARG, "fused dump: set_tap() first; which in {0,2,3}");
        HIP_TRY(hipMemcpy(host_out, src, (size_t)count * sizeof(float), hipMemcpyDeviceToHost));
        return LDPC_OK;
    }
    if (d->cfg.algo == LDPC_ALGO_LAYERED || d->cfg.algo == LDPC_ALGO_LAYERED_HOST) {
        hipError_t e = ldpc::layered_dump(&d->layered, which, host_out, count, frames, d->hard.p,
                                          d->h_cols.data());
        if (e == hipErrorInvalidValue) return fail(LDPC_ERR_ARG, "bad `which`/count for layered dump");
        if (e != hipSuccess) return fail(LDPC_ERR_HIP, "layered dump: %s", hipGetErrorString(e));
        return LDPC_OK;
    }
    if (which == 0 || which == 1 || which == 2) {
        const int64_t per = (which == 2) ? d->N : d->E;
        if (count != frames * per) return fail(LDPC_ERR_ARG, "count must be frames*%lld", (long long)per);
        const uint8_t *src = which == 0 ? d->R.p : (which == 1 ? d->Q.p : d->chan.p);
        const size_t esz = (size_t)d->msg_size;
        std::vector<uint8_t> tile((size_t)per * F * esz);
        for (int t = 0; t < tiles; ++t) {
            HIP_TRY(hipMemcpy(tile.data(), src + (size_t)t * per * F * esz, tile.size(), hipMemcpyDeviceToHost));
            for (int fi = 0; fi < F; ++fi) {
                const int64_t f = (int64_t)t * F + fi;
                if (f >= frames) break;
                for (int64_t i = 0; i < per; ++i) {
                    const size_t ir = (which == 1 && !d->h_qpos.empty()) ? (size_t)d->h_qpos[(size_t)i] : (size_t)i;   /* Q: slot of edge i */
                    if (esz == 4) {
                        memcpy(&host_out[f * per + i], &tile[(ir * F + fi) * 4], 4);
                    } else {
                        _Float16 h;
                        memcpy(&h, &tile[(ir * F + fi) * 2], 2);
                        host_out[f * per + i] = (float)h;
                    }
                }
            }
        }
        return LDPC_OK;
    }
    if (which == 3) {
        if (count != frames * d->N) return fail(LDPC_ERR_ARG, "count must be frames*N");
        std::vector<uint64_t> w((size_t)tiles * d->N * V);
        HIP_TRY(hipMemcpy(w.data(), d->hard.p, w.size() * sizeof(uint64_t), hipMemcpyDeviceToHost));
        for (int64_t f = 0; f < frames; ++f) {
            const int64_t t = f / F;
            const int fi = (int)(f % F);
            for (int32_t n = 0; n < d->N; ++n)
                host_out[f * d->N + n] =
                    (float)((w[((size_t)t * d->N + n) * V + fi % V] >> (fi / V)) & 1ull);
        }
        return LDPC_OK;
    }
    return fail(LDPC_ERR_ARG, "unknown `which` %d", which);
}

int ldpc_awgn_device(float *llr_dev, int64_t frames, int32_t N, const uint8_t *bits_dev, float sd,
                     uint64_t seed, int64_t first_frame, int32_t device, void *stream)
{
    if (!llr_dev) return fail(LDPC_ERR_ARG, "llr_dev is NULL");
    if (frames < 0 || N <= 0 || first_frame < 0) return fail(LDPC_ERR_ARG, "frames=%lld, N=%d, first_frame=%lld",
                                                             (long long)frames, N, (long long)first_frame);
    if (!(sd >= 0.0f)) return fail(LDPC_ERR_ARG, "sd must be >= 0");
    if (frames == 0) return LDPC_OK;
    HIP_TRY(hipSetDevice(device));
    const int32_t groups = (N + 3) / 4;
    const int64_t threads = frames * groups;
    const int64_t blocks = (threads + 255) / 256;
    if (blocks > 0x7fffffffLL) return fail(LDPC_ERR_ARG, "too many samples for one call");
    ldpc::awgn_kernel<<<(unsigned)blocks, 256, 0, (hipStream_t)stream>>>(llr_dev, bits_dev, frames, N, groups, sd,
                                                                         seed, first_frame);
    HIP_TRY(hipGetLastError());
    return LDPC_OK;
}

int ldpc_count_errors_device(const uint8_t *out_dev, const uint8_t *ref_dev, int64_t frames,
                             int64_t bytes_per_frame, int64_t errors[3], int32_t device, void *stream)
{
    if (!out_dev || !errors) return fail(LDPC_ERR_ARG, "out_dev/errors is NULL");
    if (frames < 0 || bytes_per_frame <= 0 || frames > 0x7fffffffLL) return fail(LDPC_ERR_ARG, "bad frames/bytes_per_frame");
    errors[0] = errors[1] = errors[2] = 0;
    if (frames == 0) return LDPC_OK;
    HIP_TRY(hipSetDevice(device));
    hipStream_t s = (hipStream_t)stream;
    unsigned long long *totals = nullptr;
    HIP_TRY(hipMalloc((void **)&totals, 3 * sizeof(unsigned long long)));
    hipError_t e = hipMemsetAsync(totals, 0, 3 * sizeof(unsigned long long), s);
    if (e == hipSuccess) {
        ldpc::count_errors_kernel<<<(unsigned)frames, 256, 0, s>>>(out_dev, ref_dev, frames, bytes_per_frame, totals);
        e = hipGetLastError();
    }
    unsigned long long h[3] = {0, 0, 0};
    if (e == hipSuccess) e = hipMemcpyAsync(h, totals, sizeof h, hipMemcpyDeviceToHost, s);
    if (e == hipSuccess) e = hipStreamSynchronize(s);
    (void)hipFree(totals);
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "count_errors: %s", hipGetErrorString(e));
    for (int i = 0; i < 3; ++i) errors[i] = (int64_t)h[i];
    return LDPC_OK;
}

/* ---- measurement aid: what this box's HBM sustains right now (a float4 copy: the figure the
 *      microarchitecture guide quotes as achievable, 6.3 of 8.0 TB/s), with the default cache policy
 *      and with the non-temporal one the streaming kernels use; the better of the two ------------ */

int ldpc_hbm_probe_device(int32_t device, int64_t bytes, int32_t reps, double *copy_gbs, double *by_policy)
{
    if (!copy_gbs) return fail(LDPC_ERR_ARG, "copy_gbs is NULL");
    *copy_gbs = 0.0;
    if (by_policy) by_policy[0] = by_policy[1] = 0.0;
    if (bytes < (1 << 20) || bytes > ((int64_t)16 << 30) || reps <= 0 || reps > 1000)
        return fail(LDPC_ERR_ARG, "probe: bytes in [1 MiB, 16 GiB], reps in [1, 1000]");
    HIP_TRY(hipSetDevice(device));
    const size_t n4 = (size_t)bytes / sizeof(ldpc::vf4);
    ldpc::vf4 *src = nullptr, *dst = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipMalloc((void **)&src, n4 * sizeof(ldpc::vf4));
    if (e == hipSuccess) e = hipMalloc((void **)&dst, n4 * sizeof(ldpc::vf4));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    if (e == hipSuccess) e = hipMemsetAsync(src, 0x3c, n4 * sizeof(ldpc::vf4), s);
    float best_ms[2] = {0.0f, 0.0f};                             /* default policy, non-temporal */
    if (e == hipSuccess) {
        const unsigned grid = (unsigned)std::min<size_t>((n4 + 1023) / 1024, 256 * 64);
        hbm_probe_copy_kernel<false><<<grid, 256, 0, s>>>(src, dst, n4);        /* warm-up */
        for (int r = 0; r < 2 * reps && e == hipSuccess; ++r) {
            e = hipEventRecord(a, s);
            if (r & 1) hbm_probe_copy_kernel<true><<<grid, 256, 0, s>>>(src, dst, n4);
            else hbm_probe_copy_kernel<false><<<grid, 256, 0, s>>>(src, dst, n4);
            if (e == hipSuccess) e = hipEventRecord(b, s);
            if (e == hipSuccess) e = hipEventSynchronize(b);
            float ms = 0.0f;
            if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
            if (e == hipSuccess && (best_ms[r & 1] == 0.0f || ms < best_ms[r & 1])) best_ms[r & 1] = ms;
        }
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (s) (void)hipStreamDestroy(s);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "hbm probe: %s", hipGetErrorString(e));
    for (int k = 0; k < 2; ++k) {
        const double gbs = best_ms[k] > 0.0f ? 2.0 * (double)(n4 * sizeof(ldpc::vf4)) / (best_ms[k] * 1e-3) / 1e9 : 0.0;
        if (by_policy) by_policy[k] = gbs;
        if (gbs > *copy_gbs) *copy_gbs = gbs;
    }
    return LDPC_OK;
}

/* The same non-temporal copy, back to back for `milliseconds`: what the box sustains (its memory throttles
 * under load at times: a burst of a few launches does not see that). */
int ldpc_hbm_sustained_device(int32_t device, int64_t bytes, int32_t milliseconds, double *copy_gbs)
{
    if (!copy_gbs) return fail(LDPC_ERR_ARG, "copy_gbs is NULL");
    *copy_gbs = 0.0;
    if (bytes < (1 << 20) || bytes > ((int64_t)16 << 30) || milliseconds < 1 || milliseconds > 10000)
        return fail(LDPC_ERR_ARG, "sustained probe: bytes in [1 MiB, 16 GiB], milliseconds in [1, 10000]");
    HIP_TRY(hipSetDevice(device));
    const size_t n4 = (size_t)bytes / sizeof(ldpc::vf4);
    ldpc::vf4 *src = nullptr, *dst = nullptr;
    hipStream_t s = nullptr;
    hipEvent_t a = nullptr, b = nullptr;
    hipError_t e = hipMalloc((void **)&src, n4 * sizeof(ldpc::vf4));
    if (e == hipSuccess) e = hipMalloc((void **)&dst, n4 * sizeof(ldpc::vf4));
    if (e == hipSuccess) e = hipStreamCreateWithFlags(&s, hipStreamNonBlocking);
    if (e == hipSuccess) e = hipEventCreate(&a);
    if (e == hipSuccess) e = hipEventCreate(&b);
    if (e == hipSuccess) e = hipMemsetAsync(src, 0x3c, n4 * sizeof(ldpc::vf4), s);
    const unsigned grid = (unsigned)std::min<size_t>((n4 + 1023) / 1024, 256 * 64);
    /* one launch's time from a short burst, then a third of the time untimed and two thirds timed */
    float one_ms = 0.0f;
    if (e == hipSuccess) {
        hbm_probe_copy_kernel<true><<<grid, 256, 0, s>>>(src, dst, n4);
        e = hipEventRecord(a, s);
        for (int r = 0; r < 4; ++r) hbm_probe_copy_kernel<true><<<grid, 256, 0, s>>>(src, dst, n4);
        if (e == hipSuccess) e = hipEventRecord(b, s);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        if (e == hipSuccess) e = hipEventElapsedTime(&one_ms, a, b);
        one_ms /= 4.0f;
    }
    int timed = 0;
    float ms = 0.0f;
    if (e == hipSuccess && one_ms > 0.0f) {
        const int total = std::max(6, std::min(200000, (int)((float)milliseconds / one_ms)));
        const int lead = total / 3;
        timed = total - lead;
        for (int r = 0; r < lead; ++r) hbm_probe_copy_kernel<true><<<grid, 256, 0, s>>>(src, dst, n4);
        e = hipEventRecord(a, s);
        for (int r = 0; r < timed; ++r) hbm_probe_copy_kernel<true><<<grid, 256, 0, s>>>(src, dst, n4);
        if (e == hipSuccess) e = hipEventRecord(b, s);
        if (e == hipSuccess) e = hipEventSynchronize(b);
        if (e == hipSuccess) e = hipEventElapsedTime(&ms, a, b);
    }
    if (a) (void)hipEventDestroy(a);
    if (b) (void)hipEventDestroy(b);
    if (s) (void)hipStreamDestroy(s);
    if (src) (void)hipFree(src);
    if (dst) (void)hipFree(dst);
    if (e != hipSuccess) return fail(LDPC_ERR_HIP, "hbm sustained probe: %s", hipGetErrorString(e));
    if (ms > 0.0f) *copy_gbs = 2.0 * (double)(n4 * sizeof(ldpc::vf4)) * timed / (ms * 1e-3) / 1e9;
    return LDPC_OK;
}

}  /* extern "C" */
