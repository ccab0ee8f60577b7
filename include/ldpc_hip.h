/*
 * ldpc_hip.h -- C ABI of libldpc_hip.so: batched LDPC belief-propagation decoding
 * on AMD MI355X (gfx950), hand-written HIP kernels.
 *
 * This is the drop-in boundary for the decode hot path of wing02/MyLdpcCppApi.
 * The reference has no FFI layer: its boundary is the C++ class `Coder`
 * (MyLdpc.h:104-238) whose decode() (MyLdpc.cpp:571-618) drives OpenCL kernels
 * (decodeCL.c).  Each entry point below names the reference interface it
 * replaces.  include/MyLdpc.h + csrc/MyLdpc.cpp re-create `Coder` on top of this
 * ABI; INTEGRATION.md shows the binding.
 *
 * Conventions (all taken from the reference):
 *   - H is given as its nonzeros in ROW-MAJOR order; edge id = rank in that
 *     order (MyLdpc.cpp:186-219).  fp32 reductions follow the reference's
 *     orders: ascending edge id along a row and along a column.
 *   - channel values `llr`: N floats per frame, frame-major, +1 <-> bit 0,
 *     -1 <-> bit 1 (MyLdpc.cpp:1066-1069).
 *   - output: the first K hard bits of every frame, LSB-first
 *     (decodeCL.c:188-199 / MyLdpc.cpp:765-774).
 *   - every function returns 0 on success (LDPC_SUCCESS == 0, MyLdpc.h:24) and
 *     a positive LDPC_ERR_* code otherwise; it never exits the process
 *     (the reference calls exit(0), MyLdpc.cpp:243-254).  ldpc_last_error()
 *     returns the message of the calling thread's last failure.
 *   - a decoder handle is not re-entrant (neither is Coder, MyLdpc.h:184-236):
 *     one handle per host thread / stream.
 *
 * There is no CPU fallback: without a usable HIP device every compute entry
 * point fails with LDPC_ERR_HIP.
 */
#ifndef LDPC_HIP_H_
#define LDPC_HIP_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

#define LDPC_HIP_ABI_VERSION 3   /* 2: tuning fields in the config (were reserved[8]), device lists;
                                    3: host_input / host_copy_threads replace the experimental `streams` */

enum ldpc_status {
    LDPC_OK = 0,
    LDPC_ERR_ARG = 1,         /* bad argument (message says which)            */
    LDPC_ERR_HIP = 2,         /* HIP runtime error / no device                */
    LDPC_ERR_NOMEM = 3,
    LDPC_ERR_UNSUPPORTED = 4, /* valid request this build cannot serve        */
    LDPC_ERR_STATE = 5        /* call order / handle misuse                   */
};

/* decodeType of the reference (MyLdpc.h:37-39) maps as:
 *   DecodeSP                 -> LDPC_ALGO_SP       (decodeCL.c:3-108)
 *   DecodeMS, DecodeCPU      -> LDPC_ALGO_MS       (decodeCL.c:113-186, MyLdpc.cpp:684-784)
 *   DecodeTDMPCL             -> LDPC_ALGO_LAYERED  (the fused kernel, decodeCL.c:307-426)
 *   DecodeTDMP               -> LDPC_ALGO_LAYERED_HOST where the reference's host-layered path is
 *                               well defined (all rows of H of one weight), else LDPC_ALGO_LAYERED
 *   DecodeMSCL               -> LDPC_ALGO_MS_FUSED (decodeCL.c:432-567) */
enum ldpc_algo {
    LDPC_ALGO_SP = 0,      /* flooding sum-product, probability domain, fp32     */
    LDPC_ALGO_MS = 1,      /* flooding min-sum, fp32                             */
    LDPC_ALGO_LAYERED = 2, /* layered (TDMP) min-sum                             */
    LDPC_ALGO_MS_FUSED = 3,/* flooding min-sum with the arithmetic of the fused kernel
                              decodeOnceMS (DecodeMSCL, decodeCL.c:432-567): short
                              quasi-cyclic codes only, whole decode in LDS; the reference
                              hard-codes max_iter = 120 there                            */
    LDPC_ALGO_LAYERED_HOST = 4 /* layered min-sum as the reference's HOST drives it (DecodeTDMP,
                              MyLdpc.cpp:889-976 over decodeCL.c:203-300): check node of the MS
                              kernel chain, three-way hard decision once per iteration.  The
                              reference mis-sizes its layers unless every row of H has the same
                              weight (MyLdpc.cpp:907,958): any other H is LDPC_ERR_UNSUPPORTED.
                              fp32, one launch per layer                                    */
};

enum ldpc_msg_dtype { LDPC_MSG_F32 = 0, LDPC_MSG_F16 = 1 };

/* ldpc_decode() and the caller's input buffer (the reference copies it with a blocking
 * enqueueWriteBuffer, MyLdpc.cpp:796 / :988).
 *   STAGED (the default): the library never hands the caller's pageable memory to the HIP runtime and
 *     never changes its page state.  Worker threads owned by the handle copy each launch group, 8 MiB
 *     at a time, into a ring of the library's own pinned buffers, from which it is DMA-copied to the
 *     device while the previous group decodes.
 *   LOCK_PAGES (opt-in): groups larger than 4 MiB are page-locked where they lie (hipHostRegister) for the
 *     duration of the call and DMA-read in place: no CPU copy (worth it for long streams of cheap
 *     decodes, where the CPU copy is slower than the decode).  Whole pages strictly inside the call's
 *     own byte range only; every range is recorded process-wide, released before the call returns, and a
 *     range that cannot be released makes the call -- and ldpc_decoder_destroy -- fail.  A group whose
 *     pages cannot be locked (another call of this library holds them) is staged instead.
 * In both modes groups of up to 4 MiB are copied by the calling thread into pinned scratch. */
enum ldpc_host_input { LDPC_HOST_INPUT_AUTO = 0, LDPC_HOST_INPUT_STAGED = 1, LDPC_HOST_INPUT_LOCK_PAGES = 2 };

enum ldpc_pack_mode {
    LDPC_PACK_BYTES = 0, /* toChar, decodeCL.c:188-199: K/8 whole bytes per frame at (frame*K)/8 */
    LDPC_PACK_BITS = 1   /* decodeCPU, MyLdpc.cpp:765-774: bit i of frame b at bit b*K+i         */
};

typedef struct ldpc_graph ldpc_graph;
typedef struct ldpc_decoder ldpc_decoder;

typedef struct ldpc_decoder_config {
    uint32_t struct_size;   /* = sizeof(ldpc_decoder_config); ABI guard                    */
    int32_t K;              /* information bits per frame (ldpcK)                          */
    int32_t max_batch;      /* frames per launch group (forDecoder's batchSize)            */
    int32_t algo;           /* enum ldpc_algo                                              */
    int32_t msg_dtype;      /* enum ldpc_msg_dtype                                         */
    int32_t max_iter;       /* `times`, MyLdpc.cpp:24 (reference: 40)                      */
    float llr_scale;        /* SP only: q = exp(llr_scale*y)..., decodeCL.c:9 (reference: 8) */
    int32_t early_term;     /* 1: frames freeze when their syndrome is clean (reference
                               behaviour, decodeCL.c:27,48-49); 0: always run max_iter      */
    int32_t device;         /* HIP device ordinal                                          */
    int32_t layer_rows;     /* rows per layer = circulant size z.  Required for LDPC_ALGO_LAYERED / MS_FUSED; with
                               SP / MS it lets the one-launch kernels for quasi-cyclic codes apply (0: streaming) */
    int32_t pack_mode;      /* enum ldpc_pack_mode                                         */
    int32_t frames_per_lane;/* tuning: 0 = auto, else 1, 2 or 4 (tile = 64*frames_per_lane) */
    int32_t poll_interval;  /* early_term: host checks "all frames done" every this many
                               iterations (0 = never; finished tiles still skip on device).
                               With polling on, the last <= 512 running frames of a multi-tile
                               batch are handed to a small child decoder (tail compaction)   */
    /* ---- tuning (was reserved[8]): 0 = automatic everywhere.  Kernel selection and launch
     *      shapes only -- results never depend on these (the parity tests run the alternatives
     *      against each other).  The library reads NO environment variables. ---------------- */
    int32_t tune_flags;         /* LDPC_TUNE_* two-bit fields below: 0 auto, 1 force on, 2 force off */
    int32_t tune_rows_per_wave; /* check kernels: rows per wave                                   */
    int32_t tune_cols_per_wave; /* variable kernels: columns per wave                             */
    int32_t tune_link_rows;     /* rows per wave of the column-fused check kernel (default 16; 4 for a
                                   single tile); -1 = column-local fusion off                     */
    int32_t tune_compact;       /* tail compaction: hand over when <= this many frames still run
                                   (default and maximum 512); -1 = off                            */
    int32_t tune_ldsp_grid;     /* record kernels (ldsp_kernels.hpp): persistent workgroups        */
    int32_t tune_ldsp_shape;    /* workgroups per CU | waves per workgroup << 8                    */
    int32_t tune_place;         /* streaming flooding decoders: fresh allocations tried for the check->variable and for
                                   the variable->check array when the decoder is created (one message round is timed with
                                   each; the fastest combination is kept, the others are released).  WHICH allocations
                                   lie behind these two arrays decides between speeds of the streaming check kernel that
                                   differ by up to 19 % and last as long as the allocations.
                                   0 = automatic (up to 6 per array when the arrays hold at least 256 MiB and memory
                                   allows; after four measurements a stage stops once it has seen the fast speed next to the slow one), 1 = take the first allocations
                                   as they come, 2..8 = that many per array                              */
    int32_t host_input;         /* enum ldpc_host_input: how ldpc_decode() moves the caller's pageable channel
                                   values to the device (memory the caller has page-locked itself is always
                                   copied from directly)                                               */
    int32_t host_copy_threads;  /* LDPC_HOST_INPUT_STAGED: CPU threads that copy a launch group into the
                                   library's pinned ring (0 = automatic: 4; 1..16).  They belong to the
                                   handle: started on its first large host-buffer call, joined when it is
                                   destroyed                                                           */
    int32_t tune_q_order;       /* streaming flooding decoders: the variable->check array is stored in the order its
                                   writers produce it (variable-node kernels column by column), so that every store
                                   of a round streams and the check kernels gather their inputs instead
                                   (0 = automatic: on; 1 = on; -1 = off: edge order, as the check->variable array) */
} ldpc_decoder_config;

/* two-bit fields of tune_flags: LDPC_TUNE_ON(f) forces the choice on, LDPC_TUNE_OFF(f) off */
enum ldpc_tune_field {
    LDPC_TUNE_FUSED = 0,        /* LDS-resident one-launch kernels (fused_kernels.hpp)              */
    LDPC_TUNE_LDSP = 2,         /* record kernels: posteriors in LDS, check records in cache        */
    LDPC_TUNE_LDSP_EXT = 4,     /* single-layer columns travel with the records (off: all in LDS)   */
    LDPC_TUNE_LDSP_PACK = 6,    /* several frames per wave for circulants of <= 32 rows             */
    LDPC_TUNE_LINK_NARROW = 8,  /* column-fused check kernel in narrow waves (1 value per lane) or wide
                                   (V per lane); default: both -- and LINK_HALF -- are timed when a
                                   decoder with 2 or 4 frames per lane is created and the fastest is kept          */
    LDPC_TUNE_CHECK_WIDE = 10,  /* check kernels move V floats per lane (default off)               */
    LDPC_TUNE_SYN_XCD = 12,     /* XCD-aware syndrome grid (default on)                             */
    LDPC_TUNE_FUSED_PACK = 14,  /* fused layered kernel: several frames per wave (default on)       */
    LDPC_TUNE_FUSED_LOOP = 16,  /* fused kernels: run-time row loops instead of unrolled (default off) */
    LDPC_TUNE_DEVICE_TAIL = 18, /* device-side early exit + tail compaction without host polling
                                   (default: on when early_term && poll_interval == 0)              */
    LDPC_TUNE_MERGE = 20,       /* degree classes of one bucket share a launch (default on)         */
    LDPC_TUNE_LINK_DEEP = 22,   /* column-fused check kernel requests its inputs two rows ahead     */
    LDPC_TUNE_LINK_HALF = 24,   /* column-fused check kernel with 2 values per lane (tiles of 256)  */
    LDPC_TUNE_LINK_GUIDED = 26, /* column-fused check kernel: its launch ends with shorter row chunks
                                   (default on from 4 tiles)                                         */
    LDPC_TUNE_TILES_FIRST = 28  /* flooding launches as grids of (tiles, blocks): the blocks in flight
                                   are spread over all tiles of the batch (default: the column-fused
                                   check launch only; on: all; off: none)                               */
};
#define LDPC_TUNE_ON(field) (1 << (field))
#define LDPC_TUNE_OFF(field) (2 << (field))

/* Counts (iterations, frames, converged frames, frame_rounds) cover the last call -- every launch group of an
 * ldpc_decode() call; the times are those of its last launch group. */
typedef struct ldpc_decode_stats {
    int32_t iterations_launched; /* check/variable rounds enqueued by the last call (maximum over its launch groups) */
    int32_t batch_time;          /* the reference's `Time=` (MyLdpc.cpp:838,1048): max iters  */
    int64_t frames;              /* frames of the last call                                  */
    int64_t frames_converged;    /* frames whose syndrome was clean                          */
    float ms_total;              /* HIP-event time of the whole decode on its stream         */
    float ms_check;              /* summed check-node kernels (needs timing enabled)         */
    float ms_var;                /* summed variable-node kernels                             */
    float ms_other;              /* init / pack / bookkeeping kernels                        */
    int32_t launches_check;      /* kernel launches behind ms_check                          */
    int32_t launches_var;
    int64_t frame_rounds;        /* streaming flooding kernels: sum over the rounds of the frames in tiles that
                                    still did work (tile size x tiles with a running frame when the round began;
                                    finished tiles leave at kernel entry): what a round's traffic is priced at
                                    under early termination.  0 for the one-launch and layered kernels    */
} ldpc_decode_stats;

/* ---- library -------------------------------------------------------------- */
int ldpc_abi_version(void);
const char *ldpc_last_error(void);
/* Number of HIP devices (0 and LDPC_ERR_HIP if the runtime is unusable). */
int ldpc_device_count(int *count);

/* ---- graph: replaces Coder::forDecoder's adjacency build, MyLdpc.cpp:171-222 ---- */
/* rows/cols: the E nonzeros of the M x N parity-check matrix in row-major order
 * (strictly ascending (row, col)).  Host-only; no device is touched. */
int ldpc_graph_create(const int32_t *rows, const int32_t *cols, int64_t E, int32_t M, int32_t N,
                      ldpc_graph **out);
int ldpc_graph_destroy(ldpc_graph *g);
int ldpc_graph_info(const ldpc_graph *g, int32_t *M, int32_t *N, int64_t *E, int32_t *max_row_deg,
                    int32_t *max_col_deg);

/* ---- decoder: replaces Coder::forDecoder's device setup + addDecodeType,
 *      MyLdpc.cpp:226-305, 307-552 -------------------------------------------- */
void ldpc_decoder_config_init(ldpc_decoder_config *cfg); /* reference defaults */
int ldpc_decoder_create(const ldpc_graph *g, const ldpc_decoder_config *cfg, ldpc_decoder **out);
/* The reference uses devices[0] of its context only (MyLdpc.cpp:226-235).  Here one handle may
 * span several HIP devices: ldpc_decode() cuts the caller's frame stream into n_devices contiguous
 * ranges (ldpc_shard_range), and one host thread per entry of devices[] runs that range through
 * its own device decoder -- own stream, own pinned staging, results copied straight to the
 * caller's buffers at the range's offsets; no exchange between devices (frames are independent).
 * Bytes and iteration counts equal the single-device result.  An ordinal may appear more than
 * once (two decoders sharing one GPU).  cfg->device is ignored; cfg->max_batch is per device.
 * Device-pointer entry points (ldpc_decode_device, taps, dumps) need a single-device handle. */
int ldpc_decoder_create_multi(const ldpc_graph *g, const ldpc_decoder_config *cfg, const int32_t *devices,
                              int32_t n_devices, ldpc_decoder **out);
/* Waits for this handle's own work only (its streams), not for the device; joins the handle's worker
 * threads.  Returns LDPC_ERR_STATE (after freeing the handle all the same) if a page-locked block of a
 * caller's buffer could not be released by an earlier LDPC_HOST_INPUT_LOCK_PAGES call. */
int ldpc_decoder_destroy(ldpc_decoder *d);
/* Frames [*lo, *hi) of part `part` of `parts` when `frames` frames are cut into contiguous, balanced
 * ranges whose boundaries are multiples of `unit` frames (earlier parts take the remainder).  The
 * same arithmetic shards frames over ranks in the benchmark (one process per GPU). */
int ldpc_shard_range(int64_t frames, int32_t part, int32_t parts, int32_t unit, int64_t *lo, int64_t *hi);

/* ---- decode: replaces Coder::decode + decodeOnceSP/MS/TDMP*, MyLdpc.cpp:571-618,
 *      786-1059.  Host buffers; chunks `frames` into max_batch groups; blocking.
 *      out must hold ldpc_out_bytes(K, frames, pack_mode) bytes.
 *      iters (nullable): per frame, the iteration at which its syndrome first was
 *      clean, or max_iter. ------------------------------------------------------- */
int ldpc_decode(ldpc_decoder *d, const float *llr_host, int64_t frames, uint8_t *out_host,
                int64_t out_bytes, int32_t *iters);

/* Same, on buffers already resident in this decoder's device memory; enqueued on
 * `stream` (a hipStream_t, NULL = default stream), returns without waiting unless
 * poll_interval > 0.  frames <= max_batch.  iters_dev nullable. */
int ldpc_decode_device(ldpc_decoder *d, const float *llr_dev, int64_t frames, uint8_t *out_dev,
                       int64_t out_bytes, int32_t *iters_dev, void *stream);

int64_t ldpc_out_bytes(int32_t K, int64_t frames, int32_t pack_mode);

/* Per-kernel HIP-event timing of subsequent decode calls (two event records per
 * launch on the decode stream; off by default).  enable = 1: every call; enable = k > 1:
 * every k-th ldpc_decode_device call, starting with the next one (the events cost about 2 %
 * of a 4096-frame step, so a benchmark times a sample of its steps).  Enabling (again)
 * clears what was gathered; times then accumulate over all following timed calls.
 * ldpc_decoder_stats: stats of the last call (ms_check/ms_var/ms_other: everything
 * gathered since timing was enabled); blocks until that call has finished. */
int ldpc_decoder_set_timing(ldpc_decoder *d, int enable);
int ldpc_decoder_stats(ldpc_decoder *d, ldpc_decode_stats *stats);

/* One line per distinct kernel launched since timing was enabled. */
typedef struct ldpc_kernel_time {
    int32_t phase;         /* 0 check node, 1 variable node, 2 layer, 3 other          */
    int32_t degree;        /* row / column degree the kernel is specialised for        */
    int32_t launches;
    float ms_total;        /* sum of HIP-event durations of those launches             */
    int64_t bytes_total;   /* ALGORITHMIC bytes of those launches in the two-kernel formulation
                              (16 E + 4 N per frame-iteration over all kernels): 4 B per message
                              read or written + 4 B per channel value read, per frame    */
    char name[64];         /* e.g. "check_link_kernel<sp,7,4>" (algo, degree, frames/lane), the
                              kernel's name in a rocprofv3 trace up to the spelling of the
                              template arguments; the one-launch record kernels carry their
                              launch shape instead: "layered_ldsp_kernel[768x384,1]" =
                              [persistent grid x workgroup size, frames per workgroup]  */
    int64_t bytes_moved;   /* bytes those launches' own loads and stores move: = bytes_total except
                              for the column-fused check kernel, whose fused columns' messages
                              never travel through HBM (the figure the PMC counters confirm)  */
} ldpc_kernel_time;
int ldpc_decoder_kernel_times(ldpc_decoder *d, ldpc_kernel_time *out, int32_t capacity,
                              int32_t *count);

/* Which form of the column-fused check kernel this decoder runs (0 wide: V values per lane, 1 narrow: one,
 * 2 half: two) and, if it was chosen by the creation-time measurement, what each candidate took per launch
 * on the decoder's own arrays (ms[0..2] in that order, 0 = not a candidate / not measured; *calibrated = 0 when
 * the form was fixed by the tuning fields or the code has no column-fused rows). */
int ldpc_decoder_link_form(ldpc_decoder *d, int32_t *form, int32_t *calibrated, float ms[3]);

/* The placement search of tune_place: how many combinations were timed (*candidates, 0 = no search; the first is the
 * decoder's original allocations, then fresh R arrays, then fresh Q arrays), which was kept (*kept) and the time of one
 * message round with each (ms[0 .. *candidates), at most 15). */
int ldpc_decoder_placement(ldpc_decoder *d, int32_t *candidates, int32_t *kept, float ms[16]);
/* Measurement aid: device addresses of a streaming decoder's arrays, out[0..3] = Q, R, channel term, hard-bit masks. */
int ldpc_decoder_array_addresses(ldpc_decoder *d, uint64_t out[4]);

/* ---- debug taps (tolerance checks against the oracle) ------------------------
 * Stop the NEXT decode call after `iter` check/variable rounds (0 = off) and keep
 * its messages.  ldpc_decoder_dump then copies them out in the reference's layout
 * [frame][E] / [frame][N] (decodeCL.c: q/r at b*nonZeros+e, posteriors at b*N+n).
 * which: 0 = check->variable messages R (SP: r0-r1), 1 = variable->check
 * messages Q (SP: q0-q1), 2 = channel term (SP: exp(scale*y), MS: y; layered:
 * posterior P), 3 = hard bits as floats 0/1 [frame][N]. */
int ldpc_decoder_set_tap(ldpc_decoder *d, int32_t iter);
int ldpc_decoder_dump(ldpc_decoder *d, int32_t which, float *host_out, int64_t count);

/* ---- test channel and error count on the device: replace Coder::test / gaussian
 *      (MyLdpc.cpp:1061-1105: BPSK, bit 0 -> +1.0, bit 1 -> -1.0, plus N(0, sd^2)) and the
 *      comparison loop of Test.cpp:105-110, for data that never leaves HBM.
 * ldpc_awgn_device: llr_dev[f*N + n] = (bits ? 1 - 2*bits[f*N + n] : +1) + sd * z(seed, first_frame + f, n)
 *      for f < frames; bits_dev: one byte per code bit (0/1), NULL = the all-zero codeword.  z is
 *      the counter-based standard normal of csrc/ldpc_channel.h (Philox4x32-10 + Box-Muller in
 *      IEEE double; identical on host and device, any frame range reproducible anywhere), NOT
 *      the reference's rand()-based gaussian().  Enqueued on `stream`, returns without waiting.
 * ldpc_count_errors_device: compares two packed outputs of `frames` x `bytes_per_frame` bytes
 *      (ref_dev NULL = all zero); blocks; errors[0] = differing bits, [1] = differing bytes (the
 *      reference's ErrNum), [2] = frames with at least one difference. */
int ldpc_awgn_device(float *llr_dev, int64_t frames, int32_t N, const uint8_t *bits_dev, float sd,
                     uint64_t seed, int64_t first_frame, int32_t device, void *stream);
int ldpc_count_errors_device(const uint8_t *out_dev, const uint8_t *ref_dev, int64_t frames,
                             int64_t bytes_per_frame, int64_t errors[3], int32_t device, void *stream);

/* ---- measurement aid: the rate a plain float4 copy of `bytes` bytes (read + write counted)
 *      sustains on `device` right now, best of `reps` launches each with the default cache policy
 *      and with non-temporal loads and stores (the streaming kernels' policy), HIP-event timed on
 *      a stream of its own.  *copy_gbs = the better of the two; by_policy (may be NULL) receives
 *      {default, non-temporal}.  The benchmark reports it next to its roofline figures so that a kernel's fraction of
 *      the 8 TB/s specification can also be read against what the box at hand delivers. */
int ldpc_hbm_probe_device(int32_t device, int64_t bytes, int32_t reps, double *copy_gbs, double *by_policy);
/* The non-temporal copy back to back for `milliseconds` (the first third untimed): the rate the box SUSTAINS.
 * The boxes of this pool drop to about 5.2 TB/s under load at times while a burst still shows 6.4. */
int ldpc_hbm_sustained_device(int32_t device, int64_t bytes, int32_t milliseconds, double *copy_gbs);

/* ---- diagnostics of the host-buffer path (no device is touched) --------------------------------
 * ldpc_host_block_plan: the page arithmetic of LDPC_HOST_INPUT_LOCK_PAGES for launch group `group` of a
 *      call over `frames` frames of N floats, `max_batch` frames per group, first byte at address `base`:
 *      out[0..1] = the group's bytes [s0, s1); out[2..3] = the page-locked block [b0, b1) (equal: none);
 *      out[4] = end of the DMA-read body [b0, body_end); out[5] = 1 if the CPU copies the whole group.
 * ldpc_host_locked_ranges: ranges of caller memory this library holds page-locked right now (*live,
 *      0 between calls) and ranges it failed to release (*stale, 0 unless hipHostUnregister failed). */
int ldpc_host_block_plan(uint64_t base, int64_t frames, int32_t N, int32_t max_batch, int64_t group, uint64_t out[6]);
int ldpc_host_locked_ranges(int64_t *live, int64_t *stale);

#ifdef __cplusplus
}
#endif
#endif /* LDPC_HIP_H_ */
