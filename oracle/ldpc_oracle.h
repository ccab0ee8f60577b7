/*
 * ldpc_oracle.h -- CPU restatement of the reference's LDPC decode hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / the timed CPU baseline.  The product path
 * (myldpccppapi_amd/, include/) never links or imports it.
 *
 * Parity status: PINNED against outputs of the reference's own kernel source
 * (/root/reference/decodeCL.c compiled as host C, oracle/ref_host) through the
 * golden vectors in tests/golden/ (made by oracle/make_golden.py).  The
 * reference itself ships no golden vectors or known-answer tests
 * (Test.cpp:29 seeds from time(0)).  MyLdpc.cpp is NOT built: it needs Eigen
 * (absent from this image) and an OpenCL GPU device; no stand-ins were written
 * for either.
 *
 * Every function cites the reference lines it restates.  Arithmetic is plain
 * fp32 in the reference's operation order; compile with -ffp-contract=off and
 * without -ffast-math.
 */
#ifndef LDPC_ORACLE_H_
#define LDPC_ORACLE_H_

#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* Seed (base) matrices of the six code rates, MyLdpc.h:40-101, as data.
 * rate: 0=1/2, 1=2/3A, 2=2/3B, 3=3/4A, 4=3/4B, 5=5/6 (enum rate_type order,
 * MyLdpc.h:33-35).  Returns the number of seed rows (12/8/8/6/6/4), 24 columns. */
int oracle_seed_rows(int rate);
const signed char *oracle_seed(int rate);

/* H construction, MyLdpc.cpp:52-109: expands the seed into z x z circulants
 * and lists the nonzeros in row-major order (the order Eigen's RowMajor
 * iteration yields at MyLdpc.cpp:186-191).  rows/cols must hold
 * oracle_wimax_nnz(rate) * z entries.  Returns E. */
int64_t oracle_wimax_nnz(int rate);
int64_t oracle_wimax_edges(int rate, int32_t N, int32_t *rows, int32_t *cols);

/* Adjacency build, MyLdpc.cpp:171-222, restated as CSR/CSC arrays that keep the
 * reference's traversal orders (row lists and column lists both ascending in
 * edge id).  row_ptr[M+1] (= hRowRange), col_ptr[N+1], col_edge[E].
 * Returns 0, or -1 if the edge list is not in row-major order. */
int oracle_build_adjacency(int32_t M, int32_t N, int64_t E,
                           const int32_t *rows, const int32_t *cols,
                           int32_t *row_ptr, int32_t *col_ptr, int32_t *col_edge);

typedef struct {
    int32_t M, N, K;
    int64_t E;
    const int32_t *rows;      /* [E] hRows */
    const int32_t *cols;      /* [E] hCols */
    const int32_t *row_ptr;   /* [M+1] */
    const int32_t *col_ptr;   /* [N+1] */
    const int32_t *col_edge;  /* [E] */
} oracle_graph;

/* Optional taps: state of ONE iteration (1-based `iter`) copied out for every
 * frame, laid out [frame][E] / [frame][N].  NULL pointers are skipped.
 * MS / layered: r = lR after the check-node update, post = lPostP, q = lQ after
 * the following variable-node update (unchanged if the loop exits first).
 * SP: r0/r1 after refreshR, q0/q1 after the following refreshQ. */
typedef struct {
    int iter;
    float *r, *q, *post;
    float *r0, *r1, *q0, *q1;
} oracle_taps;

/* pack_mode 0: toChar (decodeCL.c:188-199): K/8 whole bytes per frame at byte
 *              offset (frame*K)/8.
 * pack_mode 1: decodeCPU (MyLdpc.cpp:765-774): bit i of frame b at bit offset
 *              b*K+i (out must be zeroed by the caller; bytes are OR-ed). */

/* Min-sum flooding: decodeCPU MyLdpc.cpp:684-784 == decodeOnceMS
 * MyLdpc.cpp:786-848 + kernels decodeCL.c:113-186,88-108.  iters[frame] =
 * iteration at which the syndrome first became clean, else max_iter.
 * hard_out (nullable): [frames][N] hard bits, one byte each.
 * msg_f16 != 0: this repository's fp16-message extension (not in the reference):
 * channel values and variable->check messages are rounded to IEEE binary16 when
 * stored, arithmetic stays fp32.  Parity for that mode is vs this definition only. */
int oracle_decode_ms(const oracle_graph *g, const float *y, int64_t frames,
                     int max_iter, int pack_mode, uint8_t *out, int64_t out_bytes,
                     int32_t *iters, uint8_t *hard_out, const oracle_taps *taps, int msg_f16);

/* The binary16 round trip used by msg_f16 (exposed for its own test). */
void oracle_f16_round(const float *in, float *out, int64_t n);

/* Sum-product flooding, probability domain: decodeOnceSP MyLdpc.cpp:977-1059 +
 * kernels decodeCL.c:3-108.  llr_scale is the reference's hard-coded 8
 * (decodeCL.c:9).  exp is the host libm expf. */
int oracle_decode_sp(const oracle_graph *g, const float *y, int64_t frames,
                     int max_iter, float llr_scale, int pack_mode, uint8_t *out,
                     int64_t out_bytes, int32_t *iters, uint8_t *hard_out,
                     const oracle_taps *taps);

/* Layered (TDMP) min-sum with the semantics of the fused kernel
 * decodeOnceTDMP, decodeCL.c:307-426, on a general edge list: layer l = rows
 * [l*layer_rows, (l+1)*layer_rows).  Rows of one layer must not share a column.
 * undefined_frames (nullable) [frames]: set to 1 for frames on which the
 * reference kernel reads its uninitialised `bInd` (a row with every |Q| > 1000);
 * the restatement then gives every edge of that row magnitude 1000. */
int oracle_decode_layered(const oracle_graph *g, int32_t layer_rows, const float *y,
                          int64_t frames, int max_iter, int pack_mode, uint8_t *out,
                          int64_t out_bytes, int32_t *iters, uint8_t *hard_out,
                          const oracle_taps *taps, uint8_t *undefined_frames);

/* Layered min-sum as the reference's HOST-layered path runs it: Coder::decodeOnceTDMP,
 * MyLdpc.cpp:889-976, over decodeInitTDMP / refreshQTDMP / refreshRTDMP / refreshPostPTDMP /
 * hardDecisionTDMP / checkResult / checkDones (decodeCL.c:88-108, 203-300).  Same schedule as
 * above, but the check node of the MS kernel chain (sign = XOR of `< 0`, magnitude = fmin chain
 * from 1000) and the three-way hard decision after the last layer (P > 0 -> 0, P < 0 -> 1, else
 * the bit stays; bits start at 0 here -- the reference leaves that buffer uninitialised).  The
 * reference sizes its layers as hRowRange[l + z] - hRowRange[l] (:907, :958), which is the edge
 * count of layer l only when all rows have the same weight: returns -3 for any other H. */
int oracle_decode_layered_host(const oracle_graph *g, int32_t layer_rows, const float *y,
                               int64_t frames, int max_iter, int pack_mode, uint8_t *out,
                               int64_t out_bytes, int32_t *iters, uint8_t *hard_out,
                               const oracle_taps *taps);

/* Fused flooding min-sum with the arithmetic of decodeOnceMS, decodeCL.c:432-567
 * (DecodeMSCL; the reference hard-codes 120 iterations). */
int oracle_decode_ms_fused(const oracle_graph *g, const float *y, int64_t frames, int max_iter,
                           int pack_mode, uint8_t *out, int64_t out_bytes, int32_t *iters,
                           uint8_t *hard_out, const oracle_taps *taps, uint8_t *undefined_frames);

/* Length helpers, MyLdpc.cpp:620-631. */
int64_t oracle_code_size(int64_t src_length, int32_t K);

/* Test channel, Coder::test MyLdpc.cpp:1061-1078 + gaussian :1093-1105, with
 * libc rand() (caller seeds with srand). */
void oracle_test_channel(const uint8_t *prior, float *post, int64_t prior_len, float sd);

#ifdef __cplusplus
}
#endif
#endif
