/*
 * ref_driver.c -- runs the reference's OpenCL C kernels (compiled as host C from
 * /root/reference/decodeCL.c through cl_on_host.h) in exactly the order and
 * with exactly the NDRanges the reference's host code enqueues them:
 *   ref_decode_ms   <- Coder::decodeOnceMS   MyLdpc.cpp:786-848
 *   ref_decode_sp   <- Coder::decodeOnceSP   MyLdpc.cpp:977-1059
 *   ref_decode_tdmp <- Coder::decodeOnceTDMPCL MyLdpc.cpp:850-868 (fused kernel)
 *   ref_decode_tdmp_host <- Coder::decodeOnceTDMP MyLdpc.cpp:889-976 (host-layered; valid for
 *                      seeds whose rows all have the same weight, see below)
 * One kernel call = one work-item; an NDRange (B, G) is the double loop below.
 * The graph arrays are the reference's linked-list form (MyLdpc.cpp:171-222),
 * built by the caller (oracle/make_golden.py) from the CSR/CSC form.
 *
 * TEST INFRASTRUCTURE ONLY.  Lives in oracle/_ref/libldpc_ref.so, exists only
 * where /root/reference exists (the build container).
 */
#include <pthread.h>
#include <stdbool.h>
#include <stdint.h>
#include <stdlib.h>
#include <string.h>

_Thread_local int clh_global_id[3];
_Thread_local int clh_local_id[3];
_Thread_local int clh_group_id[3];
_Thread_local pthread_barrier_t *clh_group_barrier;
_Thread_local pthread_barrier_t *clh_start_barrier;

/* kernel entry points of decodeCL.c (signatures as declared there) */
void decodeInit(int *hCols, float *codes, float *q0, float *q1, float *priorP0, float *priorP1,
                const int ldpcN, const int nonZeros, bool *isDones);
void refreshR(int *hRows, float *q0, float *q1, float *r0, float *r1, int *hRowFirstPtr,
              int *hRowNextPtr, const int ldpcM, const int nonZeros, bool *isDones);
void refreshQ(int *hCols, float *q0, float *q1, float *r0, float *r1, float *priorP0,
              float *priorP1, int *hColFirstPtr, int *hColNextPtr, const int ldpcN,
              const int nonZeros, bool *flags, bool *isDones);
void hardDecision(bool *srcBool, float *r0, float *r1, float *priorP0, float *priorP1,
                  int *hColFirstPtr, int *hColNextPtr, const int ldpcN, const int nonZeros,
                  bool *flags, bool *isDones);
void checkResult(bool *srcBool, int *hCols, int *hRowFirstPtr, int *hRowNextPtr, const int ldpcM,
                 const int ldpcN, const int nonZeros, bool *flags, bool *isDones);
void decodeInitMS(int *hCols, float *postCodes, float *lQ, const int ldpcN, const int nonZeros,
                  bool *isDones);
void refreshRMS(int *hRows, float *lQ, float *lR, int *hRowFirstPtr, int *hRowNextPtr,
                const int nonZeros, bool *isDones);
void refreshPostPMS(bool *srcBool, float *lR, float *postCodes, float *lPostP, int *hColFirstPtr,
                    int *hColNextPtr, const int ldpcN, const int nonZeros, bool *flags,
                    bool *isDones);
void refreshQMS(int *hCols, float *lQ, float *lR, float *lPostP, const int ldpcN,
                const int nonZeros, bool *flags, bool *isDones);
void toChar(bool *srcBool, char *srcCode, const int ldpcN, const int ldpcK);
void decodeOnceTDMP(float *postCode, char *srcCode, const char z, const char seedRowLength,
                    const char *hSeed, float *lP, bool *srcBool, bool *flag);

void decodeOnceMS(float *postCode, char *srcCode, const char z, const char seedRowLength, char *hSeed,
                  float *lP, float *lR, bool *srcBool, bool *flag);

void decodeInitTDMP(int *hCols, float *postCodes, float *lQ, const int ldpcN, const int nonZeros,
                    bool *isDones, float *lPostP, const int blockHeavy, float *lR);
void refreshRTDMP(int *hRows, float *lQ, float *lR, int *hRowFirstPtr, int *hRowNextPtr,
                  const int nonZeros, bool *isDones, const int blockHeavy, const int offset);
void refreshPostPTDMP(float *lR, float *lQ, float *lPostP, int *hCols, const int ldpcN,
                      const int nonZeros, bool *isDones, const int blockHeavy, const int offset);
void hardDecisionTDMP(float *lPostP, bool *srcBool, bool *flags, const int ldpcN, bool *isDones);
void refreshQTDMP(int *hCols, float *lQ, float *lR, float *lPostP, const int ldpcN, const int nonZeros,
                  bool *flags, bool *isDones, const int blockHeavy, const int offset);
void checkDones(bool *flags, bool *isDones);

#define NDRANGE2(B, G, CALL)                                   \
    for (int b_ = 0; b_ < (B); ++b_)                           \
        for (int g_ = 0; g_ < (G); ++g_) {                     \
            clh_global_id[0] = b_; clh_global_id[1] = g_;      \
            CALL;                                              \
        }

typedef struct {
    int M, N, K, E;
    int *hRows, *hCols, *hRowFirstPtr, *hRowNextPtr, *hColFirstPtr, *hColNextPtr;
} ref_graph;

typedef struct {
    int iter;                /* 1-based iteration to copy out, 0 = none */
    float *a, *b, *c, *d;    /* MS: a=lR b=lPostP c=lQ ; SP: a=r0 b=r1 c=q0 d=q1 */
} ref_taps;

static void copy_if(float *dst, const float *src, size_t n)
{
    if (dst) memcpy(dst, src, n * sizeof(float));
}

/* Coder::decodeOnceMS, MyLdpc.cpp:786-848.  Returns the batch's `time`. */
int ref_decode_ms(const ref_graph *g, float *postCode, int B, int times, char *srcCode,
                  uint8_t *hard_out, uint8_t *flags_out, const ref_taps *taps)
{
    const int N = g->N, M = g->M, E = g->E, K = g->K;
    float *lQ = calloc((size_t)B * E, sizeof(float));
    float *lR = calloc((size_t)B * E, sizeof(float));
    float *lPostP = calloc((size_t)B * N, sizeof(float));
    bool *srcBool = calloc((size_t)B * N, 1);     /* zero-initialised (the reference leaves it undefined) */
    bool *flags = calloc((size_t)B, 1), *isDones = calloc((size_t)B, 1);
    int time = 0;
    clh_group_barrier = NULL;

    NDRANGE2(B, E, decodeInitMS(g->hCols, postCode, lQ, N, E, isDones));            /* :798-800 */
    while (1) {
        NDRANGE2(B, E, refreshRMS(g->hRows, lQ, lR, g->hRowFirstPtr, g->hRowNextPtr, E, isDones)); /* :804 */
        NDRANGE2(B, N + 1, refreshPostPMS(srcBool, lR, postCode, lPostP, g->hColFirstPtr,
                                          g->hColNextPtr, N, E, flags, isDones));   /* :808 */
        NDRANGE2(B, M, checkResult(srcBool, g->hCols, g->hRowFirstPtr, g->hRowNextPtr, M, N, E,
                                   flags, isDones));                                /* :813 */
        ++time;
        if (taps && taps->iter == time) {
            copy_if(taps->a, lR, (size_t)B * E);
            copy_if(taps->b, lPostP, (size_t)B * N);
        }
        int sumFlag = 0;
        for (int i = 0; i < B; ++i) if (flags[i]) ++sumFlag;                         /* :824-828 */
        if (sumFlag == 0) break;
        if (time == times) break;
        NDRANGE2(B, E, refreshQMS(g->hCols, lQ, lR, lPostP, N, E, flags, isDones));  /* :834 */
        if (taps && taps->iter == time) copy_if(taps->c, lQ, (size_t)B * E);
    }
    NDRANGE2(B, K / 8, toChar(srcBool, srcCode, N, K));                              /* :839 */
    if (hard_out) memcpy(hard_out, srcBool, (size_t)B * N);
    if (flags_out) memcpy(flags_out, flags, (size_t)B);
    free(lQ); free(lR); free(lPostP); free(srcBool); free(flags); free(isDones);
    return time;
}

/* Coder::decodeOnceSP, MyLdpc.cpp:977-1059. */
int ref_decode_sp(const ref_graph *g, float *postCode, int B, int times, char *srcCode,
                  uint8_t *hard_out, uint8_t *flags_out, const ref_taps *taps)
{
    const int N = g->N, M = g->M, E = g->E, K = g->K;
    float *q0 = calloc((size_t)B * E, sizeof(float)), *q1 = calloc((size_t)B * E, sizeof(float));
    float *r0 = calloc((size_t)B * E, sizeof(float)), *r1 = calloc((size_t)B * E, sizeof(float));
    float *p0 = calloc((size_t)B * N, sizeof(float)), *p1 = calloc((size_t)B * N, sizeof(float));
    bool *srcBool = calloc((size_t)B * N, 1);
    bool *flags = calloc((size_t)B, 1), *isDones = calloc((size_t)B, 1);
    int time = 0;
    clh_group_barrier = NULL;

    NDRANGE2(B, E, decodeInit(g->hCols, postCode, q0, q1, p0, p1, N, E, isDones));   /* :993 */
    while (1) {
        NDRANGE2(B, E, refreshR(g->hRows, q0, q1, r0, r1, g->hRowFirstPtr, g->hRowNextPtr, M, E,
                                isDones));                                           /* :1001 */
        NDRANGE2(B, E, hardDecision(srcBool, r0, r1, p0, p1, g->hColFirstPtr, g->hColNextPtr, N, E,
                                    flags, isDones));                                /* :1008 */
        NDRANGE2(B, E, checkResult(srcBool, g->hCols, g->hRowFirstPtr, g->hRowNextPtr, M, N, E,
                                   flags, isDones));                                 /* :1016 */
        int sumFlag = 0;
        for (int i = 0; i < B; ++i) sumFlag += flags[i];                             /* :1031-1034 */
        ++time;
        if (taps && taps->iter == time) {
            copy_if(taps->a, r0, (size_t)B * E);
            copy_if(taps->b, r1, (size_t)B * E);
        }
        if (sumFlag == 0) break;
        if (time == times) break;
        NDRANGE2(B, E, refreshQ(g->hCols, q0, q1, r0, r1, p0, p1, g->hColFirstPtr, g->hColNextPtr,
                                N, E, flags, isDones));                              /* :1040 */
        if (taps && taps->iter == time) {
            copy_if(taps->c, q0, (size_t)B * E);
            copy_if(taps->d, q1, (size_t)B * E);
        }
    }
    NDRANGE2(B, K / 8, toChar(srcBool, srcCode, N, K));                              /* :1049 */
    if (hard_out) memcpy(hard_out, srcBool, (size_t)B * N);
    if (flags_out) memcpy(flags_out, flags, (size_t)B);
    free(q0); free(q1); free(r0); free(r1); free(p0); free(p1);
    free(srcBool); free(flags); free(isDones);
    return time;
}

/* Fused layered kernel decodeOnceTDMP (decodeCL.c:307-426), launched as
 * Coder::decodeOnceTDMPCL does (MyLdpc.cpp:857-859): global B*z, local z, one
 * work-group per frame.  Each work-item is a pthread; barrier() is a
 * pthread_barrier over the z work-items.  Local buffers sized as
 * MyLdpc.cpp:520-522. */
typedef struct {
    int group, lid, z, seedRows;
    float *postCode; char *srcCode; const char *hSeed;
    float *lP; bool *srcBool; bool *flag;
    pthread_barrier_t *bar, *start;
} wi_arg;

static void *wi_main(void *p)
{
    wi_arg *a = (wi_arg *)p;
    clh_group_id[0] = a->group;
    clh_local_id[0] = a->lid;
    clh_global_id[0] = a->group * a->z + a->lid;
    clh_group_barrier = a->bar;
    clh_start_barrier = a->start;
    decodeOnceTDMP(a->postCode, a->srcCode, (char)a->z, (char)a->seedRows, a->hSeed, a->lP,
                   a->srcBool, a->flag);
    return NULL;
}

int ref_decode_tdmp(int z, int seedRows, const char *hSeed, float *postCode, int B, char *srcCode)
{
    const int N = 24 * z;
    if (z > 127 || N > 32767) return -1;          /* char / short limits of the kernel */
    pthread_t *th = malloc(sizeof(pthread_t) * (size_t)z);
    wi_arg *args = malloc(sizeof(wi_arg) * (size_t)z);
    float *lP = malloc(sizeof(float) * (size_t)N);
    bool *srcBool = malloc((size_t)N), flag[1];
    for (int b = 0; b < B; ++b) {
        pthread_barrier_t bar, start;
        pthread_barrier_init(&bar, NULL, (unsigned)z);
        pthread_barrier_init(&start, NULL, (unsigned)z);
        memset(srcBool, 0, (size_t)N);
        flag[0] = 0;
        /* The kernel copies postCode into lP (decodeCL.c:331-334) and enters layer
         * 0 WITHOUT a barrier, although layer 0 reads and updates entries other
         * work-items initialise.  On a GPU with z <= one wavefront that is
         * lock-step; free-running host threads would race.  lP is pre-loaded with
         * the same values and the work-items are held at their first sign() call
         * until every copy is done (cl_on_host.h): the lock-step result. */
        memcpy(lP, postCode + (size_t)b * N, sizeof(float) * (size_t)N);
        for (int l = 0; l < z; ++l) {
            wi_arg a = { b, l, z, seedRows, postCode, srcCode, hSeed, lP, srcBool, flag, &bar, &start };
            args[l] = a;
            pthread_create(&th[l], NULL, wi_main, &args[l]);
        }
        for (int l = 0; l < z; ++l) pthread_join(th[l], NULL);
        pthread_barrier_destroy(&bar);
        pthread_barrier_destroy(&start);
    }
    free(th); free(args); free(lP); free(srcBool);
    return 0;
}

/* Fused flooding kernel decodeOnceMS (decodeCL.c:432-567), launched as Coder::decodeOnceMSCL does
 * (MyLdpc.cpp:877-879): global (B*z, M/z), local (z, M/z): one work-group of M work-items per
 * frame.  Local buffers as MyLdpc.cpp:543-547: lP[N], lR[M*24], srcBool[N], flag. */
typedef struct {
    int group, lid, z, seedRows;
    float *postCode; char *srcCode; char *hSeed;
    float *lP, *lR; bool *srcBool; bool *flag;
    pthread_barrier_t *bar;
} ms_arg;

static void *ms_main(void *p)
{
    ms_arg *a = (ms_arg *)p;
    clh_group_id[0] = a->group;
    clh_local_id[0] = a->lid % a->z;
    clh_local_id[1] = a->lid / a->z;
    clh_global_id[0] = a->group * a->z + clh_local_id[0];
    clh_global_id[1] = clh_local_id[1];
    clh_group_barrier = a->bar;
    clh_start_barrier = NULL;
    decodeOnceMS(a->postCode, a->srcCode, (char)a->z, (char)a->seedRows, a->hSeed, a->lP, a->lR,
                 a->srcBool, a->flag);
    return NULL;
}

int ref_decode_mscl(int z, int seedRows, char *hSeed, float *postCode, int B, char *srcCode)
{
    const int N = 24 * z, M = seedRows * z;
    if (z > 127 || N > 32767) return -1;
    pthread_t *th = malloc(sizeof(pthread_t) * (size_t)M);
    ms_arg *args = malloc(sizeof(ms_arg) * (size_t)M);
    float *lP = malloc(sizeof(float) * (size_t)N);
    float *lR = malloc(sizeof(float) * (size_t)M * 24);
    bool *srcBool = malloc((size_t)N), flag[1];
    pthread_attr_t attr;
    pthread_attr_init(&attr);
    pthread_attr_setstacksize(&attr, 256 * 1024);
    for (int b = 0; b < B; ++b) {
        pthread_barrier_t bar;
        pthread_barrier_init(&bar, NULL, (unsigned)M);
        memset(srcBool, 0, (size_t)N);
        memset(lR, 0, sizeof(float) * (size_t)M * 24);
        flag[0] = 0;
        for (int l = 0; l < M; ++l) {
            ms_arg a = { b, l, z, seedRows, postCode, srcCode, hSeed, lP, lR, srcBool, flag, &bar };
            args[l] = a;
            if (pthread_create(&th[l], &attr, ms_main, &args[l])) return -2;
        }
        for (int l = 0; l < M; ++l) pthread_join(th[l], NULL);
        pthread_barrier_destroy(&bar);
    }
    pthread_attr_destroy(&attr);
    free(th); free(args); free(lP); free(lR); free(srcBool);
    return 0;
}


/* Coder::decodeOnceTDMP, MyLdpc.cpp:889-976 (kernels bound at :440-500: decodeInitTDMP, refreshRTDMP,
 * refreshQTDMP, refreshPostPTDMP, hardDecisionTDMP, the shared checkResult / checkDones / toChar),
 * statement for statement -- INCLUDING the layer size `hRowRange[blockRow + z] - hRowRange[blockRow]`
 * (:907, :958), which indexes hRowRange with the layer number instead of the layer's first row.  It
 * equals the edges of layer `blockRow` exactly when every window of z rows holds the same number of
 * edges, i.e. for seeds whose rows all have the same weight (2/3A, 5/6); the caller only uses those.
 * hRowRange: [M + 1] first edge of each row.  Returns the batch's `time`. */
int ref_decode_tdmp_host(const ref_graph *g, const int *hRowRange, int z, float *postCode, int B, int times,
                         char *srcCode, uint8_t *hard_out, uint8_t *flags_out, const ref_taps *taps)
{
    const int N = g->N, M = g->M, E = g->E, K = g->K;
    const int blockHeavy = N;                                                         /* :441 */
    float *lR = calloc((size_t)B * E, sizeof(float));
    float *lQ = calloc((size_t)B * blockHeavy, sizeof(float));
    float *lPostP = calloc((size_t)B * N, sizeof(float));
    bool *srcBool = calloc((size_t)B * N, 1);     /* zero-initialised (the reference leaves it undefined) */
    bool *flags = calloc((size_t)B, 1), *isDones = calloc((size_t)B, 1);
    int time = 0;
    clh_group_barrier = NULL;

    NDRANGE2(B, N + 1, decodeInitTDMP(g->hCols, postCode, lQ, N, E, isDones, lPostP, blockHeavy, lR));  /* :899-901 */
    int offset = 0, blockRow = 0;
    int threadNum = hRowRange[blockRow + z] - hRowRange[blockRow];                     /* :905-907 */
    while (1) {
        NDRANGE2(B, threadNum, refreshRTDMP(g->hRows, lQ, lR, g->hRowFirstPtr, g->hRowNextPtr, E, isDones,
                                            blockHeavy, offset));                      /* :910-913 */
        NDRANGE2(B, threadNum, refreshPostPTDMP(lR, lQ, lPostP, g->hCols, N, E, isDones, blockHeavy, offset)); /* :915-919 */
        ++blockRow;
        offset += threadNum;
        if (blockRow == M / z) {                                                       /* :923 */
            NDRANGE2(B, N + 1, hardDecisionTDMP(lPostP, srcBool, flags, N, isDones));  /* :924-927 */
            NDRANGE2(B, M, checkResult(srcBool, g->hCols, g->hRowFirstPtr, g->hRowNextPtr, M, N, E, flags, isDones)); /* :929 */
            NDRANGE2(B, 1, checkDones(flags, isDones));                                /* :934 */
            ++time;
            if (taps && taps->iter == time) {
                copy_if(taps->a, lR, (size_t)B * E);
                copy_if(taps->b, lPostP, (size_t)B * N);
            }
            int sumFlag = 0;
            for (int i = 0; i < B; ++i) if (flags[i]) ++sumFlag;                       /* :945-948 */
            if (sumFlag == 0) break;
            if (time == times) break;
            blockRow = 0;
            offset = 0;
        }
        threadNum = hRowRange[blockRow + z] - hRowRange[blockRow];                     /* :958 */
        NDRANGE2(B, threadNum, refreshQTDMP(g->hCols, lQ, lR, lPostP, N, E, flags, isDones, blockHeavy, offset)); /* :959-963 */
    }
    NDRANGE2(B, K / 8, toChar(srcBool, srcCode, N, K));                                /* :967 */
    if (hard_out) memcpy(hard_out, srcBool, (size_t)B * N);
    if (flags_out) memcpy(flags_out, flags, (size_t)B);
    free(lR); free(lQ); free(lPostP); free(srcBool); free(flags); free(isDones);
    return time;
}
