/*
 * check_expf.c -- compare ldpc_expf() with the host libm expf() over every
 * float bit pattern (2^32 inputs, split over threads).
 *
 *   gcc -O2 -ffp-contract=off -pthread tools/check_expf.c -lm -o /tmp/check_expf
 *   /tmp/check_expf [nthreads]
 *
 * Prints the number of inputs whose results differ in any bit (NaN payloads
 * are compared as "both NaN").  Expected: 0 on glibc >= 2.27 / x86-64 with FMA.
 */
#include <math.h>
#include <pthread.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>
#include "../myldpccppapi_amd/csrc/ldpc_expf.h"

typedef struct { uint64_t lo, hi; uint64_t bad; uint32_t first_bad; } job_t;

static void *run(void *p)
{
    job_t *j = (job_t *)p;
    j->bad = 0;
    for (uint64_t u = j->lo; u < j->hi; ++u) {
        uint32_t b = (uint32_t)u;
        float x; memcpy(&x, &b, 4);
        float a = ldpc_expf(x), r = expf(x);
        uint32_t ab, rb; memcpy(&ab, &a, 4); memcpy(&rb, &r, 4);
        if (ab != rb && !(a != a && r != r)) {
            if (!j->bad) j->first_bad = b;
            ++j->bad;
        }
    }
    return NULL;
}

int main(int argc, char **argv)
{
    int nt = argc > 1 ? atoi(argv[1]) : 8;
    pthread_t th[64]; job_t jobs[64];
    if (nt < 1 || nt > 64) nt = 8;
    uint64_t total = 1ULL << 32, per = total / nt;
    for (int i = 0; i < nt; ++i) {
        jobs[i].lo = i * per; jobs[i].hi = (i == nt - 1) ? total : (i + 1) * per;
        pthread_create(&th[i], NULL, run, &jobs[i]);
    }
    uint64_t bad = 0; uint32_t fb = 0;
    for (int i = 0; i < nt; ++i) {
        pthread_join(th[i], NULL);
        if (jobs[i].bad && !bad) fb = jobs[i].first_bad;
        bad += jobs[i].bad;
    }
    printf("inputs=%llu mismatches=%llu first_bad=0x%08x\n",
           (unsigned long long)total, (unsigned long long)bad, fb);
    return bad ? 1 : 0;
}
