/*
 * cl_on_host.h -- lets gcc read the reference's OpenCL C kernel file
 * (/root/reference/decodeCL.c) as ordinary host C, so that the reference's own
 * arithmetic can be run in this container (no OpenCL device here).
 *
 * It is force-included (-include) in front of the UNMODIFIED reference file,
 * which is compiled where it lies; nothing of the reference is copied.  It only
 * maps the OpenCL C execution model onto a host thread:
 *   - address-space / kernel qualifiers vanish,
 *   - work-item ids come from thread-local variables the driver sets before
 *     each call (one call = one work-item),
 *   - barrier() is a pthread barrier shared by the work-items of a work-group
 *     (only the two fused kernels use it),
 *   - exp/fmin/fabs are the float libm forms (OpenCL overloads on float),
 *     sign() as OpenCL defines it.
 * TEST INFRASTRUCTURE ONLY; output goes to oracle/_ref/ (git-ignored).
 */
#ifndef CL_ON_HOST_H_
#define CL_ON_HOST_H_

#include <math.h>
#include <pthread.h>
#include <stdbool.h>

#define kernel
#define global
#define local
#define constant const
#define CLK_LOCAL_MEM_FENCE 1

extern _Thread_local int clh_global_id[3];
extern _Thread_local int clh_local_id[3];
extern _Thread_local int clh_group_id[3];
extern _Thread_local pthread_barrier_t *clh_group_barrier;
/* Lock-step start for the fused layered kernel.  decodeOnceTDMP fills the shared
 * lP array (decodeCL.c:331-334) and enters layer 0 with no barrier in between,
 * which is only correct when the work-items of a group run in lock-step (one
 * wavefront).  Free-running host threads would let a late work-item's fill
 * overwrite values an early one has already updated.  The driver pre-loads lP
 * (so early reads are right) and parks every work-item at its first sign() call
 * -- which in layer 0 comes before the work-item's first store to lP
 * (decodeCL.c:353-357) -- until all fills are done. */
extern _Thread_local pthread_barrier_t *clh_start_barrier;

static inline int get_global_id(int dim) { return clh_global_id[dim]; }
static inline int get_local_id(int dim) { return clh_local_id[dim]; }
static inline int get_group_id(int dim) { return clh_group_id[dim]; }
static inline void barrier(int flags)
{
    (void)flags;
    if (clh_group_barrier) pthread_barrier_wait(clh_group_barrier);
}
static inline float sign(float x)
{
    if (clh_start_barrier) {
        pthread_barrier_wait(clh_start_barrier);
        clh_start_barrier = NULL;
    }
    return x > 0.0f ? 1.0f : (x < 0.0f ? -1.0f : (x == 0.0f ? x : 0.0f));
}
#define exp(x) expf(x)
#define fmin(a, b) fminf(a, b)
#define fabs(x) fabsf(x)

#endif
