// One big allocation: the float4 non-temporal copy of 4 GiB from offset 0 (and from offset 2 GiB) to offset d, for d = 4 ... 60 GiB.
// Is there a distance between a read stream and a write stream that is always fast?  usage: offset_map [GiB total] [GiB per copy] [GiB step]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float vf4 __attribute__((ext_vector_type(4)));
__global__ __launch_bounds__(256) void copy_kernel(const vf4 *__restrict__ src, vf4 *__restrict__ dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 256 * 4;
    for (size_t i = (size_t)blockIdx.x * 256 * 4 + threadIdx.x; i < n4; i += stride) {
        vf4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + (size_t)k * 256 < n4) v[k] = __builtin_nontemporal_load(&src[i + (size_t)k * 256]);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + (size_t)k * 256 < n4) __builtin_nontemporal_store(v[k], &dst[i + (size_t)k * 256]);
    }
}
int main(int argc, char **argv)
{
    const size_t G = 1ull << 30, total = (size_t)(argc > 1 ? atoi(argv[1]) : 64) * G;
    const size_t bytes = (size_t)((argc > 2 ? atof(argv[2]) : 4.0) * G), step = (size_t)((argc > 3 ? atof(argv[3]) : 2.0) * G), n4 = bytes / sizeof(vf4);
    for (int trial = 0; trial < 3; ++trial) {
        char *p = nullptr;
        if (hipMalloc((void **)&p, total) != hipSuccess) return 1;
        hipMemset(p, 0, total);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        printf("allocation %d at %p:", trial, (void *)p);
        for (size_t d = bytes; d + bytes <= total; d += step) {
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(e0, 0);
                copy_kernel<<<256 * 64, 256>>>((const vf4 *)p, (vf4 *)(p + d), n4);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r && ms < best) best = ms;
            }
            printf(" %.2f:%.0f", (double)d / G, 2.0 * bytes / best / 1e6);
        }
        printf("\n");
        fflush(stdout);
        hipFree(p);
    }
    return 0;
}
