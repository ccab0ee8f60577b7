// Is some device memory slower than other?  Allocates blocks of `gib` GiB one after the other (all held) and measures a
// streaming float4 copy inside each block (first half -> second half, non-temporal, best of 5; read + write counted), then a
// second pass over all blocks in the same order.  usage: memory_map [blocks] [GiB per block]
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
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
    const int n = argc > 1 ? atoi(argv[1]) : 64;
    const double gib = argc > 2 ? atof(argv[2]) : 2.0;
    const size_t bytes = (size_t)(gib * (double)(1ull << 30));
    std::vector<char *> blocks;
    hipEvent_t a, b;
    hipEventCreate(&a); hipEventCreate(&b);
    auto measure = [&](char *p) {
        const size_t n4 = bytes / 2 / sizeof(vf4);
        float best = 1e30f;
        for (int r = 0; r < 6; ++r) {
            hipEventRecord(a, 0);
            copy_kernel<<<256 * 64, 256>>>((const vf4 *)p, (vf4 *)(p + bytes / 2), n4);
            hipEventRecord(b, 0); hipEventSynchronize(b);
            float ms; hipEventElapsedTime(&ms, a, b);
            if (r && ms < best) best = ms;
        }
        return (double)bytes / best / 1e6;
    };
    for (int i = 0; i < n; ++i) {
        char *p = nullptr;
        if (hipMalloc((void **)&p, bytes) != hipSuccess) { printf("block %d: out of memory\n", i); break; }
        hipMemset(p, 0, bytes);
        blocks.push_back(p);
        printf("block %3d at %p  %6.0f GB/s\n", i, (void *)p, measure(p));
        fflush(stdout);
    }
    printf("second pass:");
    for (size_t i = 0; i < blocks.size(); ++i) printf(" %.0f", measure(blocks[i]));
    printf("\n");
    return 0;
}
