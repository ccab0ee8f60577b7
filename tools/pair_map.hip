// Streaming copy between SEPARATE allocations: n buffers of `gib` GiB, the float4 non-temporal copy i -> j for every ordered
// pair (best of 4, read + write counted), printed as a matrix of GB/s.  usage: pair_map [buffers] [GiB]
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
    const int n = argc > 1 ? atoi(argv[1]) : 10;
    const double gib = argc > 2 ? atof(argv[2]) : 4.0;
    const size_t bytes = (size_t)(gib * (double)(1ull << 30)), n4 = bytes / sizeof(vf4);
    std::vector<char *> b((size_t)n);
    for (int i = 0; i < n; ++i) { if (hipMalloc((void **)&b[i], bytes) != hipSuccess) return 1; hipMemset(b[i], 0, bytes); printf("buffer %d at %p\n", i, (void *)b[i]); }
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    printf("GB/s, row = source, column = destination\n     ");
    for (int j = 0; j < n; ++j) printf("%6d", j);
    printf("\n");
    for (int i = 0; i < n; ++i) {
        printf("%3d: ", i);
        for (int j = 0; j < n; ++j) {
            if (i == j) { printf("     -"); continue; }
            float best = 1e30f;
            for (int r = 0; r < 5; ++r) {
                hipEventRecord(e0, 0);
                copy_kernel<<<256 * 64, 256>>>((const vf4 *)b[i], (vf4 *)b[j], n4);
                hipEventRecord(e1, 0); hipEventSynchronize(e1);
                float ms; hipEventElapsedTime(&ms, e0, e1);
                if (r && ms < best) best = ms;
            }
            printf("%6.0f", 2.0 * bytes / best / 1e6);
        }
        printf("\n");
        fflush(stdout);
    }
    return 0;
}
