// gather_probe.hip -- what the memory system gives the variable-node access pattern, by segment size.
//
// The variable-node kernels read D whole message segments per column from pseudo-random places of a
// multi-GB array, add them, and write D segments back to the same places (plus one channel segment
// read).  This probe reproduces exactly that traffic with trivial arithmetic, for segments of 1 KiB
// (the product's V = 4 layout: 64 lanes x 16 B), 2 KiB (two 16-B accesses per lane) and 512 B, and
// for D = 8 and D = 3, next to a plain float4 copy of the same volume.  It answers VERDICT r1 #6(ii):
// would 2-KiB segments lift the variable phase?
//
//   hipcc -O3 --offload-arch=gfx950 tools/gather_probe.hip -o /tmp/gather_probe && /tmp/gather_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <numeric>
#include <random>
#include <vector>

typedef float vf4 __attribute__((ext_vector_type(4)));

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

// One wave per column: reads D segments of SEG4 float4 per lane-slot (SEG4 = 1: 1 KiB per segment,
// 2: 2 KiB, with lanes covering 64 x 16 B twice), sums, writes D segments back.
template <int D, int SEG4, bool NT>
__global__ __launch_bounds__(256) void gather_kernel(const vf4 *__restrict__ src, vf4 *__restrict__ dst,
                                                     const vf4 *__restrict__ chan, const int32_t *__restrict__ idx,
                                                     int n_cols, const int32_t *__restrict__ widx = nullptr)
{
    const int lane = threadIdx.x & 63;
    const int col = blockIdx.x * 4 + (threadIdx.x >> 6);
    if (col >= n_cols) return;
    int e[D];
#pragma unroll
    for (int k = 0; k < D; ++k) e[k] = __builtin_amdgcn_readfirstlane(idx[(size_t)col * D + k]);
    int w[D];
#pragma unroll
    for (int k = 0; k < D; ++k) w[k] = widx ? __builtin_amdgcn_readfirstlane(widx[(size_t)col * D + k]) : e[k];
    vf4 r[D][SEG4], c[SEG4];
#pragma unroll
    for (int s = 0; s < SEG4; ++s) c[s] = NT ? __builtin_nontemporal_load(&chan[((size_t)col * SEG4 + s) * 64 + lane])
                                             : chan[((size_t)col * SEG4 + s) * 64 + lane];
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
        for (int s = 0; s < SEG4; ++s)
            r[k][s] = NT ? __builtin_nontemporal_load(&src[((size_t)e[k] * SEG4 + s) * 64 + lane])
                         : src[((size_t)e[k] * SEG4 + s) * 64 + lane];
    vf4 sum[SEG4];
#pragma unroll
    for (int s = 0; s < SEG4; ++s) {
        sum[s] = c[s];
#pragma unroll
        for (int k = 0; k < D; ++k) sum[s] += r[k][s];
    }
#pragma unroll
    for (int k = 0; k < D; ++k)
#pragma unroll
        for (int s = 0; s < SEG4; ++s) {
            const vf4 q = sum[s] - r[k][s];
            if (NT) __builtin_nontemporal_store(q, &dst[((size_t)w[k] * SEG4 + s) * 64 + lane]);
            else dst[((size_t)w[k] * SEG4 + s) * 64 + lane] = q;
        }
}

__global__ __launch_bounds__(256) void copy_kernel(const vf4 *__restrict__ src, vf4 *__restrict__ dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
        vf4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < n4) v[k] = src[i + k * 256];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < n4) dst[i + k * 256] = v[k];
    }
}

__global__ __launch_bounds__(256) void copy_nt_kernel(const vf4 *__restrict__ src, vf4 *__restrict__ dst, size_t n4)
{
    const size_t stride = (size_t)gridDim.x * 1024;
    for (size_t i = (size_t)blockIdx.x * 1024 + threadIdx.x; i < n4; i += stride) {
        vf4 v[4];
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < n4) v[k] = __builtin_nontemporal_load(&src[i + k * 256]);
#pragma unroll
        for (int k = 0; k < 4; ++k) if (i + k * 256 < n4) __builtin_nontemporal_store(v[k], &dst[i + k * 256]);
    }
}

template <typename F> static float time_ms(F launch, int reps)
{
    hipEvent_t a, b;
    CHECK(hipEventCreate(&a));
    CHECK(hipEventCreate(&b));
    launch();
    CHECK(hipDeviceSynchronize());
    float best = 1e30f;
    for (int r = 0; r < reps; ++r) {
        CHECK(hipEventRecord(a));
        launch();
        CHECK(hipEventRecord(b));
        CHECK(hipEventSynchronize(b));
        float ms;
        CHECK(hipEventElapsedTime(&ms, a, b));
        if (ms < best) best = ms;
    }
    CHECK(hipEventDestroy(a));
    CHECK(hipEventDestroy(b));
    return best;
}

template <int D, int SEG4, bool NT> static void run(const char *name, vf4 *src, vf4 *dst, vf4 *chan, size_t total_bytes, int streams = 0)
{
    // edges = disjoint segments of the arrays; every column owns D of them, randomly placed
    const size_t seg_bytes = (size_t)SEG4 * 1024;
    const size_t n_edges = total_bytes / seg_bytes;
    const int n_cols = (int)(n_edges / D);
    std::vector<int32_t> perm(n_edges);
    std::iota(perm.begin(), perm.end(), 0);
    std::mt19937_64 rng(12345);
    std::shuffle(perm.begin(), perm.end(), rng);
    /* streams = 1: column c's k-th segment is segment c of stream k (D contiguous streams: what a quasi-cyclic code's
     * column groups would read if the edges were stored group-wise); 2: the same in runs of 360 columns whose bases are
     * random (360 = the DVB-S2 group size) */
    if (streams == 1)
        for (int c = 0; c < n_cols; ++c)
            for (int k = 0; k < D; ++k) perm[(size_t)c * D + k] = (int32_t)((size_t)k * n_cols + c);
    if (streams == 2) {
        const int groups = n_cols / 360;
        std::vector<int32_t> base((size_t)groups * D);
        std::iota(base.begin(), base.end(), 0);
        std::shuffle(base.begin(), base.end(), rng);
        for (int c = 0; c < groups * 360; ++c)
            for (int k = 0; k < D; ++k) perm[(size_t)c * D + k] = base[(size_t)(c / 360) * D + k] * 360 + c % 360;
    }
    /* streams = 3: random reads, writes in D contiguous streams; 4: reads in D contiguous streams, random writes
     * (which of the two random sides costs: a layout can move all the randomness to the reads or to the writes) */
    std::vector<int32_t> lin((size_t)n_cols * D);
    for (int c = 0; c < n_cols; ++c)
        for (int k = 0; k < D; ++k) lin[(size_t)c * D + k] = (int32_t)((size_t)k * n_cols + c);
    int32_t *idx, *widx = nullptr;
    CHECK(hipMalloc((void **)&idx, (size_t)n_cols * D * sizeof(int32_t)));
    CHECK(hipMemcpy(idx, streams == 4 ? lin.data() : perm.data(), (size_t)n_cols * D * sizeof(int32_t), hipMemcpyHostToDevice));
    if (streams == 5)      /* random reads, each column's D messages written as ONE run (the product's Q layout) */
        for (size_t i = 0; i < lin.size(); ++i) lin[i] = (int32_t)i;
    if (streams >= 3) {
        CHECK(hipMalloc((void **)&widx, (size_t)n_cols * D * sizeof(int32_t)));
        CHECK(hipMemcpy(widx, streams != 4 ? lin.data() : perm.data(), (size_t)n_cols * D * sizeof(int32_t), hipMemcpyHostToDevice));
    }
    const float ms = time_ms([&] { gather_kernel<D, SEG4, NT><<<(n_cols + 3) / 4, 256>>>(src, dst, chan, idx, n_cols, widx); }, 5);
    if (widx) CHECK(hipFree(widx));
    const double bytes = (double)n_cols * (2.0 * D + 1.0) * seg_bytes;
    printf("%-44s %8.3f ms  %7.1f GB/s\n", name, ms, bytes / (ms * 1e-3) / 1e9);
    CHECK(hipFree(idx));
}

int main()
{
    const size_t total = (size_t)3 << 30;       // message array of 3 GiB, as one class of the headline batch
    vf4 *src, *dst, *chan;
    CHECK(hipMalloc((void **)&src, total));
    CHECK(hipMalloc((void **)&dst, total));
    CHECK(hipMalloc((void **)&chan, total / 2));
    CHECK(hipMemset(src, 0, total));
    CHECK(hipMemset(dst, 0, total));
    CHECK(hipMemset(chan, 0, total / 2));
    {
        const size_t n4 = total / 16;
        const float ms = time_ms([&] { copy_kernel<<<256 * 64, 256>>>(src, dst, n4); }, 5);
        printf("%-44s %8.3f ms  %7.1f GB/s\n", "float4 copy, 3 GiB", ms, 2.0 * total / (ms * 1e-3) / 1e9);
    }
    run<8, 1, true>("gather D=8, 1-KiB segments, nt", src, dst, chan, total);
    run<8, 2, true>("gather D=8, 2-KiB segments, nt", src, dst, chan, total);
    run<8, 1, false>("gather D=8, 1-KiB segments, default policy", src, dst, chan, total);
    run<8, 2, false>("gather D=8, 2-KiB segments, default policy", src, dst, chan, total);
    run<3, 1, true>("gather D=3, 1-KiB segments, nt", src, dst, chan, total);
    run<3, 2, true>("gather D=3, 2-KiB segments, nt", src, dst, chan, total);
    run<3, 4, true>("gather D=3, 4-KiB segments, nt", src, dst, chan, total);
    run<8, 4, true>("gather D=8, 4-KiB segments, nt", src, dst, chan, total);
    run<8, 1, true>("D=8 contiguous streams, nt", src, dst, chan, total, 1);
    run<8, 1, false>("D=8 contiguous streams, default policy", src, dst, chan, total, 1);
    run<3, 1, true>("D=3 contiguous streams, nt", src, dst, chan, total, 1);
    run<8, 1, true>("D=8 random reads, streamed writes, nt", src, dst, chan, total, 3);
    run<8, 1, false>("D=8 random reads, streamed writes, default", src, dst, chan, total, 3);
    run<8, 1, true>("D=8 streamed reads, random writes, nt", src, dst, chan, total, 4);
    run<8, 1, false>("D=8 streamed reads, random writes, default", src, dst, chan, total, 4);
    run<3, 1, true>("D=3 random reads, streamed writes, nt", src, dst, chan, total, 3);
    run<3, 1, true>("D=3 streamed reads, random writes, nt", src, dst, chan, total, 4);
    run<8, 1, true>("D=8 random reads, one write run per column, nt", src, dst, chan, total, 5);
    run<3, 1, true>("D=3 random reads, one write run per column, nt", src, dst, chan, total, 5);
    run<8, 1, true>("D=8 runs of 360 segments, nt", src, dst, chan, total, 2);
    run<8, 1, false>("D=8 runs of 360 segments, default policy", src, dst, chan, total, 2);
    run<3, 1, true>("D=3 runs of 360 segments, nt", src, dst, chan, total, 2);
    {
        const size_t n4 = total / 16;
        const float ms = time_ms([&] { copy_nt_kernel<<<256 * 64, 256>>>(src, dst, n4); }, 5);
        printf("%-44s %8.3f ms  %7.1f GB/s\n", "float4 copy, 3 GiB, nt", ms, 2.0 * total / (ms * 1e-3) / 1e9);
    }
    return 0;
}
