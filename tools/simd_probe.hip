// simd_probe.hip -- where does the dispatcher put the waves of a workgroup?  Every wave records the SIMD, CU,
// shader engine and XCC it runs on (s_getreg HW_ID / XCC_ID) and then stays resident for a while, so that
// the workgroups that share a CU are there together.  Printed: for workgroup shapes of 6, 8 and 12 waves
// with 40 / 80 KB of LDS (the shapes of layered_ldsp_kernel at BG1, Z = 384), the histogram of
// waves-per-SIMD patterns of single workgroups and of whole CUs.
//
//   hipcc -O3 --offload-arch=gfx950 tools/simd_probe.hip -o /tmp/simd_probe && /tmp/simd_probe
#include <hip/hip_runtime.h>

#include <cstdint>
#include <cstdio>
#include <cstdlib>
#include <map>
#include <string>
#include <vector>

#define CHECK(x)                                                                         \
    do {                                                                                 \
        hipError_t e_ = (x);                                                             \
        if (e_ != hipSuccess) { fprintf(stderr, "%s: %s\n", #x, hipGetErrorString(e_)); exit(1); } \
    } while (0)

// HW_REG_HW_ID = 4 (all 32 bits), HW_REG_XCC_ID = 20 (bits 3:0): simm16 = (size - 1) << 11 | offset << 6 | id
#define GETREG(id, off, size) __builtin_amdgcn_s_getreg((((size) - 1) << 11) | ((off) << 6) | (id))

// drop = 0: all waves stay.  drop = 1: workgroups take a ticket per CU (global counter) and two of the eight
// waves leave at once: the last two for even tickets, waves 4 and 5 for odd ones.
__global__ __attribute__((amdgpu_waves_per_eu(6))) void probe(uint32_t *out, uint32_t *tickets, int drop, long long spin)
{
    extern __shared__ float lds[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const uint32_t hw = GETREG(4, 0, 32), xcc = GETREG(20, 0, 4);
    const long long t0 = __builtin_amdgcn_s_memtime();
    uint32_t ticket = 0;
    if (drop) {
        // CU identity: xcc (4 bits) | se (bits 15:13) | sh (bit 12) | cu (bits 11:8)
        const uint32_t cu = (xcc << 8) | ((hw >> 8) & 0xffu);
        if (threadIdx.x == 0) lds[0] = __uint_as_float(atomicAdd(&tickets[cu], 1u));
        __syncthreads();
        ticket = __float_as_uint(lds[0]);
        const bool odd = ticket & 1u;
        if (odd ? (wave == 4 || wave == 5) : (wave == 6 || wave == 7)) return;
    }
    if (lane == 0) {
        uint32_t *o = out + ((size_t)blockIdx.x * 16 + wave) * 4;
        o[0] = hw; o[1] = xcc; o[2] = ticket; o[3] = 1u;
    }
    while (__builtin_amdgcn_s_memtime() - t0 < spin) __builtin_amdgcn_s_sleep(32);
    if (lds[threadIdx.x] == 12345.0f) out[0] = 0;          // keep the LDS allocation
}

static void run(const char *name, int waves, size_t lds, int per_cu, int drop)
{
    int cus = 0;
    CHECK(hipDeviceGetAttribute(&cus, hipDeviceAttributeMultiprocessorCount, 0));
    const int grid = cus * per_cu;
    uint32_t *out, *tickets;
    CHECK(hipMalloc((void **)&out, (size_t)grid * 16 * 4 * sizeof(uint32_t)));
    CHECK(hipMemset(out, 0, (size_t)grid * 16 * 4 * sizeof(uint32_t)));
    CHECK(hipMalloc((void **)&tickets, 65536 * sizeof(uint32_t)));
    CHECK(hipMemset(tickets, 0, 65536 * sizeof(uint32_t)));
    CHECK(hipFuncSetAttribute((const void *)probe, hipFuncAttributeMaxDynamicSharedMemorySize, 160 * 1024 - 512));
    probe<<<grid, waves * 64, lds>>>(out, tickets, drop, 20000000LL);     // 100 MHz counter: 0.2 s
    CHECK(hipDeviceSynchronize());
    std::vector<uint32_t> h((size_t)grid * 16 * 4);
    CHECK(hipMemcpy(h.data(), out, h.size() * sizeof(uint32_t), hipMemcpyDeviceToHost));
    std::map<std::string, int> wg_pat, cu_pat;
    std::map<uint32_t, std::vector<int>> cu_simd;
    std::map<uint32_t, int> cu_wgs;
    for (int b = 0; b < grid; ++b) {
        int s[4] = {0, 0, 0, 0};
        uint32_t cu = 0;
        for (int w = 0; w < 16; ++w) {
            const uint32_t *o = &h[((size_t)b * 16 + w) * 4];
            if (!o[3]) continue;
            const int simd = (o[0] >> 4) & 3;
            cu = (o[1] << 8) | ((o[0] >> 8) & 0xffu);
            ++s[simd];
            auto &v = cu_simd[cu];
            v.resize(4);
            ++v[simd];
        }
        ++cu_wgs[cu];
        char buf[64];
        snprintf(buf, sizeof buf, "%d-%d-%d-%d", s[0], s[1], s[2], s[3]);
        ++wg_pat[buf];
    }
    for (auto &kv : cu_simd) {
        char buf[64];
        snprintf(buf, sizeof buf, "%d-%d-%d-%d (%d wg)", kv.second[0], kv.second[1], kv.second[2], kv.second[3], cu_wgs[kv.first]);
        ++cu_pat[buf];
    }
    printf("%s: grid %d x %d threads, %zu B LDS, %zu CUs seen\n  per workgroup:", name, grid, waves * 64, lds, cu_simd.size());
    for (auto &kv : wg_pat) printf("  %s x%d", kv.first.c_str(), kv.second);
    printf("\n  per CU:");
    for (auto &kv : cu_pat) printf("  %s x%d", kv.first.c_str(), kv.second);
    printf("\n");
    CHECK(hipFree(out));
    CHECK(hipFree(tickets));
}

int main()
{
    run("6 waves, 40 KB, 3 per CU", 6, 39952, 3, 0);
    run("6 waves, 40 KB, 4 per CU", 6, 39952, 4, 0);
    run("12 waves, 80 KB, 2 per CU", 12, 79888, 2, 0);
    run("8 waves, 40 KB, 3 per CU", 8, 39952, 3, 0);
    run("8 waves of which 2 leave (by CU ticket), 40 KB, 3 per CU", 8, 39952, 3, 1);
    run("8 waves of which 2 leave (by CU ticket), 40 KB, 4 per CU", 8, 39952, 4, 1);
    return 0;
}
