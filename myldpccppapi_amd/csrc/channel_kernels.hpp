/*
 * channel_kernels.hpp -- the reference's test harness pieces that sit either side of decode(),
 * on the device: Coder::test (BPSK + Gaussian noise, MyLdpc.cpp:1061-1078) and the error count of
 * Test.cpp:105-110.  The noise source is the counter-based generator of ldpc_channel.h, so a
 * batch's channel values are produced where they are consumed (no 1.06 GB host-to-device copy per
 * batch) and any frame range can be regenerated on any rank.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ldpc_channel.h"

namespace ldpc {

/* one thread = the four samples of one Philox call: frame f, samples 4g .. 4g+3 */
__global__ __launch_bounds__(256) void awgn_kernel(float *__restrict__ llr, const uint8_t *__restrict__ bits,
                                                   int64_t frames, int32_t N, int32_t groups, float sd,
                                                   uint64_t seed, int64_t first_frame)
{
    const int64_t t = (int64_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (t >= frames * groups) return;
    const int64_t f = t / groups;
    const int32_t g = (int32_t)(t - f * groups);
    double z[4];
    ldpc_ch_normal4(seed, (uint64_t)(first_frame + f), (uint32_t)g, z);
    const size_t base = (size_t)f * N + (size_t)g * 4;
    float v[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        const int n = g * 4 + i;
        const int bit = (bits && n < N) ? (bits[base + i] & 1) : 0;
        v[i] = ldpc_ch_sample(bit, sd, z[i]);
    }
    if ((N & 3) == 0) {
        *reinterpret_cast<float4 *>(llr + base) = float4{v[0], v[1], v[2], v[3]};
    } else {
#pragma unroll
        for (int i = 0; i < 4; ++i)
            if (g * 4 + i < N) llr[base + i] = v[i];
    }
}

/* differing bits, bytes (the reference's ErrNum, Test.cpp:105-110) and frames between two packed
 * outputs; ref == nullptr compares with the all-zero codeword.  totals[0..2] += */
__global__ __launch_bounds__(256) void count_errors_kernel(const uint8_t *__restrict__ out, const uint8_t *__restrict__ ref,
                                                           int64_t frames, int64_t bytes_per_frame,
                                                           unsigned long long *__restrict__ totals)
{
    const int64_t f = blockIdx.x;
    if (f >= frames) return;
    unsigned bit_err = 0, byte_err = 0;
    for (int64_t j = threadIdx.x; j < bytes_per_frame; j += blockDim.x) {
        const size_t i = (size_t)f * bytes_per_frame + j;
        const unsigned x = (unsigned)(out[i] ^ (ref ? ref[i] : (uint8_t)0));
        bit_err += __popc(x);
        byte_err += x != 0;
    }
    __shared__ unsigned s_bits[4], s_bytes[4];
    for (int o = 32; o > 0; o >>= 1) {
        bit_err += __shfl_down(bit_err, o);
        byte_err += __shfl_down(byte_err, o);
    }
    if ((threadIdx.x & 63) == 0) { s_bits[threadIdx.x >> 6] = bit_err; s_bytes[threadIdx.x >> 6] = byte_err; }
    __syncthreads();
    if (threadIdx.x == 0) {
        const unsigned b = s_bits[0] + s_bits[1] + s_bits[2] + s_bits[3];
        const unsigned y = s_bytes[0] + s_bytes[1] + s_bytes[2] + s_bytes[3];
        if (b) {
            atomicAdd(&totals[0], (unsigned long long)b);
            atomicAdd(&totals[1], (unsigned long long)y);
            atomicAdd(&totals[2], 1ull);
        }
    }
}

}  // namespace ldpc
