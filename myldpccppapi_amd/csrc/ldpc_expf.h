/*
 * ldpc_expf.h -- one single-precision exp() shared by host and device code.
 *
 * The reference's sum-product initialisation is `exp(8*code)` in OpenCL C
 * (decodeCL.c:9,15).  OpenCL leaves exp() accuracy to the device runtime, so
 * the reference is not bit-reproducible across devices at this point.  We pin
 * it: ldpc_expf() evaluates exp() in IEEE double arithmetic only (mul, add,
 * fma, integer bit operations), so the same source gives the same float on the
 * x86 host and on gfx950, and it is the algorithm glibc >= 2.27 uses for
 * expf() (exp2f-style: 32-entry table of 2^(i/32) + a cubic in double), so it
 * agrees with the host libm the reference kernels get when they are compiled
 * as host C for the oracle (oracle/ref_host).  tools/check_expf.c compares it
 * with libm's expf over all 2^32 float inputs.
 *
 * The table is 2^(i/32) rounded to double, stored as bits minus (i << 47) so
 * that adding (k << 47) yields 2^(k/32) for any integer k in range.
 */
#ifndef LDPC_EXPF_H_
#define LDPC_EXPF_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define LDPC_HD __host__ __device__ __forceinline__
#else
#define LDPC_HD static inline
#endif

LDPC_HD uint64_t ldpc_exp2_tab(unsigned i)
{
    /* a switch rather than a static array so it is usable from device code
     * without a __constant__ symbol; compiles to a table either way. */
    switch (i & 31u) {
    case 0: return 0x3ff0000000000000ULL;  case 1: return 0x3fefd9b0d3158574ULL;
    case 2: return 0x3fefb5586cf9890fULL;  case 3: return 0x3fef9301d0125b51ULL;
    case 4: return 0x3fef72b83c7d517bULL;  case 5: return 0x3fef54873168b9aaULL;
    case 6: return 0x3fef387a6e756238ULL;  case 7: return 0x3fef1e9df51fdee1ULL;
    case 8: return 0x3fef06fe0a31b715ULL;  case 9: return 0x3feef1a7373aa9cbULL;
    case 10: return 0x3feedea64c123422ULL; case 11: return 0x3feece086061892dULL;
    case 12: return 0x3feebfdad5362a27ULL; case 13: return 0x3feeb42b569d4f82ULL;
    case 14: return 0x3feeab07dd485429ULL; case 15: return 0x3feea47eb03a5585ULL;
    case 16: return 0x3feea09e667f3bcdULL; case 17: return 0x3fee9f75e8ec5f74ULL;
    case 18: return 0x3feea11473eb0187ULL; case 19: return 0x3feea589994cce13ULL;
    case 20: return 0x3feeace5422aa0dbULL; case 21: return 0x3feeb737b0cdc5e5ULL;
    case 22: return 0x3feec49182a3f090ULL; case 23: return 0x3feed503b23e255dULL;
    case 24: return 0x3feee89f995ad3adULL; case 25: return 0x3feeff76f2fb5e47ULL;
    case 26: return 0x3fef199bdd85529cULL; case 27: return 0x3fef3720dcef9069ULL;
    case 28: return 0x3fef5818dcfba487ULL; case 29: return 0x3fef7c97337b9b5fULL;
    case 30: return 0x3fefa4afa2a490daULL; default: return 0x3fefd0765b6e4540ULL;
    }
}

LDPC_HD float ldpc_expf(float x)
{
    union { float f; uint32_t u; } fx; fx.f = x;
    const uint32_t ax = fx.u & 0x7fffffffu;
    if (ax >= 0x42b00000u) {                 /* |x| >= 88 or NaN/inf */
        if (fx.u == 0xff800000u) return 0.0f;            /* -inf */
        if (ax >= 0x7f800000u) return x + x;             /* +inf, NaN */
        if (x > 0x1.62e42ep6f) return __builtin_inff();  /* overflow */
        if (x < -0x1.9fe368p6f) return 0.0f;             /* underflow to 0 */
        /* else fall through: results down to the subnormal range come out of
         * the double computation and the final double->float rounding */
    }
    const double invln2n = 0x1.71547652b82fep+0 * 32.0;
    const double shift = 0x1.8p+52;
    const double c0 = 0x1.c6af84b912394p-5 / 32.0 / 32.0 / 32.0;
    const double c1 = 0x1.ebfce50fac4f3p-3 / 32.0 / 32.0;
    const double c2 = 0x1.62e42ff0c52d6p-1 / 32.0;
    double z = invln2n * (double)x;
    union { double d; uint64_t u; } kd; kd.d = z + shift;
    const uint64_t ki = kd.u;
    /* glibc's x86-64 FMA build (selected by ifunc on every CPU with FMA) fuses
     * the scaling product into this subtraction; that one fusion decides two of
     * the 2^32 inputs (0x4202422f, 0xc27c65d9).  Whether the three polynomial
     * multiply-adds below are fused changes no float result (all eight
     * combinations checked exhaustively), so they stay plain. */
    const double r = __builtin_fma(invln2n, (double)x, -(kd.d - shift));
    union { double d; uint64_t u; } s;
    s.u = ldpc_exp2_tab((unsigned)ki) + (ki << 47);
    const double p = c0 * r + c1;
    const double r2 = r * r;
    double y = c2 * r + 1.0;
    y = p * r2 + y;
    y = y * s.d;
    return (float)y;
}

#endif /* LDPC_EXPF_H_ */
