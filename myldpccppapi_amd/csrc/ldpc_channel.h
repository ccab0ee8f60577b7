/*
 * ldpc_channel.h -- the test channel of the reference (Coder::test, MyLdpc.cpp:1061-1078: BPSK,
 * bit 0 -> +1, bit 1 -> -1, plus N(0, sd^2) noise) with a COUNTER-BASED noise source, shared by
 * host and device code.
 *
 * The reference draws its noise from libc rand() seeded with time(0) (Test.cpp:29,
 * MyLdpc.cpp:1093-1105): sequential, not reproducible, host only -- at the benchmark's size the
 * 1.06 GB of channel values then has to cross PCIe for every batch.  Here sample n of frame f is
 * a pure function of (seed, f, n): any frame range can be produced on any GPU (or on the host)
 * independently and identically, which is what the sharded BER sweeps and the tests need.
 *
 *   Philox4x32-10 (Salmon et al., SC'11), counter = (n / 4, f low, f high, "LDPC"), key = seed
 *   -> 4 x 32 random bits -> two Box-Muller pairs -> 4 standard normals.
 *
 * Everything after the integer generator is IEEE double arithmetic built from + - * / only (own
 * log, square root and sine/cosine kernels; no libm, no FMA contraction), so the same source
 * gives the same float on x86 and on gfx950.  Accuracy of the normals is ~1e-15, far below the
 * float they are rounded to.
 */
#ifndef LDPC_CHANNEL_H_
#define LDPC_CHANNEL_H_

#include <stdint.h>

#if defined(__HIPCC__)
#define LDPC_CH_HD __host__ __device__ __forceinline__
#else
#define LDPC_CH_HD static inline
#endif

LDPC_CH_HD void ldpc_philox4x32_10(uint32_t c[4], uint32_t k0, uint32_t k1)
{
    for (int round = 0; round < 10; ++round) {
        const uint64_t p0 = (uint64_t)0xD2511F53u * c[0];
        const uint64_t p1 = (uint64_t)0xCD9E8D57u * c[2];
        const uint32_t n0 = (uint32_t)(p1 >> 32) ^ c[1] ^ k0;
        const uint32_t n2 = (uint32_t)(p0 >> 32) ^ c[3] ^ k1;
        c[0] = n0;
        c[1] = (uint32_t)p1;
        c[2] = n2;
        c[3] = (uint32_t)p0;
        k0 += 0x9E3779B9u;
        k1 += 0xBB67AE85u;
    }
}

LDPC_CH_HD double ldpc_ch_bits_to_double(uint64_t b)
{
    union { uint64_t u; double d; } v;
    v.u = b;
    return v.d;
}

LDPC_CH_HD uint64_t ldpc_ch_double_to_bits(double d)
{
    union { uint64_t u; double d; } v;
    v.d = d;
    return v.u;
}

/* -2 ln(u) for u = (x + 0.5) / 2^32, x a 32-bit integer: u = m * 2^e with m in [1/sqrt2, sqrt2),
 * ln m = 2 atanh(s), s = (m - 1) / (m + 1), |s| < 0.1716 */
LDPC_CH_HD double ldpc_ch_minus2log(uint32_t x)
{
    const double u = ((double)x + 0.5) * (1.0 / 4294967296.0);
    uint64_t b = ldpc_ch_double_to_bits(u);
    int e = (int)((b >> 52) & 0x7ff) - 1023;
    b = (b & 0x000fffffffffffffULL) | 0x3ff0000000000000ULL;        /* m in [1, 2) */
    double m = ldpc_ch_bits_to_double(b);
    if (m > 1.4142135623730951) { m = m * 0.5; e += 1; }
    const double s = (m - 1.0) / (m + 1.0), s2 = s * s;
    double p = 1.0 / 25.0;
    p = p * s2 + 1.0 / 23.0;
    p = p * s2 + 1.0 / 21.0;
    p = p * s2 + 1.0 / 19.0;
    p = p * s2 + 1.0 / 17.0;
    p = p * s2 + 1.0 / 15.0;
    p = p * s2 + 1.0 / 13.0;
    p = p * s2 + 1.0 / 11.0;
    p = p * s2 + 1.0 / 9.0;
    p = p * s2 + 1.0 / 7.0;
    p = p * s2 + 1.0 / 5.0;
    p = p * s2 + 1.0 / 3.0;
    p = p * s2 + 1.0;
    const double lnu = (double)e * 0.6931471805599453 + 2.0 * s * p;
    return -2.0 * lnu;
}

/* square root of a positive finite double by Newton's iteration on division (deterministic on
 * every IEEE machine; the start value halves the exponent) */
LDPC_CH_HD double ldpc_ch_sqrt(double a)
{
    if (!(a > 0.0)) return 0.0;
    double x = ldpc_ch_bits_to_double((ldpc_ch_double_to_bits(a) >> 1) + 0x1ff8000000000000ULL);
    x = 0.5 * (x + a / x);
    x = 0.5 * (x + a / x);
    x = 0.5 * (x + a / x);
    x = 0.5 * (x + a / x);
    x = 0.5 * (x + a / x);
    return x;
}

/* cos and sin of 2 pi (x + 0.5) / 2^32: the octant comes from the top three bits, the angle
 * inside it (0 < t < pi/4) feeds two Taylor polynomials */
LDPC_CH_HD void ldpc_ch_cossin(uint32_t x, double *c, double *s)
{
    const uint32_t oct = x >> 29;
    const double t = ((double)(x & 0x1fffffffu) + 0.5) * (0.7853981633974483 / 536870912.0);
    const double t2 = t * t;
    double sp = -1.0 / 1307674368000.0;                            /* -1/15! */
    sp = sp * t2 + 1.0 / 6227020800.0;
    sp = sp * t2 - 1.0 / 39916800.0;
    sp = sp * t2 + 1.0 / 362880.0;
    sp = sp * t2 - 1.0 / 5040.0;
    sp = sp * t2 + 1.0 / 120.0;
    sp = sp * t2 - 1.0 / 6.0;
    sp = sp * t2 + 1.0;
    const double sn = sp * t;
    double cp = 1.0 / 20922789888000.0;                            /* 1/16! */
    cp = cp * t2 - 1.0 / 87178291200.0;
    cp = cp * t2 + 1.0 / 479001600.0;
    cp = cp * t2 - 1.0 / 3628800.0;
    cp = cp * t2 + 1.0 / 40320.0;
    cp = cp * t2 - 1.0 / 720.0;
    cp = cp * t2 + 1.0 / 24.0;
    cp = cp * t2 - 0.5;
    const double cs = cp * t2 + 1.0;
    /* angle = oct * pi/4 + t; odd octants mirror: pi/4 - t' is not needed because t already
     * runs upwards inside every octant, so use the addition formulas for k * pi/4 */
    const double h = 0.7071067811865476;                           /* cos(pi/4) */
    double c0, s0;
    switch (oct & 1u) {
    case 0: c0 = cs; s0 = sn; break;
    default: c0 = (cs - sn) * h; s0 = (sn + cs) * h; break;        /* + pi/4 */
    }
    switch (oct >> 1) {                                            /* + k * pi/2 */
    case 0: *c = c0; *s = s0; break;
    case 1: *c = -s0; *s = c0; break;
    case 2: *c = -c0; *s = -s0; break;
    default: *c = s0; *s = -c0; break;
    }
}

/* the four standard normals of counter (group, frame) */
LDPC_CH_HD void ldpc_ch_normal4(uint64_t seed, uint64_t frame, uint32_t group, double z[4])
{
    uint32_t c[4] = {group, (uint32_t)frame, (uint32_t)(frame >> 32), 0x4c445043u};
    ldpc_philox4x32_10(c, (uint32_t)seed, (uint32_t)(seed >> 32));
    for (int h = 0; h < 2; ++h) {
        const double r = ldpc_ch_sqrt(ldpc_ch_minus2log(c[2 * h]));
        double cs, sn;
        ldpc_ch_cossin(c[2 * h + 1], &cs, &sn);
        z[2 * h] = r * cs;
        z[2 * h + 1] = r * sn;
    }
}

/* channel value of one sample: BPSK symbol of `bit` plus sd * z, rounded once to float */
LDPC_CH_HD float ldpc_ch_sample(int bit, float sd, double z)
{
    return (float)((bit ? -1.0 : 1.0) + (double)sd * z);
}

#endif /* LDPC_CHANNEL_H_ */
