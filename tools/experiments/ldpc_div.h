/*
 * ldpc_div.h -- the IEEE fp32 division the compiler emits (-fhip-fp32-correctly-rounded-divide-sqrt), taken apart.
 *
 * hipcc expands a / b into eleven instructions (checked in the ISA, tools/kernel_isa_counts.py):
 *     ds = v_div_scale(b, b, a)      ns = v_div_scale(a, b, a) -> vcc
 *     r  = v_rcp(ds)   e = fma(-ds, r, 1)   r1 = fma(e, r, r)
 *     q  = ns * r1     e2 = fma(-ds, q, ns) q2 = fma(e2, r1, q)   e3 = fma(-ds, q2, ns)
 *     q3 = v_div_fmas(e3, r1, q2)    result = v_div_fixup(q3, b, a)
 * For operands in the domain  2^-60 <= a <= b <= 2^60  the three helpers do nothing: v_div_scale
 * returns its first operand and clears vcc unless an operand is zero or denormal, the exponents differ
 * by 96 or more, or 1/b or a/b would be denormal; v_div_fmas with vcc clear IS the fma; v_div_fixup
 * returns |q3| with the sign of a*b for finite non-zero operands and a normal quotient.  What remains are the
 * eight operations below, and two quotients with one denominator share the first three.  Same
 * instructions on the same values: the same bits, by construction; ldpc_selftest_division (C ABI) compares
 * them with the compiler's division on billions of operand pairs of the domain anyway.
 */
#pragma once

#include <hip/hip_runtime.h>

namespace ldpc {

/* the refined reciprocal r1 of a denominator in the domain */
__device__ __forceinline__ float div_reciprocal(float den)
{
    const float r = __builtin_amdgcn_rcpf(den);
    const float e = __builtin_fmaf(-den, r, 1.0f);
    return __builtin_fmaf(e, r, r);
}

/* num / den given r1 = div_reciprocal(den); num, den in the domain */
__device__ __forceinline__ float div_with_reciprocal(float num, float den, float r1)
{
    const float q = num * r1;
    const float e2 = __builtin_fmaf(-den, q, num);
    const float q2 = __builtin_fmaf(e2, r1, q);
    const float e3 = __builtin_fmaf(-den, q2, num);
    return __builtin_fmaf(e3, r1, q2);
}

constexpr float kDivDomainLo = 0x1p-59f;      /* one binade inside the domain: products of up to 24 roundings stay in it */
constexpr float kDivDomainHi = 0x1p59f;

}  // namespace ldpc
