/*
 * flood_tables.hpp -- the kernel tables of the streaming flooding decoders, filled per arithmetic in
 * translation units of their own (flood_sp.hip, flood_ms.hip, flood_ms16.hip) so that the library
 * builds in parallel: the host driver (ldpc_hip.hip) only sees function pointers.
 */
#pragma once

#include "flood_kernels.hpp"

namespace ldpc {

using CheckFn = void (*)(const CheckArgs);
using VarFn = void (*)(const VarArgs);
using LinkFn = void (*)(const CheckArgs, const LinkArgs);
/* group launches (several degree classes of a bucket in one launch, flood_kernels.hpp) */
using CheckGroupFn = void (*)(const CheckArgs, const GroupClass *, int);
using VarGroupFn = void (*)(const VarArgs, const GroupClass *, int);
using InitFn = void (*)(const InitArgs);

constexpr int kVarBuckets = 3, kCheckBuckets = 4;
constexpr int kVarBucketLo[kVarBuckets] = {1, 5, 9}, kVarBucketHi[kVarBuckets] = {4, 8, 16};
constexpr int kCheckBucketLo[kCheckBuckets] = {1, 9, 17, 25}, kCheckBucketHi[kCheckBuckets] = {8, 16, 24, 32};

struct FloodFns {
    CheckFn check[kMaxUnrolledCheckDegreeMS + 1] = {};        /* narrow waves (1 value per lane) */
    CheckFn check_wide[kMaxUnrolledCheckDegreeMS + 1] = {};   /* V values per lane */
    LinkFn link[kMaxUnrolledDegree + 1] = {};                 /* wide waves */
    LinkFn link_narrow[kMaxUnrolledDegree + 1] = {};          /* narrow waves */
    LinkFn link_deep[kMaxUnrolledDegree + 1] = {};            /* narrow waves, inputs two rows ahead */
    LinkFn link_half[kMaxUnrolledDegree + 1] = {};            /* 2 values per lane (V = 4) */
    VarFn var[kMaxUnrolledDegree + 1] = {};
    CheckGroupFn check_group[kCheckBuckets] = {};
    int check_group_width = 1;                                /* values per lane of the group check kernels */
    VarGroupFn var_group[kVarBuckets] = {};
    InitFn init = nullptr;
    int max_check_unrolled = kMaxUnrolledDegree;
};

/* V = frames per lane (1, 2 or 4) */
void fill_flood_sp(int V, FloodFns *f);      /* sum-product, fp32            (flood_sp.hip)   */
void fill_flood_ms(int V, FloodFns *f);      /* min-sum, fp32 messages       (flood_ms.hip)   */
void fill_flood_ms16(int V, FloodFns *f);    /* min-sum, fp16 message storage (flood_ms16.hip) */

}  // namespace ldpc
