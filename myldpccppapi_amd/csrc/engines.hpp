/*
 * engines.hpp -- the entry points of the one-launch and layered engines that instantiate kernels,
 * compiled in translation units of their own (engine_ldsp.hip, engine_fused.hip, engine_layered.hip)
 * so that the library builds in parallel.  The host driver calls these instead of the inline
 * functions of the kernel headers; plan structs and everything without kernels stay in the headers.
 */
#pragma once

#include <hip/hip_runtime.h>
#include <stdint.h>

#include <vector>

#include "tune.hpp"

namespace ldpc {

struct LdspPlan;
struct FusedPlan;
struct FusedRun;
struct LayeredPlan;
struct LayeredRun;

hipError_t engine_ldsp_plan_create(LdspPlan *pl, int32_t M, int32_t N, int64_t E, const std::vector<int32_t> &row_ptr,
                                   const std::vector<int32_t> &cols, int32_t z, int32_t K, int64_t max_batch, int device,
                                   const Tune &tune, int flood);
hipError_t engine_ldsp_run(LdspPlan *pl, const FusedRun &r, hipStream_t s, int32_t *launched);
hipError_t engine_fused_run(FusedPlan *pl, const FusedRun &r, hipStream_t s, int32_t *launched);
hipError_t engine_layered_run(LayeredPlan *pl, const LayeredRun &r, hipStream_t s, int32_t *launched);

}  // namespace ldpc
