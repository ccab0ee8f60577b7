/* engine_ldsp.hip -- the record kernels (ldsp_kernels.hpp): posteriors in LDS, check records in cache. */
#define LDPC_ENGINE_LDSP
#include "ldsp_kernels.hpp"
#include "engines.hpp"
namespace ldpc {
hipError_t engine_ldsp_plan_create(LdspPlan *pl, int32_t M, int32_t N, int64_t E, const std::vector<int32_t> &row_ptr,
                                   const std::vector<int32_t> &cols, int32_t z, int32_t K, int64_t max_batch, int device,
                                   const Tune &tune, int flood)
{
    return ldsp_plan_create(pl, M, N, E, row_ptr, cols, z, K, max_batch, device, tune, flood);
}
hipError_t engine_ldsp_run(LdspPlan *pl, const FusedRun &r, hipStream_t s, int32_t *launched) { return ldsp_run(pl, r, s, launched); }
}  // namespace ldpc
