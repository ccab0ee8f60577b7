/* engine_fused.hip -- the LDS-resident one-launch decoders for short quasi-cyclic codes (fused_kernels.hpp). */
#define LDPC_ENGINE_FUSED
#include "fused_kernels.hpp"
#include "engines.hpp"
namespace ldpc {
hipError_t engine_fused_run(FusedPlan *pl, const FusedRun &r, hipStream_t s, int32_t *launched) { return fused_run(pl, r, s, launched); }
}  // namespace ldpc
