/* engine_layered.hip -- the streaming layered kernels, one launch per layer (layered_kernels.hpp). */
#define LDPC_ENGINE_LAYERED
#include "layered_kernels.hpp"
#include "engines.hpp"
namespace ldpc {
hipError_t engine_layered_run(LayeredPlan *pl, const LayeredRun &r, hipStream_t s, int32_t *launched) { return layered_run(pl, r, s, launched); }
}  // namespace ldpc
