/* flood_ms16.hip -- min-sum with fp16 message storage: instantiations of the streaming flooding kernels. */
#include "flood_tables_impl.hpp"
namespace ldpc { void fill_flood_ms16(int V, FloodFns *f) { tables::fill<kAlgoMS, hf>(V, f); } }
