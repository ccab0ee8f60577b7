/* flood_ms.hip -- min-sum (fp32 messages) instantiations of the streaming flooding kernels. */
#include "flood_tables_impl.hpp"
namespace ldpc { void fill_flood_ms(int V, FloodFns *f) { tables::fill<kAlgoMS, float>(V, f); } }
