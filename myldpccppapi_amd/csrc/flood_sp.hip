/* flood_sp.hip -- sum-product (probability domain, fp32) instantiations of the streaming flooding kernels. */
#include "flood_tables_impl.hpp"
namespace ldpc { void fill_flood_sp(int V, FloodFns *f) { tables::fill<kAlgoSP, float>(V, f); } }
