/* The C ABI from plain C (C99): the header must compile as C, the library must link, and the host-only
 * entry points must work without a GPU.  Exit code 0 = all good. */
#include <stdio.h>
#include <string.h>

#include "ldpc_hip.h"

int main(void)
{
    /* H = [1 1 0; 0 1 1] in row-major order */
    const int32_t rows[4] = {0, 0, 1, 1}, cols[4] = {0, 1, 1, 2};
    ldpc_graph *g = NULL;
    int32_t M = 0, N = 0, rd = 0, cd = 0;
    int64_t E = 0, lo = -1, hi = -1;
    ldpc_decoder_config cfg;

    if (ldpc_abi_version() != LDPC_HIP_ABI_VERSION) return 1;
    if (ldpc_graph_create(rows, cols, 4, 2, 3, &g) != LDPC_OK) return 2;
    if (ldpc_graph_info(g, &M, &N, &E, &rd, &cd) != LDPC_OK || M != 2 || N != 3 || E != 4 || rd != 2 || cd != 2) return 3;
    ldpc_decoder_config_init(&cfg);
    if (cfg.struct_size != sizeof cfg || cfg.max_iter != 40 || cfg.llr_scale != 8.0f || cfg.tune_flags != 0) return 4;
    cfg.tune_flags = LDPC_TUNE_OFF(LDPC_TUNE_MERGE) | LDPC_TUNE_ON(LDPC_TUNE_LINK_NARROW);
    if (ldpc_shard_range(10, 1, 3, 1, &lo, &hi) != LDPC_OK || lo != 4 || hi != 7) return 5;
    if (ldpc_out_bytes(324, 9, LDPC_PACK_BYTES) != 8 * 324 / 8 + 40) return 6;
    if (ldpc_graph_create(cols, rows, 4, 2, 3, &g) == LDPC_OK) return 7;        /* not row-major: refused ... */
    if (strlen(ldpc_last_error()) == 0) return 8;                               /* ... with a message */
    ldpc_graph_destroy(g);
    puts("cabi ok");
    return 0;
}
