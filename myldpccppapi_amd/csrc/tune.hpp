/*
 * tune.hpp -- the tuning fields of ldpc_decoder_config (include/ldpc_hip.h) unpacked once per
 * decoder.  Kernel selection and launch shapes only: results never depend on them.  The library
 * reads no environment variables; harnesses translate their own switches into these fields.
 */
#pragma once

#include "../../include/ldpc_hip.h"

namespace ldpc {

struct Tune {
    /* tri-state: 0 automatic, 1 forced on, 2 forced off */
    int fused = 0, ldsp = 0, ldsp_ext = 0, ldsp_pack = 0, link_narrow = 0, check_wide = 0, syn_xcd = 0,
        fused_pack = 0, fused_loop = 0, device_tail = 0, merge = 0, link_deep = 0, link_half = 0, link_guided = 0, tiles_first = 0;
    int rows_per_wave = 0, cols_per_wave = 0;
    int link_rows = 0;          /* 0 automatic, -1 fusion off */
    int compact = 0;            /* 0 automatic, -1 off */
    int ldsp_grid = 0, ldsp_per_cu = 0, ldsp_waves = 0;
    int place = 0;              /* array sets timed at creation: 0 automatic, 1 none, 2..8 */
    int q_order = 0;            /* Q stored in the order its writers produce it: 0 automatic (on), 1 on, -1 off */
};

inline Tune tune_from_config(const ldpc_decoder_config &c)
{
    Tune t;
    auto f = [&](int field) { return (c.tune_flags >> field) & 3; };
    t.fused = f(LDPC_TUNE_FUSED);
    t.ldsp = f(LDPC_TUNE_LDSP);
    t.ldsp_ext = f(LDPC_TUNE_LDSP_EXT);
    t.ldsp_pack = f(LDPC_TUNE_LDSP_PACK);
    t.link_narrow = f(LDPC_TUNE_LINK_NARROW);
    t.check_wide = f(LDPC_TUNE_CHECK_WIDE);
    t.syn_xcd = f(LDPC_TUNE_SYN_XCD);
    t.fused_pack = f(LDPC_TUNE_FUSED_PACK);
    t.fused_loop = f(LDPC_TUNE_FUSED_LOOP);
    t.device_tail = f(LDPC_TUNE_DEVICE_TAIL);
    t.merge = f(LDPC_TUNE_MERGE);
    t.link_deep = f(LDPC_TUNE_LINK_DEEP);
    t.link_half = f(LDPC_TUNE_LINK_HALF);
    t.link_guided = f(LDPC_TUNE_LINK_GUIDED);
    t.tiles_first = f(LDPC_TUNE_TILES_FIRST);
    t.rows_per_wave = c.tune_rows_per_wave;
    t.cols_per_wave = c.tune_cols_per_wave;
    t.link_rows = c.tune_link_rows;
    t.compact = c.tune_compact;
    t.ldsp_grid = c.tune_ldsp_grid;
    t.ldsp_per_cu = c.tune_ldsp_shape & 255;
    t.ldsp_waves = (c.tune_ldsp_shape >> 8) & 255;
    t.place = c.tune_place;
    t.q_order = c.tune_q_order;
    return t;
}

/* the choice a tri-state field makes when the automatic answer is `dflt` */
inline bool tune_pick(int tri, bool dflt) { return tri == 1 ? true : (tri == 2 ? false : dflt); }
inline bool tune_forced_on(int tri) { return tri == 1; }
inline bool tune_forced_off(int tri) { return tri == 2; }

/* field 3 (both bits) is not a value */
inline bool tune_valid(const ldpc_decoder_config &c)
{
    for (int field = 0; field <= LDPC_TUNE_TILES_FIRST; field += 2)
        if (((c.tune_flags >> field) & 3) == 3) return false;
    if (c.tune_flags >> (LDPC_TUNE_TILES_FIRST + 2)) return false;
    return c.tune_rows_per_wave >= 0 && c.tune_rows_per_wave <= 4096 && c.tune_cols_per_wave >= 0 &&
           c.tune_cols_per_wave <= 4096 && c.tune_link_rows >= -1 && c.tune_link_rows <= 4096 &&
           c.tune_compact >= -1 && c.tune_ldsp_grid >= 0 && c.tune_ldsp_shape >= 0 && c.tune_ldsp_shape < 65536 &&
           c.tune_place >= 0 && c.tune_place <= 8 && c.host_input >= 0 && c.host_input <= 2 && c.host_copy_threads >= 0 && c.host_copy_threads <= 16 &&
           c.tune_q_order >= -1 && c.tune_q_order <= 1;
}

}  // namespace ldpc
