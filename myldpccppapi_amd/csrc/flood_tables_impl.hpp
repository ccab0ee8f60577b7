/*
 * flood_tables_impl.hpp -- instantiates every streaming flooding kernel of ONE arithmetic
 * (algorithm + message type) for V = 1, 2, 4 and hands out the function pointers.  Included by
 * flood_sp.hip / flood_ms.hip / flood_ms16.hip only.
 */
#pragma once

#include "flood_tables.hpp"

namespace ldpc {
namespace tables {

/* c: check kernels moving 1 float per lane (narrow waves), cw: V floats per lane */
template <int ALGO, int V, typename T, int D> struct FloodTable {
    static void fill(CheckFn *c, CheckFn *cw, VarFn *v)
    {
        c[D] = check_kernel<ALGO, D, V, 1, T>;
        cw[D] = check_kernel<ALGO, D, V, V, T>;
        v[D] = var_kernel<ALGO, D, V, T>;
        FloodTable<ALGO, V, T, D - 1>::fill(c, cw, v);
    }
};
template <int ALGO, int V, typename T> struct FloodTable<ALGO, V, T, 0> {
    static void fill(CheckFn *c, CheckFn *cw, VarFn *v)
    {
        c[0] = cw[0] = check_kernel_generic<ALGO, V, T>;
        v[0] = var_kernel_generic<ALGO, V, T>;
    }
};

/* min-sum rows of degree 17..32: narrow unrolled kernels only */
template <int V, typename T, int D> struct CheckTableMS {
    static void fill(CheckFn *c, CheckFn *cw)
    {
        c[D] = cw[D] = check_kernel<kAlgoMS, D, V, 1, T>;
        CheckTableMS<V, T, D - 1>::fill(c, cw);
    }
};
template <int V, typename T> struct CheckTableMS<V, T, kMaxUnrolledDegree> {
    static void fill(CheckFn *, CheckFn *) {}
};

template <int ALGO, int V, typename T, int D> struct LinkHalf {      /* 2 values per lane: V = 4 only */
    static LinkFn get() { return nullptr; }
};
template <int ALGO, typename T, int D> struct LinkHalf<ALGO, 4, T, D> {
    static LinkFn get() { return check_link_narrow_kernel<ALGO, D, 4, T, 2>; }
};
template <int ALGO, int V, typename T, int D> struct LinkTable {
    static void fill(LinkFn *t, LinkFn *tn, LinkFn *td, LinkFn *th)
    {
        t[D] = check_link_kernel<ALGO, D, V, T>;
        tn[D] = check_link_narrow_kernel<ALGO, D, V, T>;
        td[D] = check_link_narrow2_kernel<ALGO, D, V, T>;
        th[D] = LinkHalf<ALGO, V, T, D>::get();
        LinkTable<ALGO, V, T, D - 1>::fill(t, tn, td, th);
    }
};
template <int ALGO, int V, typename T> struct LinkTable<ALGO, V, T, 1> {
    static void fill(LinkFn *, LinkFn *, LinkFn *, LinkFn *) {}
};

template <int ALGO, int V, typename T> void fill_v(FloodFns *f)
{
    constexpr int DM = kMaxUnrolledDegree;
    FloodTable<ALGO, V, T, DM>::fill(f->check, f->check_wide, f->var);
    LinkTable<ALGO, V, T, DM>::fill(f->link, f->link_narrow, f->link_deep, f->link_half);
    /* fp16 messages in tiles of 256 frames: two values per lane (flood_kernels.hpp: check_group_kernel) */
    constexpr int GW = (sizeof(T) == 2 && V == 4) ? 2 : 1;
    f->check_group_width = GW;
    f->check_group[0] = check_group_kernel<ALGO, V, T, 1, 8, GW>;
    f->check_group[1] = check_group_kernel<ALGO, V, T, 9, 16, GW>;
    f->check_group[2] = f->check_group[3] = nullptr;
    f->var_group[0] = var_group_kernel<ALGO, V, T, 1, 4>;
    f->var_group[1] = var_group_kernel<ALGO, V, T, 5, 8>;
    f->var_group[2] = var_group_kernel<ALGO, V, T, 9, 16>;
    f->init = init_kernel<ALGO, V, T>;
    f->max_check_unrolled = DM;
    if (ALGO == kAlgoMS) {       /* min-sum rows of degree 17..32: narrow unrolled kernels */
        CheckTableMS<V, T, kMaxUnrolledCheckDegreeMS>::fill(f->check, f->check_wide);
        f->check_group[2] = check_group_kernel<kAlgoMS, V, T, 17, 24, GW>;
        f->check_group[3] = check_group_kernel<kAlgoMS, V, T, 25, 32, GW>;
        f->max_check_unrolled = kMaxUnrolledCheckDegreeMS;
    }
}

template <int ALGO, typename T> void fill(int V, FloodFns *f)
{
    *f = FloodFns();
    if (V == 1) fill_v<ALGO, 1, T>(f);
    else if (V == 2) fill_v<ALGO, 2, T>(f);
    else fill_v<ALGO, 4, T>(f);
}

}  // namespace tables
}  // namespace ldpc
