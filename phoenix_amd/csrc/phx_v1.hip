// phx_v1.hip -- the first-generation persistent solve kernels (k1_solve_fwd / k1_solve_adj: C5's H = 200 in two hidden
// chunks, the fixed-grid methods, H > 48 forward solves) as a translation unit of their own: instantiated here, launched
// from phx_engine.hip (extern templates there), so that this unit can take its own compiler flags (phoenix_amd/build.py:
// UNIT_FLAGS -- the iterative-ILP machine scheduler, which crashes clang on phx_engine.hip as a whole).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "phx_solver.hpp"
#include "phx_host.hpp"

#include "phx_mfma_common.inc"
#include "phx_mfma_fwd.inc"
#include "phx_mfma_adj.inc"

#define PHX_V1_FWD(HT, MAXT, CH)                                                                                        \
    template __global__ void phxk::k1_solve_fwd<HT, MAXT, CH>(Net, D1, W1, SolveCfg, const float *, const double *, float *, \
                                                             int *, int *, int *)
#define PHX_V1_ADJ(HT, MAXT, CH)                                                                                        \
    template __global__ void phxk::k1_solve_adj<HT, MAXT, CH>(Net, D1, W1, SolveCfg, const double *, const float *,       \
                                                             const float *, float *, int *, int *, int *, int, long long)
PHX_V1_FWD(7, 256, true);
PHX_V1_FWD(8, 256, true);
PHX_V1_FWD(3, 512, false);
PHX_V1_FWD(3, 256, false);
PHX_V1_FWD(8, 256, false);
PHX_V1_ADJ(7, 256, true);
PHX_V1_ADJ(8, 256, true);
PHX_V1_ADJ(3, 256, false);
PHX_V1_ADJ(8, 256, false);
