// phx_solver.hpp -- types and solver helpers shared by every translation unit of the engine (device + host).
// Reference citations are on the individual helpers.
#pragma once
#include <hip/hip_runtime.h>

#include "../../include/phoenix_hip.h"
#include "phx_device.hpp"

using namespace phx;

namespace phxt {   // plain types that cross translation units (kernel signatures, host entry points)

struct Net {
    const float *Ws, *bs, *Wp, *bp, *WaT, *g;
    int N, H;
    const float *wimg;   // packed LDS weight images of these values (phx_params.wimg) or null; set by the batch launchers only
};

struct SolveCfg {
    int method, control, t_per_sample, t_is_f32;   // t_is_f32: 1 = the caller's grid was fp32, 2 = ... and `t` still is
                                                   // (float data, read as double on the fly: no conversion kernel)
    float rtol, atol;  // the reference multiplies fp32 tensors by these (cast to fp32)
    long long max_steps;
};

// launch geometry (D1) and workspace pointers (W1) of the v1 / v2 / v3 MFMA kernels (phx_mfma_common.inc)
struct D1 {
    int N, H, B, T;
    int HT;    // hidden tiles of 16 rows (H <= 16*HT)
    int NB;    // gene blocks per workgroup
    int NW;    // waves per workgroup
    int TPW;   // trajectory tiles per wave
    int G;     // gene tiles  (workgroups per batch group)
    int TG;    // batch groups
    int nblk;  // gene blocks in total
    int ntg;   // trajectory tiles per group = NW*TPW
    int Bt;    // trajectories per group = 16*ntg
    int nvec;  // private state vectors per workgroup
    long long BN;
    int HC;    // hidden-row chunks (1: all weights of the gene tile stay LDS resident; >1: H > 128, the chunks of
    int Hc;    // Hc rows are re-staged per evaluation: f = sum over chunks, see DESIGN.md "wide hidden layers")
    int Bcall; // > 0: the batch is `TG` independent odeint calls of Bcall trajectories (shared control per call), batch
               // group g = call g, rows [g*Bcall, (g+1)*Bcall) of the caller's arrays;  0: groups are Bt consecutive rows
    long long cntN;   // elements under one shared-control norm (rows of one call x N)
    int res;   // chunked third-generation kernels (k1_solve_fwd3c / adj3c): 1 = the images of ALL hidden chunks of the
               // workgroup's gene blocks stay LDS resident, 0 = one chunk slot, re-staged as the sweeps walk the chunks
    int hb;    // chunked kernels: 1 = HALF-BLOCK gene tiles (G = 2 * nblk workgroups per batch group, 16 genes each)
    int split; // chunked kernels: > 1 = the waves of a trajectory tile take its gene blocks (block `wave / ntg`), partial rows added in LDS
};

struct W1 {
    unsigned long long *cnt;   // [TG] monotonically increasing group counters (zeroed per launch)
    unsigned int *abort_flag;
    unsigned long long *part;  // [TG][G][R][64] granules {tag<<32 | f32}: per-workgroup partial rows
    unsigned long long *zbuf;  // [slots][TG][R][64] granules: reduced rows (hidden vector, then norm rows)
    float *scratch;            // [TG*G][nvec][ntg*NB][64][8]     workgroup-private state tiles
    float *dtheta;             // [TG][PP] parameter-gradient partials (adjoint)
    unsigned long long *prof;  // [TG*G][16] optional per-workgroup segment timers (PHX_PROF=1), else null
    const float *wimg;         // chunked hidden layer: pre-packed LDS images [gene block][chunk][blk_floats_ch]
    float *hq;                 // k1_solve_adj2: wave-private transposed hidden rows [WG][8][TPW][7][2HT][64] float4
    unsigned long long *part1, *zbuf1;   // third-generation kernels: the second set of exchange buffers (xset_begin)
    int prof_level;            // k1_solve_fwd3: PHX_PROF level (2: per-block timers inside the sweeps)
};

}  // namespace phxt
using namespace phxt;

namespace {

inline size_t align_up(size_t x, size_t a) { return (x + a - 1) / a * a; }

// one trajectory's row of the time grid; the buffer holds doubles, or floats when cfg.t_is_f32 == 2
struct TimeRow {
    const void *p;
    bool f32;
    __device__ __forceinline__ double operator[](int i) const
    {
        return f32 ? (double)static_cast<const float *>(p)[i] : static_cast<const double *>(p)[i];
    }
};
__device__ __forceinline__ TimeRow trowT(const double *t, int T, const SolveCfg &cfg, int b)
{
    const bool f32 = cfg.t_is_f32 == 2;
    const long long off = cfg.t_per_sample ? (long long)b * T : 0;
    return TimeRow{f32 ? static_cast<const void *>(reinterpret_cast<const float *>(t) + off)
                       : static_cast<const void *>(t + off), f32};
}

__device__ __forceinline__ int fixed_nstages(int method)
{
    return method == PHX_EULER ? 1 : (method == PHX_MIDPOINT ? 2 : 4);
}

// quadrature weight of stage st for the parameter gradient (same combination as fixed_final)
__device__ __forceinline__ float fixed_weight(int method, int st, float dt)
{
    if (method == PHX_EULER) return dt;
    if (method == PHX_MIDPOINT) return st == 1 ? dt : 0.f;
    return ((st == 0 || st == 3) ? 1.0f : 3.0f) * dt * 0.125f;
}

// quartic dense output (interp.py:1-47) at fraction x of the step
struct InterpX { float x1, x2, x3, x4; };
__device__ __forceinline__ InterpX make_interp_x(double x)
{
    InterpX r;
    double xp = x;
    r.x1 = (float)x;
    xp *= x; r.x2 = (float)xp;
    xp *= x; r.x3 = (float)xp;
    xp *= x; r.x4 = (float)xp;
    return r;
}
__device__ __forceinline__ float interp_eval(float y0, float y1, float ym, float f0, float f1, float dt,
                                             const InterpX &ix)
{
    const float a = ((2.0f * dt) * (f1 - f0) - 8.0f * (y1 + y0)) + 16.0f * ym;
    const float bb = ((dt * (5.0f * f0 - 3.0f * f1) + 18.0f * y0) + 14.0f * y1) - 32.0f * ym;
    const float cc = ((dt * (f1 - 4.0f * f0) - 11.0f * y0) - 5.0f * y1) + 16.0f * ym;
    const float dd = dt * f0;
    float total = y0 + ix.x1 * dd;
    total = total + ix.x2 * cc;
    total = total + ix.x3 * bb;
    total = total + ix.x4 * a;
    return total;
}

// _select_initial_step, first half (misc.py:64-72)
__device__ __forceinline__ float init_h0(float d0, float d1)
{
    if (d0 < 1e-5f || d1 < 1e-5f) return 1e-6f;
    return (0.01f * d0) / d1;
}

// second half (misc.py:77-86), order + 1 = 5
__device__ __forceinline__ double init_dt(float h0, float d1, float d2)
{
    float h1;
    if (d1 <= 1e-15f && d2 <= 1e-15f) h1 = tmaxf(1e-6f, h0 * 1e-3f);
    else h1 = powf(0.01f / tmaxf(d1, d2), (float)(1.0 / 5.0));
    return (double)tminf(100.0f * h0, h1);
}

}  // namespace
