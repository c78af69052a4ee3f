// phx_device.hpp -- device-side building blocks of the PHOENIX NeuralODE engine (gfx950).
//
// Data model (DESIGN.md "data layout"):
//   * the expression state and every RK stage live as [B, N] fp32 rows (gene-contiguous);
//   * all weight matrices are gene-contiguous [rows, N]: Ws [H,N], Wp [H,N], WaT [2H,N];
//   * a work ITEM is (trajectory b, gene chunk c) with CH = 512 genes = one lane per gene;
//     items are statically assigned to workgroups, so every elementwise quantity of an item is
//     produced and consumed by the same lane (no synchronisation needed for it);
//   * the only cross-workgroup data of one RHS evaluation is the hidden vector
//     (per-item partial sums -> [B, 2H] / [B, 4H]), exchanged through a grid barrier.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace phx {

constexpr int NT = 512;          // threads per workgroup
constexpr int NWAVE = NT / 64;   // wavefronts per workgroup (wave64)
constexpr int CH = NT;           // genes per item
constexpr int RT = 8;            // register tile (rows) of the parameter-gradient pass
constexpr int MAXS = 7;          // max RK stages kept (dopri5: k1..k7)

// ----------------------------------------------------------------------------------------
// grid-wide synchronisation block (zeroed by a memset node before every launch)
// ----------------------------------------------------------------------------------------
struct SyncBlock {
    unsigned long long counter;   // monotonically increasing arrival counter
    unsigned int abort_flag;      // set when a barrier timed out -> everyone bails out
    unsigned int remaining;       // trajectories (controllers) not yet finished
    unsigned int pad[12];
};
static_assert(sizeof(SyncBlock) == 64, "SyncBlock must be 64 bytes");

struct GridSync {
    SyncBlock *blk;
    unsigned int nwg;
    unsigned long long epoch;     // per-workgroup private copy (uniform by construction)
    bool aborted;
};

// Counter barrier following MI355X_MICROARCH.md "Valid forms": every storing wave drains its
// stores, workgroup barrier, ONE lane issues the agent-scope release (+ explicit vmcnt(0), the
// ROCm 7.2 compiler hazard), relaxed agent atomic arrive, relaxed polling with s_sleep, ONE
// agent-scope acquire, vmcnt(0), workgroup barrier.  Every spin is bounded (5 s wall clock).
__device__ __forceinline__ void grid_barrier(GridSync &gs, unsigned int *lds_flag)
{
    asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
    __syncthreads();
    if (gs.nwg > 1 && !gs.aborted) {
        if (threadIdx.x == 0) {
            __builtin_amdgcn_fence(__ATOMIC_RELEASE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            const unsigned long long target = (gs.epoch + 1ull) * (unsigned long long)gs.nwg;
            __hip_atomic_fetch_add(&gs.blk->counter, 1ull, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            unsigned int bad = 0;
            const unsigned long long t_start = wall_clock64();   // 100 MHz constant clock
            unsigned int spins = 0;
            while (__hip_atomic_load(&gs.blk->counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < target) {
                __builtin_amdgcn_s_sleep(1);
                if ((++spins & 1023u) == 0) {
                    if (__hip_atomic_load(&gs.blk->abort_flag, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT)) { bad = 1; break; }
                    if (wall_clock64() - t_start > 500000000ull) {   // 5 s
                        __hip_atomic_store(&gs.blk->abort_flag, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
                        bad = 1;
                        break;
                    }
                }
            }
            __builtin_amdgcn_fence(__ATOMIC_ACQUIRE, "agent");
            asm volatile("s_waitcnt vmcnt(0)" ::: "memory");
            *lds_flag = bad;
        }
        __syncthreads();
        if (*lds_flag) gs.aborted = true;
        __syncthreads();
    }
    gs.epoch += 1ull;
}

// ----------------------------------------------------------------------------------------
// small math helpers
// ----------------------------------------------------------------------------------------
// SoftsignMod / LogShiftedSoftSignMod (reference odenet.py:21-25, 31-35), as the reference computes them
__device__ __forceinline__ void act_pair(float y, float &a, float &l)
{
    const float s = y - 0.5f;
    const float d = 1.0f + fabsf(s);
    a = s / d;
    l = log1pf(a);
}
// Same two activations with hardware reciprocal/log (v_rcp_f32 + one Newton step; v_log_f32 away from
// a = 0, alternating series near it): ~2e-7 relative, a third of the instructions of the libm forms.
__device__ __forceinline__ float fast_rcp(float d)
{
    float r = __builtin_amdgcn_rcpf(d);
    return fmaf(fmaf(-d, r, 1.0f), r, r);
}
__device__ __forceinline__ float fast_log1p(float a)
{
    // |a| < 0.25: a * sum_{m=0..12} (-a)^m / (m+1)   (truncation < 2e-9)
    const float x = -a;
    float sm = 1.0f / 13.0f;
    sm = fmaf(sm, x, 1.0f / 12.0f); sm = fmaf(sm, x, 1.0f / 11.0f); sm = fmaf(sm, x, 1.0f / 10.0f);
    sm = fmaf(sm, x, 1.0f / 9.0f);  sm = fmaf(sm, x, 1.0f / 8.0f);  sm = fmaf(sm, x, 1.0f / 7.0f);
    sm = fmaf(sm, x, 1.0f / 6.0f);  sm = fmaf(sm, x, 1.0f / 5.0f);  sm = fmaf(sm, x, 1.0f / 4.0f);
    sm = fmaf(sm, x, 1.0f / 3.0f);  sm = fmaf(sm, x, 0.5f);         sm = fmaf(sm, x, 1.0f);
    const float small = a * sm;
    const float big = __builtin_amdgcn_logf(1.0f + a) * 0.69314718055994531f;   // v_log_f32 = log2
    return (fabsf(a) < 0.25f) ? small : big;
}
__device__ __forceinline__ void act_pair_fast(float y, float &a, float &l)
{
    const float s = y - 0.5f;
    const float d = 1.0f + fabsf(s);
    a = s * fast_rcp(d);
    l = fast_log1p(a);
}

// act_pair_fast on 2 NP values, written on element PAIRS so that the subtractions, the Newton step of the reciprocal
// and the 13-term log1p series issue as v_pk_add / v_pk_mul / v_pk_fma_f32 (one instruction per pair; the same operations
// in the same order per element, and v_pk_fma_f32 rounds like v_fma_f32 on this part: tools/membench/pk_fma_bits.hip).
// Only |s|, the two transcendentals and the final select stay per element: ~15 instead of ~26 vector instructions per value.
typedef float f32x2 __attribute__((ext_vector_type(2)));
__device__ __forceinline__ f32x2 pk_fma(f32x2 a, f32x2 b, f32x2 c) { return __builtin_elementwise_fma(a, b, c); }
template <int NP>
__device__ __forceinline__ void act_pair_fast_pk(const float *y, float *a, float *l)
{
#pragma unroll
    for (int p = 0; p < NP; ++p) {
        const f32x2 yv = {y[2 * p], y[2 * p + 1]};
        const f32x2 s = yv - (f32x2){0.5f, 0.5f};
        const f32x2 d = (f32x2){1.0f, 1.0f} + (f32x2){fabsf(s.x), fabsf(s.y)};
        f32x2 r = {__builtin_amdgcn_rcpf(d.x), __builtin_amdgcn_rcpf(d.y)};
        r = pk_fma(pk_fma(-d, r, (f32x2){1.0f, 1.0f}), r, r);                 // fast_rcp: one Newton step
        const f32x2 av = s * r;
        const f32x2 x = -av;
        f32x2 sm = {1.0f / 13.0f, 1.0f / 13.0f};
#define PHX_H(c) sm = pk_fma(sm, x, (f32x2){c, c})
        PHX_H(1.0f / 12.0f); PHX_H(1.0f / 11.0f); PHX_H(1.0f / 10.0f); PHX_H(1.0f / 9.0f); PHX_H(1.0f / 8.0f);
        PHX_H(1.0f / 7.0f);  PHX_H(1.0f / 6.0f);  PHX_H(1.0f / 5.0f);  PHX_H(1.0f / 4.0f); PHX_H(1.0f / 3.0f);
        PHX_H(0.5f);         PHX_H(1.0f);
#undef PHX_H
        const f32x2 small = av * sm;
        const f32x2 one_a = (f32x2){1.0f, 1.0f} + av;
        const f32x2 big = (f32x2){__builtin_amdgcn_logf(one_a.x), __builtin_amdgcn_logf(one_a.y)} *
                          (f32x2){0.69314718055994531f, 0.69314718055994531f};
        a[2 * p] = av.x; a[2 * p + 1] = av.y;
        l[2 * p] = (fabsf(av.x) < 0.25f) ? small.x : big.x;
        l[2 * p + 1] = (fabsf(av.y) < 0.25f) ? small.y : big.y;
    }
}
__device__ __forceinline__ void act_pair_fast8(const float (&y)[8], float (&a)[8], float (&l)[8]) { act_pair_fast_pk<4>(y, a, l); }

// closed-form derivatives
__device__ __forceinline__ void act_grad(float y, float &da, float &dl)
{
    const float s = y - 0.5f;
    const float d = 1.0f + fabsf(s);
    da = 1.0f / (d * d);
    dl = (s < 0.0f) ? 1.0f / d : 1.0f / ((1.0f + s) * (1.0f + 2.0f * s));
}

__device__ __forceinline__ float wave_sum(float v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_xor(v, off, 64);
    return v;
}

// sum over the whole workgroup; result valid in every thread. `scr` = NWAVE floats of LDS.
__device__ __forceinline__ float block_sum(float v, float *scr)
{
    v = wave_sum(v);
    __syncthreads();
    if ((threadIdx.x & 63) == 0) scr[threadIdx.x >> 6] = v;
    __syncthreads();
    float t = 0.0f;
#pragma unroll
    for (int w = 0; w < NWAVE; ++w) t += scr[w];
    return t;
}

// torch.min / torch.max propagate NaN
__device__ __forceinline__ double tmin(double a, double b) { return (a != a || a < b) ? a : b; }
__device__ __forceinline__ double tmax(double a, double b) { return (a != a || a > b) ? a : b; }
__device__ __forceinline__ float tminf(float a, float b) { return (a != a || a < b) ? a : b; }
__device__ __forceinline__ float tmaxf(float a, float b) { return (a != a || a > b) ? a : b; }

// ----------------------------------------------------------------------------------------
// Dormand-Prince(-Shampine) tableau, cast to fp32 as the reference does (dopri5.py:5-30,
// rk_common.py:134-138)
// ----------------------------------------------------------------------------------------
__device__ const float DP_BETA[6][6] = {
    {(float)(1.0 / 5), 0, 0, 0, 0, 0},
    {(float)(3.0 / 40), (float)(9.0 / 40), 0, 0, 0, 0},
    {(float)(44.0 / 45), (float)(-56.0 / 15), (float)(32.0 / 9), 0, 0, 0},
    {(float)(19372.0 / 6561), (float)(-25360.0 / 2187), (float)(64448.0 / 6561), (float)(-212.0 / 729), 0, 0},
    {(float)(9017.0 / 3168), (float)(-355.0 / 33), (float)(46732.0 / 5247), (float)(49.0 / 176), (float)(-5103.0 / 18656), 0},
    {(float)(35.0 / 384), 0.0f, (float)(500.0 / 1113), (float)(125.0 / 192), (float)(-2187.0 / 6784), (float)(11.0 / 84)},
};
__device__ const float DP_CERR[7] = {
    (float)(35.0 / 384 - 1951.0 / 21600), 0.0f, (float)(500.0 / 1113 - 22642.0 / 50085),
    (float)(125.0 / 192 - 451.0 / 720), (float)(-2187.0 / 6784 - -12231.0 / 42400),
    (float)(11.0 / 84 - 649.0 / 6300), (float)(-1.0 / 60.0)};
__device__ const float DP_CMID[7] = {
    (float)(6025192743.0 / 30085553152.0 / 2), 0.0f, (float)(51252292925.0 / 65400821598.0 / 2),
    (float)(-2691868925.0 / 45128329728.0 / 2), (float)(187940372067.0 / 1594534317056.0 / 2),
    (float)(-1776094331.0 / 19743644256.0 / 2), (float)(11237099.0 / 235043384.0 / 2)};
// fp64 copies for the dense-output quadrature weights of the parameter gradient
__device__ const double DP_CSOL64[7] = {35.0 / 384, 0.0, 500.0 / 1113, 125.0 / 192, -2187.0 / 6784, 11.0 / 84, 0.0};
__device__ const double DP_CMID64[7] = {
    6025192743.0 / 30085553152.0 / 2, 0.0, 51252292925.0 / 65400821598.0 / 2,
    -2691868925.0 / 45128329728.0 / 2, 187940372067.0 / 1594534317056.0 / 2,
    -1776094331.0 / 19743644256.0 / 2, 11237099.0 / 235043384.0 / 2};

// _optimal_step_size (reference misc.py:94-103)
__device__ __forceinline__ double optimal_step_size(double last_step, float error_ratio)
{
    const double safety = 0.9, ifactor = 10.0;
    double dfactor = 0.2;
    if (error_ratio == 0.0f) return last_step * ifactor;
    if (error_ratio < 1.0f) dfactor = 1.0;
    const double er = (double)error_ratio;
    const double factor = tmin(ifactor, tmax(safety / pow(er, 0.2), dfactor));
    return last_step * factor;
}

}  // namespace phx
