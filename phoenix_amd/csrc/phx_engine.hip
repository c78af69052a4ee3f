// phx_engine.hip -- MI355X (gfx950) kernels + C ABI of the PHOENIX NeuralODE engine.
//
// What it replaces in the reference (paths relative to /root/reference/ode_net/code/):
//   ODENet.forward / prior_only_forward                      odenet.py:85-98
//   torch.autograd.grad through it (RHS VJP)                  torchdiffeq/_impl/adjoint.py:101-119
//   odeint: fixed grid (euler, midpoint, rk4 = 3/8 rule)      solvers.py:77-95, fixed_grid.py:6-38, rk_common.py:96-103
//   odeint: dopri5 adaptive + dense output                    rk_common.py:39-77,140-228, misc.py:47-103, interp.py
//   OdeintAdjointMethod.backward                              adjoint.py:32-162
//
// Execution model: ONE persistent launch per solve.  The launch is a cooperative grid of at most
// one 512-thread workgroup per CU; workgroups own (trajectory, 512-gene chunk) items for the
// whole solve and meet at grid barriers only to exchange the hidden vector ([B,2H] forward,
// [B,4H] augmented) and the controller scalars.  All RK stages, the error norm, accept/reject,
// step-size control, dense output and (backward) the parameter-gradient quadrature run inside
// that one launch -- no host round trip per step.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <atomic>
#include <cstdint>
#include <cstdio>
#include <cstring>
#include <cstdlib>

#include "phx_solver.hpp"
#include "phx_host.hpp"

using namespace phxh;

namespace {


struct Dims {
    int N, H, B, T;
    int NC;        // gene chunks per trajectory
    int items;     // B * NC
    int GB;        // trajectory groups of the parameter-gradient pass
    int BG;        // trajectories per group
    int Bc;        // number of step controllers (1 = shared, B = per trajectory)
    long long BN;  // B * N
    long long PP;  // padded size of one parameter-gradient partial
};


// workspace views (device pointers)
struct WS {
    SyncBlock *sync;
    double *rk_t0, *rk_t1, *dt, *xfin;
    float *h0f, *dtf, *dtp, *sgn, *wq;  // dtp: fp32 dt of the step just taken; wq: [B][8] quadrature weights
    int *out_idx, *done, *accept, *out_lo, *out_hi, *fin, *st, *nsteps, *nfe;
    int *nsiv;                    // steps taken towards the current output time (the reference's per-_advance budget)
    float *Y0, *A0, *YS, *AS;     // [B,N] each
    float *KY, *KA;               // [S][B,N]
    float *SA, *SL, *SQ, *SG;     // [S][B,N] stage data for the parameter-gradient pass
    float *partial;               // [items][4H]
    float *red;                   // [items][4]
    float *Z, *DZ;                // [S][B][2H]
    float *dtheta;                // [GB][PP]
};

struct Layout {
    size_t total;
    size_t sync, rk_t0, rk_t1, dt, xfin, h0f, dtf, dtp, sgn, wq, ints, Y0, A0, YS, AS, KY, KA, SA, SL, SQ, SG,
        partial, red, Z, DZ, dtheta;
};


inline Dims make_dims(int N, int H, int B, int T, int control)
{
    Dims d;
    d.N = N; d.H = H; d.B = B; d.T = T;
    d.NC = (N + CH - 1) / CH;
    d.items = B * d.NC;
    d.GB = std::min(B, 8);
    d.BG = (B + d.GB - 1) / d.GB;
    d.Bc = (control == PHX_CTRL_SHARED) ? 1 : B;
    d.BN = (long long)B * N;
    d.PP = (long long)align_up((size_t)4 * H * N + N + 2 * H, 4);
    return d;
}

// nstate: number of [B,N] state vectors (Y0,YS / +A0,AS); nk: number of K stage slots kept for
// y (and for adj when aug); nsd: number of stage-data slots (SA,SL,SQ,SG) ; grads: dtheta partials
inline Layout make_layout(const Dims &d, int op)
{
    const bool solve = (op == PHX_OP_ODEINT || op == PHX_OP_ADJOINT);
    const bool aug = (op == PHX_OP_RHS_VJP || op == PHX_OP_ADJOINT);
    const int nk = solve ? MAXS : 0;
    const int nsd = (op == PHX_OP_ADJOINT) ? MAXS : (op == PHX_OP_RHS_VJP ? 1 : 0);
    const int nz = solve ? MAXS : 1;
    Layout L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t B = d.B, BN = (size_t)d.BN;
    L.sync = take(sizeof(SyncBlock));
    L.rk_t0 = take(8 * B); L.rk_t1 = take(8 * B); L.dt = take(8 * B); L.xfin = take(8 * B);
    L.h0f = take(4 * B); L.dtf = take(4 * B); L.dtp = take(4 * B); L.sgn = take(4 * B); L.wq = take(4 * B * 8);
    L.ints = take(4 * B * 10);
    L.Y0 = take(solve ? 4 * BN : 0);
    L.YS = take(solve ? 4 * BN : 0);
    L.A0 = take(op == PHX_OP_ADJOINT ? 4 * BN : 0);
    L.AS = take(op == PHX_OP_ADJOINT ? 4 * BN : 0);
    L.KY = take(4 * BN * nk);
    L.KA = take(op == PHX_OP_ADJOINT ? 4 * BN * nk : 0);
    L.SA = take(4 * BN * nsd); L.SL = take(4 * BN * nsd); L.SQ = take(4 * BN * nsd); L.SG = take(4 * BN * nsd);
    L.partial = take(4 * (size_t)d.items * (aug ? 4 : 2) * d.H);
    L.red = take(4 * (size_t)d.items * 4);
    L.Z = take(4 * B * 2 * d.H * nz);
    L.DZ = take(aug ? 4 * B * 2 * d.H * nz : 0);
    L.dtheta = take(aug ? 4 * (size_t)d.PP * d.GB : 0);
    L.total = off;
    return L;
}

inline WS make_ws(void *base, const Layout &L, const Dims &d)
{
    char *p = (char *)base;
    WS w;
    w.sync = (SyncBlock *)(p + L.sync);
    w.rk_t0 = (double *)(p + L.rk_t0); w.rk_t1 = (double *)(p + L.rk_t1);
    w.dt = (double *)(p + L.dt); w.xfin = (double *)(p + L.xfin);
    w.h0f = (float *)(p + L.h0f); w.dtf = (float *)(p + L.dtf); w.dtp = (float *)(p + L.dtp);
    w.sgn = (float *)(p + L.sgn);
    w.wq = (float *)(p + L.wq);
    int *ib = (int *)(p + L.ints);
    const size_t B = d.B;
    w.out_idx = ib; w.done = ib + B; w.accept = ib + 2 * B; w.out_lo = ib + 3 * B; w.out_hi = ib + 4 * B;
    w.fin = ib + 5 * B; w.st = ib + 6 * B; w.nsteps = ib + 7 * B; w.nfe = ib + 8 * B; w.nsiv = ib + 9 * B;
    w.Y0 = (float *)(p + L.Y0); w.A0 = (float *)(p + L.A0); w.YS = (float *)(p + L.YS); w.AS = (float *)(p + L.AS);
    w.KY = (float *)(p + L.KY); w.KA = (float *)(p + L.KA);
    w.SA = (float *)(p + L.SA); w.SL = (float *)(p + L.SL); w.SQ = (float *)(p + L.SQ); w.SG = (float *)(p + L.SG);
    w.partial = (float *)(p + L.partial); w.red = (float *)(p + L.red);
    w.Z = (float *)(p + L.Z); w.DZ = (float *)(p + L.DZ); w.dtheta = (float *)(p + L.dtheta);
    return w;
}

// ========================================================================================
// device phases of one RHS evaluation
// ========================================================================================
struct Lds {
    float xa[CH], xl[CH], xq[CH];  // activations of the item's genes (phase A) / z, du, dv (phase C)
    float red[NWAVE];
    unsigned int flag;
    unsigned int remaining;
};

// Phase A of item (b, c): per-gene activations, then the item's partial hidden sums
//   rows [0,H): sum_n a[n] Ws[r][n];  [H,2H): sum_n l[n] Wp[r-H][n];  (AUG) [2H,4H): sum_n q[n] WaT[r-2H][n]
template <bool AUG>
__device__ __forceinline__ void phaseA(const Net &net, const Dims &d, const WS &w, int item, int c, float y,
                                       float cot, bool valid, bool prior_only, int sd_slot, long long e, Lds &s)
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const int N = net.N, H = net.H;
    const int n = c * CH + tid;
    float a = 0.f, l = 0.f, q = 0.f;
    if (valid) {
        act_pair(y, a, l);
        if (AUG) {
            const float gm = net.g[n];
            const float r = prior_only ? 1.0f : (gm > 0.f ? gm : 0.f);
            q = cot * r;
        }
    }
    __syncthreads();  // previous users of the LDS arrays are done
    s.xa[tid] = a;
    s.xl[tid] = l;
    if (AUG) s.xq[tid] = q;
    if (AUG && sd_slot >= 0 && valid) {
        const long long o = (long long)sd_slot * d.BN + e;
        w.SA[o] = a; w.SL[o] = l; w.SQ[o] = q;
    }
    __syncthreads();
    const int R = AUG ? 4 * H : 2 * H;
    float *pout = w.partial + (long long)item * R;
    const int n0 = c * CH;
    for (int r = wv; r < R; r += NWAVE) {
        const float *x = (r < H) ? s.xa : ((r < 2 * H) ? s.xl : s.xq);
        const float *Wrow = (r < H) ? net.Ws + (long long)r * N
                          : (r < 2 * H) ? net.Wp + (long long)(r - H) * N
                                        : net.WaT + (long long)(r - 2 * H) * N;
        float acc = 0.f;
#pragma unroll
        for (int i = 0; i < CH / 64; ++i) {
            const int nn = n0 + i * 64 + lane;
            if (nn < N) acc += x[i * 64 + lane] * Wrow[nn];
        }
        acc = wave_sum(acc);
        if (lane == 0) pout[r] = acc;
    }
}

// Phase B: reduce the per-item partials of trajectory b in fixed chunk order -> hidden vector
//   z[k<H] = u + bs ; z[k>=H] = exp(v + bp) ; (AUG) du = dz[k<H], dv = dz[k>=H] * p
template <bool AUG>
__device__ __forceinline__ void phaseB(const Net &net, const Dims &d, const WS &w, int slot, const int *skip,
                                       bool shared_ctrl)
{
    const int H = net.H, K2 = 2 * H;
    const int R = AUG ? 4 * H : 2 * H;
    const long long total = (long long)d.B * K2;
    for (long long idx = (long long)blockIdx.x * NT + threadIdx.x; idx < total; idx += (long long)gridDim.x * NT) {
        const int b = (int)(idx / K2), k = (int)(idx % K2);
        if (skip && skip[shared_ctrl ? 0 : b]) continue;
        const float *pp = w.partial + (long long)b * d.NC * R;
        float sacc = 0.f;
        for (int c = 0; c < d.NC; ++c) sacc += pp[(long long)c * R + k];
        sacc += (k < H) ? net.bs[k] : net.bp[k - H];
        if (k >= H) sacc = expf(sacc);
        const long long zo = ((long long)slot * d.B + b) * K2 + k;
        w.Z[zo] = sacc;
        if (AUG) {
            float t = 0.f;
            for (int c = 0; c < d.NC; ++c) t += pp[(long long)c * R + K2 + k];
            if (k >= H) t *= sacc;
            w.DZ[zo] = t;
        }
    }
}

// Phase C of item (b, c): expansion back to genes.
//   j = sum_k z[k] WaT[k][n];  f = relu(g) (j - y)      (prior_only: f = j)
//   (AUG) vjp = -q + (sum_h du[h] Ws[h][n]) a'(y) + (sum_h dv[h] Wp[h][n]) l'(y);  sg = cot (j - y) [g>0]
template <bool AUG>
__device__ __forceinline__ void phaseC(const Net &net, const Dims &d, const WS &w, int b, int c, int slot, float y,
                                       float cot, bool valid, bool prior_only, float &f, float &vjp, float &sg,
                                       Lds &s)
{
    const int tid = threadIdx.x;
    const int N = net.N, H = net.H, K2 = 2 * H;
    const int n = c * CH + tid;
    __syncthreads();
    const long long zo = ((long long)slot * d.B + b) * K2;
    for (int k = tid; k < K2; k += NT) {
        s.xa[k] = w.Z[zo + k];
        if (AUG) s.xl[k] = w.DZ[zo + k];
    }
    __syncthreads();
    f = 0.f; vjp = 0.f; sg = 0.f;
    if (!valid) return;
    float j = 0.f;
    const float *wa = net.WaT + n;
    for (int k = 0; k < K2; ++k) j += s.xa[k] * wa[(long long)k * N];
    const float gm = net.g[n];
    const float r = prior_only ? 1.0f : (gm > 0.f ? gm : 0.f);
    f = prior_only ? j : r * (j - y);
    if (AUG) {
        float da_pre = 0.f, dl_pre = 0.f;
        const float *ws = net.Ws + n, *wp = net.Wp + n;
        for (int h = 0; h < H; ++h) {
            da_pre += s.xl[h] * ws[(long long)h * N];
            dl_pre += s.xl[H + h] * wp[(long long)h * N];
        }
        float da, dl;
        act_grad(y, da, dl);
        const float q = cot * r;
        vjp = da_pre * da + dl_pre * dl;
        if (!prior_only) {
            vjp -= q;
            sg = (gm > 0.f) ? cot * (j - y) : 0.f;
        }
    }
}

// Parameter-gradient pass: dtheta[bg] += sum_{b in group, stage s} wq(b,s) * vjp_theta(stage data).
// Items: (gene chunk c, row tile, trajectory group bg); one lane per gene, RT rows in registers.
//   Ws rows:  dWs[h][n]  += w du[h] a[n];   Wp rows: dWp[h][n] += w dv[h] l[n];
//   WaT rows: dWaT[k][n] += w z[k] q[n];    g: dg[n] += w sg[n];  biases: dbs += w du, dbp += w dv.
// wmode: 0 = weight 1, single stage (standalone VJP); 1 = wq array [B][8] (solver steps)
__device__ __forceinline__ void phaseG(const Net &net, const Dims &d, const WS &w, int S, int wmode)
{
    const int tid = threadIdx.x;
    const int N = net.N, H = net.H, K2 = 2 * H;
    const int ntS = (H + RT - 1) / RT, ntA = (K2 + RT - 1) / RT;
    const int NTILE = 2 * ntS + ntA + 1;  // Ws tiles, Wp tiles, WaT tiles, g tile
    const long long oWs = 0, oWp = (long long)H * N, oWa = 2LL * H * N, og = 4LL * H * N, obs = og + N,
                    obp = obs + H;
    const int gene_items = d.NC * NTILE * d.GB;
    const int total = gene_items + d.GB;  // + one bias item per group
    for (int it = blockIdx.x; it < total; it += gridDim.x) {
        if (it >= gene_items) {
            // bias item of group bg: thread k < 2H
            const int bg = it - gene_items;
            const int b0 = bg * d.BG, b1 = min(d.B, b0 + d.BG);
            if (tid < K2) {
                float acc = 0.f;
                for (int b = b0; b < b1; ++b)
                    for (int sidx = 0; sidx < S; ++sidx) {
                        const float wt = wmode ? w.wq[b * 8 + sidx] : 1.0f;
                        if (wt != 0.f) acc += wt * w.DZ[((long long)sidx * d.B + b) * K2 + tid];
                    }
                float *dst = w.dtheta + (long long)bg * d.PP + (tid < H ? obs + tid : obp + (tid - H));
                *dst += acc;
            }
            continue;
        }
        const int bg = it % d.GB;
        const int tile = (it / d.GB) % NTILE;
        const int c = it / (d.GB * NTILE);
        const int n = c * CH + tid;
        const bool valid = n < N;
        const int b0 = bg * d.BG, b1 = min(d.B, b0 + d.BG);
        // section of this tile
        int sec, row0, nrows;
        if (tile < ntS) { sec = 0; row0 = tile * RT; nrows = min(RT, H - row0); }
        else if (tile < 2 * ntS) { sec = 1; row0 = (tile - ntS) * RT; nrows = min(RT, H - row0); }
        else if (tile < 2 * ntS + ntA) { sec = 2; row0 = (tile - 2 * ntS) * RT; nrows = min(RT, K2 - row0); }
        else { sec = 3; row0 = 0; nrows = 1; }
        const float *X = (sec == 0) ? w.SA : (sec == 1) ? w.SL : (sec == 2) ? w.SQ : w.SG;
        float acc[RT];
#pragma unroll
        for (int jx = 0; jx < RT; ++jx) acc[jx] = 0.f;
        for (int b = b0; b < b1; ++b) {
            for (int sidx = 0; sidx < S; ++sidx) {
                const float wt = wmode ? w.wq[b * 8 + sidx] : 1.0f;
                if (wt == 0.f) continue;  // uniform
                const float x = valid ? X[(long long)sidx * d.BN + (long long)b * N + n] : 0.f;
                if (sec == 3) {
                    acc[0] += wt * x;
                } else {
                    const float *coef = ((sec == 2) ? w.Z : w.DZ) + ((long long)sidx * d.B + b) * K2 +
                                        (sec == 1 ? H : 0) + row0;
#pragma unroll
                    for (int jx = 0; jx < RT; ++jx)
                        if (jx < nrows) acc[jx] += (wt * coef[jx]) * x;
                }
            }
        }
        if (valid) {
            float *base = w.dtheta + (long long)bg * d.PP;
            if (sec == 3) {
                base[og + n] += acc[0];
            } else {
                const long long o = (sec == 0) ? oWs : (sec == 1) ? oWp : oWa;
#pragma unroll
                for (int jx = 0; jx < RT; ++jx)
                    if (jx < nrows) base[o + (long long)(row0 + jx) * N + n] += acc[jx];
            }
        }
    }
}

// ========================================================================================
// standalone RHS / VJP evaluation (ODENet.forward, prior_only_forward and their backward)
// ========================================================================================
template <bool AUG>
__global__ __launch_bounds__(NT) void k_eval(Net net, Dims d, WS w, const float *__restrict__ y,
                                             const float *__restrict__ cot, float *out_f, float *out_vjp,
                                             int prior_only, int want_grads, int *status)
{
    __shared__ Lds s;
    GridSync gs{w.sync, gridDim.x, 0ull, false};
    const int tid = threadIdx.x;
    for (int item = blockIdx.x; item < d.items; item += gridDim.x) {
        const int b = item / d.NC, c = item % d.NC;
        const int n = c * CH + tid;
        const bool valid = n < d.N;
        const long long e = (long long)b * d.N + n;
        const float yv = valid ? y[e] : 0.5f;
        const float cv = (AUG && valid) ? cot[e] : 0.f;
        phaseA<AUG>(net, d, w, item, c, yv, cv, valid, prior_only != 0, (AUG && want_grads) ? 0 : -1, e, s);
    }
    grid_barrier(gs, &s.flag);
    phaseB<AUG>(net, d, w, 0, nullptr, false);
    grid_barrier(gs, &s.flag);
    for (int item = blockIdx.x; item < d.items; item += gridDim.x) {
        const int b = item / d.NC, c = item % d.NC;
        const int n = c * CH + tid;
        const bool valid = n < d.N;
        const long long e = (long long)b * d.N + n;
        const float yv = valid ? y[e] : 0.5f;
        const float cv = (AUG && valid) ? cot[e] : 0.f;
        float f, vjp, sg;
        phaseC<AUG>(net, d, w, b, c, 0, yv, cv, valid, prior_only != 0, f, vjp, sg, s);
        if (valid) {
            if (out_f) out_f[e] = f;
            if (AUG && out_vjp) out_vjp[e] = vjp;
            if (AUG && want_grads) w.SG[e] = sg;
        }
    }
    if (AUG && want_grads) {
        grid_barrier(gs, &s.flag);
        phaseG(net, d, w, 1, 0);
    }
    if (blockIdx.x == 0 && tid == 0 && status) status[0] = gs.aborted ? PHX_ERR_SYNC_TIMEOUT : PHX_OK;
}

// ========================================================================================
// solver helpers
// ========================================================================================
__device__ __forceinline__ TimeRow trow(const double *t, const Dims &d, const SolveCfg &cfg, int b)
{
    return trowT(t, d.T, cfg, b);
}

// stage input of the fixed-grid methods, in the reference's own operation order
// (fixed_grid.py:13-27, rk_common.py:96-103)
__device__ __forceinline__ float fixed_stage_input(int method, int st, float y0, const float *K, long long BN,
                                                   long long e, float dt)
{
    const float third = (float)(1.0 / 3.0);
    if (st == 0) return y0;
    if (method == PHX_MIDPOINT) return y0 + K[e] * (0.5f * dt);
    // rk4 = 3/8 rule
    if (st == 1) return y0 + (dt * K[e]) * third;
    if (st == 2) return y0 + dt * (K[BN + e] - K[e] * third);
    return y0 + dt * ((K[e] - K[BN + e]) + K[2 * BN + e]);
}
__device__ __forceinline__ float fixed_final(int method, float y0, const float *K, long long BN, long long e,
                                             float dt)
{
    if (method == PHX_EULER) return y0 + dt * K[e];
    if (method == PHX_MIDPOINT) return y0 + dt * K[BN + e];
    return y0 + (((K[e] + 3.0f * (K[BN + e] + K[2 * BN + e])) + K[3 * BN + e]) * dt) * 0.125f;
}

// dopri5 stage input: y0 + k[:st] . (beta_st * dt)   (rk_common.py:64-66), fp32, j ascending
__device__ __forceinline__ float dp_stage_input(int st, float y0, const float *K, long long BN, long long e,
                                                float dt)
{
    float acc = 0.f;
    for (int j = 0; j < st; ++j) acc += K[(long long)j * BN + e] * (DP_BETA[st - 1][j] * dt);
    return y0 + acc;
}
__device__ __forceinline__ float dp_combo(const float *coef7, const float *K, long long BN, long long e, float dt)
{
    float acc = 0.f;
#pragma unroll
    for (int j = 0; j < 7; ++j) acc += K[(long long)j * BN + e] * (dt * coef7[j]);
    return acc;
}


// sum over the items of controller cb (fixed order, fp64) of red[..][slot]
__device__ __forceinline__ double ctrl_sum(const Dims &d, const WS &w, int cb, bool shared, int slot)
{
    double acc = 0.0;
    const int i0 = shared ? 0 : cb * d.NC, i1 = shared ? d.items : (cb + 1) * d.NC;
    for (int i = i0; i < i1; ++i) acc += (double)w.red[(long long)i * 4 + slot];
    return acc;
}

__device__ __forceinline__ float rms_from_sum(double sum, double count) { return sqrtf((float)(sum / count)); }


// ========================================================================================
// forward solve:  odeint(ODENet, y0, t)
// ========================================================================================
__global__ __launch_bounds__(NT) void k_solve_fwd(Net net, Dims d, WS w, SolveCfg cfg, const float *__restrict__ y0,
                                                  const double *__restrict__ t, float *sol, int *status, int *nfe,
                                                  int *nsteps)
{
    __shared__ Lds s;
    GridSync gs{w.sync, gridDim.x, 0ull, false};
    const int tid = threadIdx.x;
    const bool shared = cfg.control == PHX_CTRL_SHARED;
    const long long gtid = (long long)blockIdx.x * NT + tid, gsize = (long long)gridDim.x * NT;
    const int T = d.T;

    // ---- init: controllers + state
    for (long long cb = gtid; cb < d.Bc; cb += gsize) {
        const TimeRow tb = trow(t, d, cfg, (int)cb);
        float sg = 1.0f;
        int st = PHX_OK;
        if (T >= 2) {
            sg = (tb[1] < tb[0]) ? -1.0f : 1.0f;
            for (int i = 0; i + 1 < T; ++i)
                if (!((double)sg * tb[i + 1] > (double)sg * tb[i])) st = PHX_ERR_BAD_ARG;
        }
        w.sgn[cb] = sg;
        w.st[cb] = st;
        w.nsteps[cb] = 0;
        w.nsiv[cb] = 0;
        w.nfe[cb] = 0;
        w.rk_t0[cb] = (double)sg * tb[0];
        w.rk_t1[cb] = (double)sg * tb[0];
        w.out_idx[cb] = 1;
        const int dn = (T < 2 || st != PHX_OK) ? 1 : 0;
        w.done[cb] = dn;
        w.accept[cb] = 0;
        if (!dn && cfg.method == PHX_DOPRI5) atomicAdd(&w.sync->remaining, 1u);
    }
    for (int item = blockIdx.x; item < d.items; item += gridDim.x) {
        const int b = item / d.NC, c = item % d.NC, n = c * CH + tid;
        if (n < d.N) {
            const long long e = (long long)b * d.N + n;
            const float v = y0[e];
            w.Y0[e] = v;
            sol[e] = v;
        }
    }
    grid_barrier(gs, &s.flag);

    if (cfg.method != PHX_DOPRI5) {
        // ------------------------------------------------------------------ fixed grid
        const int S = fixed_nstages(cfg.method);
        for (int i = 0; i + 1 < T; ++i) {
            for (int st = 0; st < S; ++st) {
                for (int item = blockIdx.x; item < d.items; item += gridDim.x) {
                    const int b = item / d.NC, c = item % d.NC, n = c * CH + tid;
                    const int cb = shared ? 0 : b;
                    if (w.done[cb]) continue;
                    const bool valid = n < d.N;
                    const long long e = (long long)b * d.N + n;
                    const TimeRow tb = trow(t, d, cfg, b);
                    const double sg = (double)w.sgn[cb];
                    const double s0 = sg * tb[i], s1 = sg * tb[i + 1];
                    const float dt = cfg.t_is_f32 ? ((float)s1 - (float)s0) : (float)(s1 - s0);
                    float yv = 0.5f;
                    if (valid) {
                        yv = fixed_stage_input(cfg.method, st, w.Y0[e], w.KY, d.BN, e, dt);
                        w.YS[e] = yv;
                    }
                    phaseA<false>(net, d, w, item, c, yv, 0.f, valid, false, -1, e, s);
                }
                grid_barrier(gs, &s.flag);
                phaseB<false>(net, d, w, st, w.done, shared);
                grid_barrier(gs, &s.flag);
                for (int item = blockIdx.x; item < d.items; item += gridDim.x) {
                    const int b = item / d.NC, c = item % d.NC, n = c * CH + tid;
                    const int cb = shared ? 0 : b;
                    if (w.done[cb]) continue;
                    const bool valid = n < d.N;
                    const long long e = (long long)b * d.N + n;
                    const float yv = valid ? w.YS[e] : 0.5f;
                    float f, vjp, sgv;
                    phaseC<false>(net, d, w, b, c, st, yv, 0.f, valid, false, f, vjp, sgv, s);
                    if (valid) {
                        w.KY[(long long)st * d.BN + e] = w.sgn[cb] * f;
                        if (st == S - 1) {
                            const TimeRow tb = trow(t, d, cfg, b);
                            const double sg = (double)w.sgn[cb];
                            const double s0 = sg * tb[i], s1 = sg * tb[i + 1];
                            const float dt = cfg.t_is_f32 ? ((float)s1 - (float)s0) : (float)(s1 - s0);
                            const float y1 = fixed_final(cfg.method, w.Y0[e], w.KY, d.BN, e, dt);
                            w.Y0[e] = y1;
                            sol[(long long)(i + 1) * d.BN + e] = y1;
                        }
                    }
                }
            }
        }
        for (long long cb = gtid; cb < d.Bc; cb += gsize)
            if (!w.done[cb]) { w.nsteps[cb] = T - 1; w.nfe[cb] = S * (T - 1); }
    } else {
        // ------------------------------------------------------------------ dopri5
        // one RHS evaluation of every active trajectory: stage input -> K[kslot]
        //   mode 0: input = Y0 (f0)       reductions d0,d1        (misc.py:62-67)
        //   mode 1: input = Y0 + h0*K0    reductions d2           (misc.py:74-77)
        //   mode 2: RK stage `st`         (st == 6: error-ratio reductions, misc.py:89-91)
        auto eval = [&](int mode, int st, int kslot) {
            for (int item = blockIdx.x; item < d.items; item += gridDim.x) {
                const int b = item / d.NC, c = item % d.NC, n = c * CH + tid;
                const int cb = shared ? 0 : b;
                if (w.done[cb]) continue;
                const bool valid = n < d.N;
                const long long e = (long long)b * d.N + n;
                float yv = 0.5f;
                if (valid) {
                    if (mode == 0) yv = w.Y0[e];
                    else if (mode == 1) yv = w.Y0[e] + w.h0f[cb] * w.KY[e];
                    else yv = dp_stage_input(st, w.Y0[e], w.KY, d.BN, e, w.dtf[cb]);
                    w.YS[e] = yv;
                }
                phaseA<false>(net, d, w, item, c, yv, 0.f, valid, false, -1, e, s);
            }
            grid_barrier(gs, &s.flag);
            phaseB<false>(net, d, w, kslot, w.done, shared);
            grid_barrier(gs, &s.flag);
            for (int item = blockIdx.x; item < d.items; item += gridDim.x) {
                const int b = item / d.NC, c = item % d.NC, n = c * CH + tid;
                const int cb = shared ? 0 : b;
                if (w.done[cb]) continue;
                const bool valid = n < d.N;
                const long long e = (long long)b * d.N + n;
                const float yv = valid ? w.YS[e] : 0.5f;
                float f, vjp, sgv;
                phaseC<false>(net, d, w, b, c, kslot, yv, 0.f, valid, false, f, vjp, sgv, s);
                const float kv = w.sgn[cb] * f;
                float r0 = 0.f, r1 = 0.f;
                if (valid) {
                    if (mode == 0) {
                        w.KY[e] = kv;
                        const float y0v = yv;
                        const float scale = cfg.atol + fabsf(y0v) * cfg.rtol;
                        const float q0 = y0v / scale, q1 = kv / scale;
                        r0 = q0 * q0; r1 = q1 * q1;
                    } else if (mode == 1) {
                        const float y0v = w.Y0[e];
                        const float scale = cfg.atol + fabsf(y0v) * cfg.rtol;
                        const float q = (kv - w.KY[e]) / scale;
                        r0 = q * q;
                    } else {
                        w.KY[(long long)kslot * d.BN + e] = kv;
                        if (st == 6) {
                            const float dt = w.dtf[cb];
                            const float err = dp_combo(DP_CERR, w.KY, d.BN, e, dt);
                            const float y0v = w.Y0[e];
                            const float tol = cfg.atol + cfg.rtol * fmaxf(fabsf(y0v), fabsf(yv));
                            const float q = err / tol;
                            r0 = q * q;
                        }
                    }
                }
                if (mode != 2 || st == 6) {
                    const float t0 = block_sum(r0, s.red);
                    const float t1 = (mode == 0) ? block_sum(r1, s.red) : 0.f;
                    if (tid == 0) { w.red[(long long)item * 4 + 0] = t0; w.red[(long long)item * 4 + 1] = t1; }
                }
            }
        };
        const double cnt = shared ? (double)d.BN : (double)d.N;

        eval(0, 0, 0);
        grid_barrier(gs, &s.flag);
        for (long long cb = gtid; cb < d.Bc; cb += gsize) {
            if (w.done[cb]) continue;
            const float d0 = rms_from_sum(ctrl_sum(d, w, (int)cb, shared, 0), cnt);
            const float d1 = rms_from_sum(ctrl_sum(d, w, (int)cb, shared, 1), cnt);
            w.h0f[cb] = init_h0(d0, d1);
            w.xfin[cb] = (double)d1;  // keep d1 for the second half
        }
        grid_barrier(gs, &s.flag);
        eval(1, 0, 1);
        grid_barrier(gs, &s.flag);
        for (long long cb = gtid; cb < d.Bc; cb += gsize) {
            if (w.done[cb]) continue;
            const float h0 = w.h0f[cb];
            const float d2 = rms_from_sum(ctrl_sum(d, w, (int)cb, shared, 0), cnt) / h0;
            const double dt = init_dt(h0, (float)w.xfin[cb], d2);
            w.dt[cb] = dt;
            w.dtf[cb] = (float)dt;
            w.nfe[cb] = 2;
            const double t0 = w.rk_t1[cb];
            if (!(t0 + dt > t0)) {  // rk_common.py:175
                w.st[cb] = PHX_ERR_DT_UNDERFLOW;
                w.done[cb] = 1;
                atomicSub(&w.sync->remaining, 1u);
            }
        }
        grid_barrier(gs, &s.flag);
        if (tid == 0) s.remaining = __hip_atomic_load(&w.sync->remaining, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        __syncthreads();
        unsigned int remaining = s.remaining;

        while (remaining > 0 && !gs.aborted) {
            for (int st = 1; st <= 6; ++st) eval(2, st, st);
            grid_barrier(gs, &s.flag);
            // ---- controller: accept/reject, next dt, output scheduling (rk_common.py:150-220)
            for (long long cb = gtid; cb < d.Bc; cb += gsize) {
                if (w.done[cb]) continue;
                const TimeRow tb = trow(t, d, cfg, (int)cb);
                const double sg = (double)w.sgn[cb];
                const float ratio = rms_from_sum(ctrl_sum(d, w, (int)cb, shared, 0), cnt);
                const int acc = (ratio <= 1.0f) ? 1 : 0;
                const double t0 = w.rk_t1[cb], dt = w.dt[cb];
                w.rk_t0[cb] = t0;
                w.rk_t1[cb] = acc ? t0 + dt : t0;
                w.accept[cb] = acc;
                int ns = w.nsteps[cb] + 1;
                w.nsteps[cb] = ns;
                w.nfe[cb] += 6;
                int oi = w.out_idx[cb];
                w.out_lo[cb] = oi;
                if (acc) {
                    const double t1 = t0 + dt;
                    while (oi < T && sg * tb[oi] <= t1) ++oi;
                }
                w.out_hi[cb] = oi;
                w.out_idx[cb] = oi;
                const double dtn = optimal_step_size(dt, ratio);
                w.dtp[cb] = w.dtf[cb];
                w.dt[cb] = dtn;
                w.dtf[cb] = (float)dtn;
                // max_num_steps is a budget per output time (n_steps restarts in every _advance, rk_common.py:152-156)
                // and is tested before the step-size assert of the next step (rk_common.py:154,175)
                const int nsi = (oi > w.out_lo[cb]) ? 0 : w.nsiv[cb] + 1;
                w.nsiv[cb] = nsi;
                int dn = 0;
                if (oi >= T) dn = 1;
                else {
                    const double tn = w.rk_t1[cb];
                    if ((long long)nsi >= cfg.max_steps) { w.st[cb] = PHX_ERR_MAX_STEPS; dn = 1; }
                    else if (!(tn + dtn > tn)) { w.st[cb] = PHX_ERR_DT_UNDERFLOW; dn = 1; }
                }
                if (dn) {
                    w.fin[cb] = 1;  // finish after this step's accept pass
                    atomicSub(&w.sync->remaining, 1u);
                } else w.fin[cb] = 0;
            }
            grid_barrier(gs, &s.flag);
            if (tid == 0) s.remaining = __hip_atomic_load(&w.sync->remaining, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            __syncthreads();
            remaining = s.remaining;
            // ---- accept pass: dense output at the requested times, advance the state (FSAL)
            for (int item = blockIdx.x; item < d.items; item += gridDim.x) {
                const int b = item / d.NC, c = item % d.NC, n = c * CH + tid;
                const int cb = shared ? 0 : b;
                if (w.done[cb] || !w.accept[cb]) continue;
                if (n >= d.N) continue;
                const long long e = (long long)b * d.N + n;
                const float y0v = w.Y0[e], y1v = w.YS[e];
                const float f0 = w.KY[e], f1 = w.KY[6 * d.BN + e];
                const int lo = w.out_lo[cb], hi = w.out_hi[cb];
                if (hi > lo) {
                    // _interp_fit gets the fp32 cast of the dt of the step just taken (rk_common.py:224)
                    const double t0 = w.rk_t0[cb], t1 = w.rk_t1[cb];
                    const float dts = w.dtp[cb];
                    const float ym = y0v + dp_combo(DP_CMID, w.KY, d.BN, e, dts);
                    const TimeRow tb = trow(t, d, cfg, b);
                    const double sg = (double)w.sgn[cb];
                    for (int jo = lo; jo < hi; ++jo) {
                        const double x = (sg * tb[jo] - t0) / (t1 - t0);
                        const InterpX ix = make_interp_x(x);
                        sol[(long long)jo * d.BN + e] = interp_eval(y0v, y1v, ym, f0, f1, dts, ix);
                    }
                }
                w.Y0[e] = y1v;
                w.KY[e] = f1;
            }
            // controllers flagged `fin` retire now (their items were still needed by the accept pass)
            grid_barrier(gs, &s.flag);
            for (long long cb = gtid; cb < d.Bc; cb += gsize)
                if (!w.done[cb] && w.fin[cb]) w.done[cb] = 1;
            grid_barrier(gs, &s.flag);
        }
    }
    // ---- results
    grid_barrier(gs, &s.flag);
    for (long long b = gtid; b < d.B; b += gsize) {
        const int cb = shared ? 0 : (int)b;
        int st = w.st[cb];
        if (gs.aborted) st = PHX_ERR_SYNC_TIMEOUT;
        status[b] = st;
        nfe[b] = w.nfe[cb];
        nsteps[b] = w.nsteps[cb];
    }
    // outputs a failed trajectory never reached are NaN (a backward solve launched on them stops at once)
    for (long long e = gtid; e < d.BN; e += gsize) {
        const int cb = shared ? 0 : (int)(e / d.N);
        if (w.st[cb] != PHX_OK || gs.aborted)
            for (int jo = max(w.out_idx[cb], 1); jo < d.T; ++jo) sol[(long long)jo * d.BN + e] = __int_as_float(0x7fc00000);
    }
}

}  // namespace

// ========================================================================================
// adjoint solve (v0) and the v1 MFMA kernels live in their own files
// ========================================================================================
#include "phx_adjoint.inc"
#include "phx_mfma_common.inc"
#include "phx_mfma_fwd.inc"
#include "phx_mfma_adj.inc"
// the v1 solve kernels are instantiated in phx_v1.hip (own compiler flags); here they are only launched
#define PHX_V1_FWD(HT, MAXT, CH)                                                                                        \
    extern template __global__ void phxk::k1_solve_fwd<HT, MAXT, CH>(Net, D1, W1, SolveCfg, const float *, const double *, \
                                                                    float *, int *, int *, int *)
#define PHX_V1_ADJ(HT, MAXT, CH)                                                                                        \
    extern template __global__ void phxk::k1_solve_adj<HT, MAXT, CH>(Net, D1, W1, SolveCfg, const double *, const float *, \
                                                                    const float *, float *, int *, int *, int *, int,     \
                                                                    long long)
PHX_V1_FWD(7, 256, true);
PHX_V1_FWD(8, 256, true);
PHX_V1_FWD(3, 512, false);
PHX_V1_FWD(3, 256, false);
PHX_V1_FWD(8, 256, false);
PHX_V1_ADJ(7, 256, true);
PHX_V1_ADJ(8, 256, true);
PHX_V1_ADJ(3, 256, false);
PHX_V1_ADJ(8, 256, false);
#include "phx_mfma_eval.inc"
#include "phx_mfma_batch.inc"
#include "phx_prior.inc"
#include "phx_hill.inc"

// ========================================================================================
// host side: C ABI
// ========================================================================================
namespace {



inline bool bad_params(const phx_params *p)
{
    return !p || p->N <= 0 || p->H <= 0 || p->H > CH / 2 /* hidden vector staged in one LDS chunk */ || !p->Ws || !p->bs || !p->Wp || !p->bp || !p->WaT || !p->g;
}

inline int grid_for(int work_items)
{
    const int cus = num_cus();
    if (cus <= 0) return 0;
    return std::max(1, std::min(cus, work_items));
}

inline int launch_reduce(const Dims &d, const WS &w, const phx_grads *g, hipStream_t st)
{
    return launch_reduce_grads(w.dtheta, d.GB, d.PP, d.N, d.H, g, g->overwrite, st) ? PHX_OK : PHX_ERR_LAUNCH;
}


// ---------------------------------------------------------------------------------------------------
// v1 (MFMA) launch planning
// ---------------------------------------------------------------------------------------------------
constexpr size_t ADJ_LDS_EXTRA = 0;
constexpr int ADJ_NW_CAP = 4;   // the augmented kernel needs the 512-register budget of one wave per SIMD


// hidden-fragment tiles of the solve kernels: 3 for a narrow hidden layer, 8 (128 rows) otherwise -- except for a chunked
// hidden layer whose chunks need only 7 tiles (96 < Hc <= 112; the B-cell shape: H = 200 -> 2 x 100), where the eighth
// tile would be 12.5 % of the MFMAs, exchange rows and quadrature accumulators spent on padding
inline int solve_ht(int HC, int Hc) { return Hc <= 48 ? 3 : ((HC > 1 && Hc <= 112) ? 7 : 8); }

// picks (NW, TPW, NB): minimise the per-wave MFMA work TPW*NB subject to LDS and residency
bool plan_v1(int N, int H, int B, int T, int control, int nvec, int fwidth /* hidden frag tiles / HT */,
             size_t lds_per_block_extra, D1 *out, size_t ctl_extra_per_traj = 0, int nw_cap = 8, int calls = 1)
{
    // calls > 1 (shared control only): B = calls x Bcall rows, one batch group per call
    if (calls > 1 && (control != PHX_CTRL_SHARED || B % calls != 0)) return false;
    const int Bcall = calls > 1 ? B / calls : 0;
    const int cus = num_cus();
    if (cus <= 0 || force_v0()) return false;
    // H > 128: the hidden layer is cut into HC chunks of Hc <= 128 rows whose weights take turns in LDS
    const int HC = (H + 127) / 128, Hc = (H + HC - 1) / HC;
    const int HT = solve_ht(HC, Hc);
    if (HC > 1) nvec += (nvec >= NVEC_ADJ) ? (NVEC_ADJ_CH - NVEC_ADJ) : (NVEC_FWD_CH - NVEC_FWD);
    const size_t blkbytes = (size_t)blk_floats_ch(HT, Hc) * 4 + lds_per_block_extra;
    const int nblk = (N + 31) / 32, ntt = ((Bcall ? Bcall : B) + 15) / 16;
    long long best_cost = -1;
    D1 best{};
    int nwmax = std::min(nw_cap, (HT == 3 ? 8 : 4));
    if (const char *e = getenv("PHX_V1_MAXNW")) nwmax = std::min(nwmax, std::max(1, atoi(e)));
    for (int NW = nwmax; NW >= 1; NW >>= 1)
        for (int TPW = 1; TPW <= 4; TPW <<= 1) {
            const int slots = NW * TPW, TG = Bcall ? calls : (ntt + slots - 1) / slots;
            if (Bcall && slots < ntt) continue;   // a call is one group
            // a single group smaller than its wave slots: the spare waves become helpers of the exchange (the
            // reduce-scatter splits a row's members over all NW waves), the tile count is what the batch needs
            const int ntg = (TG == 1 || Bcall) ? std::min(slots, ntt) : slots, Bt = 16 * ntg;
            const bool helpers = ntg < slots;
            if (control == PHX_CTRL_SHARED && TG != 1 && !Bcall) continue;
            const size_t cb = ctl_bytes(Bt) + ctl_extra_per_traj * Bt;
            if (cb + blkbytes > LDS_BUDGET) continue;
            const int NBmax = (int)std::min<size_t>((LDS_BUDGET - cb) / blkbytes, 8);
            for (int NB = 1; NB <= NBmax; ++NB) {
                const int G = (nblk + NB - 1) / NB;
                if ((long long)TG * G > cus) continue;
                // per-SIMD MFMA work ~ TPW*NB*ceil(NW/4); prefer 2 waves per SIMD (latency hiding)
                // tie-break: fewer trajectories per group = smaller exchange volume and fewer members per reduction
                // (the forward kernel is lighter on registers and measured faster with two waves per SIMD)
                const long long cost = (long long)TPW * NB * ((NW + 3) / 4) * 1000 + (nw_cap >= 8 ? (8 - NW) * 100 : 0) + Bt / 4 -
                                       (helpers && TPW == 1 ? 50 : 0);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    best.N = N; best.H = H; best.B = B; best.T = T; best.HT = HT; best.NB = NB; best.NW = NW;
                    best.TPW = TPW; best.G = G; best.TG = TG; best.nblk = nblk; best.ntg = ntg; best.Bt = Bt;
                    best.nvec = nvec; best.BN = (long long)B * N; best.HC = HC; best.Hc = Hc;
                    best.Bcall = Bcall; best.cntN = (long long)(Bcall ? Bcall : B) * N;
                }
                break;  // smallest feasible NB for this (NW, TPW) is the cheapest
            }
        }
    (void)fwidth;
    if (best_cost < 0) return false;
    *out = best;
    return true;
}


// Batches larger than one residency: per-trajectory control makes trajectories independent, so the host walks the
// batch in chunks that the v1 plan accepts (pointers are offset, the [T,B,N] time stride stays B_total*N).
int pick_chunk_v1(int N, int H, int B, int T, int control, bool adj)
{
    D1 d1;
    auto ok = [&](int b) {
        return adj ? plan_v1(N, H, b, T, control, NVEC_ADJ, 4, ADJ_LDS_EXTRA, &d1, 40, ADJ_NW_CAP)
                   : plan_v1(N, H, b, T, control, NVEC_FWD, 2, 0, &d1);
    };
    if (ok(B)) return B;
    if (control != PHX_CTRL_PER_TRAJECTORY) return 0;
    for (int bc = 4096; bc >= 16; bc >>= 1)
        if (bc < B && ok(bc)) return bc;
    return 0;
}

// `calls` independent shared-control odeint calls of B/calls rows each: the largest number of calls one launch takes
// (one batch group per call), 0 when not even a single call can be planned.
int pick_calls_v1(int N, int H, int B, int T, int calls)
{
    if (calls < 1 || B % calls != 0) return 0;
    const int Bcall = B / calls;
    D1 d1;
    for (int n = std::min(calls, std::max(1, num_cus())); n >= 1; --n)
        if (plan_v1(N, H, n * Bcall, T, PHX_CTRL_SHARED, NVEC_FWD, 2, 0, &d1, 0, 8, n)) return n;
    return 0;
}

struct Layout1 {
    size_t total, cnt, part, zbuf, scratch, dtheta, prof, xbytes, wimg;
};

// ftiles: hidden fragment tiles exchanged per trajectory tile (2HT forward, 4HT adjoint)
Layout1 make_layout1(const D1 &d, int ftiles, bool grads, int zslots = 1)
{
    Layout1 L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t R = (size_t)d.ntg * ftiles * d.HC * 4 + d.ntg;   // hidden rows (all chunks) + norm rows per group
    L.cnt = take(4096);
    L.part = take((size_t)d.TG * d.G * R * 64 * 8);
    L.zbuf = take((size_t)zslots * d.TG * R * 64 * 8);
    L.xbytes = off - L.part;                                // granule buffers are zeroed before every launch
    L.scratch = take((size_t)d.TG * d.G * d.nvec * d.ntg * d.NB * 512 * 4);
    const size_t PP = align_up((size_t)4 * d.H * d.N + d.N + 2 * d.H, 4);
    L.dtheta = take(grads ? PP * 4 * d.TG * d.NW : 0);
    L.prof = take((size_t)d.TG * d.G * 16 * 8);
    L.wimg = take((size_t)d.nblk * d.HC * blk_floats_ch(d.HT, d.Hc) * 4);
    L.total = off;
    return L;
}

W1 make_w1(void *base, const Layout1 &L)
{
    char *p = (char *)base;
    W1 w{};
    w.cnt = (unsigned long long *)(p + L.cnt);
    w.abort_flag = (unsigned int *)(p + L.cnt + 2048);
    w.part = (unsigned long long *)(p + L.part);
    w.zbuf = (unsigned long long *)(p + L.zbuf);
    w.scratch = (float *)(p + L.scratch);
    w.dtheta = (float *)(p + L.dtheta);
    const char *pe = getenv("PHX_PROF");
    w.prof = (pe && pe[0] == '1') ? (unsigned long long *)(p + L.prof) : nullptr;
    w.wimg = (const float *)(p + L.wimg);
    w.hq = nullptr;
    return w;
}

size_t lds_bytes_v1(const D1 &d, size_t per_block_extra)
{
    return (size_t)blk_floats_ch(d.HT, d.Hc) * 4 * d.NB + per_block_extra * d.NB + ctl_bytes(d.Bt);
}


// ---- v1 standalone evaluation (large batches, passes)
struct PlanEval {
    D1 d;
    int npass;
    size_t lds;
};

bool plan_eval(int N, int H, int B, int nbc_cap, PlanEval *out)
{
    const int cus = num_cus();
    if (cus <= 0 || H > 128 || force_v0()) return false;
    const int HT = H <= 48 ? 3 : 8;
    const size_t blkbytes = (size_t)blk_floats(HT, H) * 4;
    const size_t extra = 16 * 64 * 4;
    if (blkbytes + extra > LDS_BUDGET) return false;
    int NB = (int)std::min<size_t>((LDS_BUDGET - extra) / blkbytes, (size_t)nbc_cap);
    const int nblk = (N + 31) / 32, ntt = (B + 15) / 16;
    NB = std::min(NB, nblk);
    int G = (nblk + NB - 1) / NB;
    while (G > cus && NB < nbc_cap) { NB++; G = (nblk + NB - 1) / NB; }
    if (G > cus || (size_t)NB * blkbytes + extra > LDS_BUDGET) return false;
    const int NW = 4;
    int TG = std::max(1, cus / G);
    TG = std::min(TG, (ntt + NW - 1) / NW);
    D1 &d = out->d;
    d = D1{};
    d.N = N; d.H = H; d.B = B; d.T = 0; d.HT = HT; d.NB = NB; d.NW = NW; d.TPW = 1; d.G = G; d.TG = TG;
    d.nblk = nblk; d.ntg = NW; d.Bt = 16 * NW; d.nvec = 0; d.BN = (long long)B * N;
    d.HC = 1; d.Hc = H;
    out->npass = (ntt + TG * NW - 1) / (TG * NW);
    out->lds = (size_t)NB * blkbytes + extra;
    return true;
}

struct LayoutE {
    size_t total, cnt, part, zbuf, dtheta, xbytes;
};

LayoutE make_layout_eval(const PlanEval &pe, bool aug)
{
    const D1 &d = pe.d;
    LayoutE L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t R = (size_t)d.NW * (aug ? 4 : 2) * d.HT * 4;
    L.cnt = take(4096);
    L.part = take((size_t)d.TG * d.G * R * 64 * 8);
    L.zbuf = take((size_t)(aug ? pe.npass : 1) * d.TG * R * 64 * 8);
    L.xbytes = off - L.part;
    const size_t PP = align_up((size_t)4 * d.H * d.N + d.N + 2 * d.H, 4);
    L.dtheta = take(aug ? PP * 4 * d.TG * d.NW : 0);
    L.total = off;
    return L;
}

constexpr int EVAL_NBC_HT3_MAX = 4, EVAL_NBC_HT8 = 1;
// gene blocks whose gradient accumulators a wave of k1_eval_pgrad keeps in registers (HT = 3); PHX_EVAL_NBC overrides
int eval_nbc_ht3()
{
    int n = 2;
    if (const char *e = getenv("PHX_EVAL_NBC")) n = atoi(e);
    return std::min(EVAL_NBC_HT3_MAX, std::max(2, n));
}

// ---- exchange-free parameter gradients of the prior branch (phx_mfma_batch.inc)
// Standalone evaluation of the RHS: the kernel chain (A -> R -> D) from this many rows up, the pass-structured k1_eval_fwd
// below.  With the caller's packed weight images the chain's launches stage in one round trip and it wins at EVERY batch
// size (round 4, C4: 16 rows 37 us against 77; 256: 64 / 85; 1000: 81 / 190); without images each of its two sweep
// kernels pays the gather staging (25-40 us at C4) and small batches stay on the one-launch kernel.  PHX_BATCH_MIN_ROWS
// overrides (tests run the small shapes on both).
inline int batch_fwd_min_rows(bool have_images)
{
    if (const char *e = getenv("PHX_BATCH_MIN_ROWS")) return std::max(1, atoi(e));
    return have_images ? 1 : 1024;
}
struct PlanBatch {
    D1 d;              // kernel A: gene tiles (NB blocks resident in LDS) x TG sweep groups
    size_t ldsA;
    int ntiles, chunk_tiles, chunk_tiles2, KS, pgW, pgG4, slabs;   // KS / pgW: partial buffers / workgroups of kernel C; chunk_tiles2: launches whose partials have half the rows (u | v or dz only)
    long long Kp;
    size_t part, hdt, dtheta, total;   // workspace offsets
    size_t zl, losspart, total_fwd;    // forward path: per-chunk z rows, per-wave loss partials (fused MSE head)
    size_t zl4, dgpart, total_vjp;     // full VJP: the four hidden sections in L_H layout per chunk, per-wave dg partials
};

bool plan_batch(int N, int H, int B, PlanBatch *out)
{
    const int cus = num_cus();
    if (cus <= 0 || force_v0()) return false;
    if (const char *e = getenv("PHX_PGRAD")) if (strcmp(e, "v1") == 0) return false;
    const int HC = (H + 127) / 128, Hc = (H + HC - 1) / HC;     // hidden chunks as in plan_v1: one chain per chunk
    const int HT = solve_ht(HC, Hc);   // the solve kernels' tiling: the caller's packed images fit (7 tiles for a 100-row chunk)
    const size_t blkbytes = (size_t)blk_floats_ch(HT, Hc) * 4;   // the slot size of the packed images (stage_block_weights)
    if (blkbytes > LDS_BUDGET) return false;
    const int nblk = (N + 31) / 32;
    const int NB = (int)std::min<size_t>(std::min<size_t>(LDS_BUDGET / blkbytes, 6), (size_t)nblk);
    D1 &d = out->d;
    d = D1{};
    d.N = N; d.H = H; d.B = B; d.HT = HT; d.NB = NB; d.NW = 4; d.TPW = 1; d.nblk = nblk;
    d.G = (nblk + NB - 1) / NB;
    d.TG = std::max(1, cus / d.G);
    d.HC = HC; d.Hc = Hc; d.BN = (long long)B * N;
    out->ldsA = (size_t)NB * blkbytes;
    out->ntiles = (B + 15) / 16;
    out->Kp = (long long)out->ntiles * 16;
    const size_t per_tile = (size_t)16 * HT * d.G * 64 * 4;           // FR rows x G members x 64 lanes
    // partials of one chunk: ~128 MB (Infinity Cache resident), but never fewer than 64 tiles per weight staging; the
    // chunks of a batch are made equal (625 tiles: 2 x 313 instead of 370 + 255: the sweep groups stay balanced)
    auto even_chunks = [&](size_t cmax) {
        const size_t nch = ((size_t)out->ntiles + cmax - 1) / cmax;
        return (int)(((size_t)out->ntiles + nch - 1) / nch);
    };
    size_t chunk_min = 64;
    if (const char *e = getenv("PHX_BATCH_CHUNK_MIN")) chunk_min = (size_t)std::max(1, atoi(e));
    out->chunk_tiles = even_chunks(std::max<size_t>(chunk_min, ((size_t)128 << 20) / per_tile));
    out->chunk_tiles2 = even_chunks(std::max<size_t>(chunk_min, ((size_t)256 << 20) / per_tile));
    const size_t part_bytes = std::max(per_tile * out->chunk_tiles, per_tile / 2 * out->chunk_tiles2);
    // kernel C: 64-gene slabs, four workgroups per CU.  (PHX_PGRAD_G4=8: 128-gene slabs, two per CU, half the A-operand
    // reads per MFMA -- measured slower at C4, 570 against 522 us: those reads are not what paces the kernel.)
    out->pgG4 = 4;
    if (const char *e = getenv("PHX_PGRAD_G4")) out->pgG4 = (atoi(e) == 8 && HT == 3) ? 8 : 4;
    out->slabs = (N + 16 * out->pgG4 - 1) / (16 * out->pgG4);
    {   // kernel C: as many workgroups per CU as its registers allow take equal ranges of the (slab, step) units; a
        // slab is then shared by at most KS of them (k2_pgrad_contract); never fewer than 8 steps per workgroup
        int per_cu = out->pgG4 == 8 ? 2 : 4;
        if (const char *e = getenv("PHX_PGRAD_WGS")) per_cu = std::max(1, std::min(8, atoi(e)));
        const long long U = (long long)out->slabs * out->ntiles;
        const long long W = std::max<long long>(1, std::min<long long>((long long)per_cu * cus, std::max<long long>(out->slabs, U / 8)));
        const long long per = std::max<long long>(1, U / W);
        out->pgW = (int)W;
        out->KS = (int)std::min<long long>(out->ntiles, (out->ntiles + per - 1) / per + 1);
    }
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t PP = align_up((size_t)4 * H * N + N + 2 * H, 4);
    out->part = take(part_bytes);
    out->hdt = take((size_t)4 * 16 * HT * out->Kp * 4);
    out->dtheta = take(PP * 4 * out->KS);
    out->total = off;
    out->zl4 = take((size_t)out->chunk_tiles * 4 * HT * 4 * 64 * 4);
    out->dgpart = take((size_t)d.TG * 4 * N * 4);
    out->total_vjp = off;
    off = align_up(part_bytes, 256);   // forward path reuses the partial buffer (half as many rows: chunk_tiles2)
    out->zl = take((size_t)out->chunk_tiles2 * 2 * HT * 4 * 64 * 4);
    out->losspart = take((size_t)d.TG * d.G * 16 * sizeof(double));   // one slot per wave (<= 16 waves per workgroup)
    out->total_fwd = off;
    return true;
}


// The Net of the batch kernels: with the caller's packed weight images when their tiling is the batch kernels' own
// (both follow solve_ht since round 4; the check stays in case the two plans diverge again)
// threads per workgroup of the sweep kernels of the batch chain (PHX_BATCH_WAVES: diagnostic, fewer waves than compiled for)
template <int HT>
inline int batch_threads(bool expansion = false)
{
    int t = expansion ? BatchThreads<HT>::D : BatchThreads<HT>::A;
    if (const char *e = getenv("PHX_BATCH_WAVES")) t = std::max(64, std::min(t, 64 * atoi(e)));
    return t;
}

inline Net batch_net(const PlanBatch &pb, const phx_params *p)
{
    Net n = to_net(p);
    n.wimg = (p->wimg && solve_ht(pb.d.HC, pb.d.Hc) == pb.d.HT && !getenv("PHX_BATCH_NO_WIMG")) ? (const float *)p->wimg : nullptr;
    return n;
}

template <int HT>
inline void launch_pgrad_contract(const PlanBatch &pb, hipStream_t st, const Net &net, const float *y, const float *cot,
                                  const float *hdt, float *dth, long long PP, int hb, int hc, int full)
{
    if constexpr (HT == 3) {
        if (pb.pgG4 == 8) {
            hipLaunchKernelGGL((k2_pgrad_contract<HT, 8>), dim3(pb.pgW), dim3(256), 0, st, net, y, cot, hdt, dth, pb.d.B,
                               pb.ntiles, pb.Kp, pb.KS, PP, hb, hc, full);
            return;
        }
    }
    hipLaunchKernelGGL((k2_pgrad_contract<HT, 4>), dim3(pb.pgW), dim3(256), 0, st, net, y, cot, hdt, dth, pb.d.B, pb.ntiles,
                       pb.Kp, pb.KS, PP, hb, hc, full);
}

// hsaved (optional, unchunked hidden layer only): the four hidden sections du | dv | z_u | z_p of every row, kept by the
// forward chain of the same step (phx_prior_mse_save) in kernel C's operand layout -- kernels A and R are not run
template <int HT>
int launch_batch_pgrad(const PlanBatch &pb, const phx_params *p, const float *y, const float *cot, const phx_grads *grads,
                       char *base, hipStream_t st, const float *hsaved = nullptr)
{
    if (pb.d.HC != 1) hsaved = nullptr;
    float *part = (float *)(base + pb.part), *hdt = (float *)(base + pb.hdt), *dth = (float *)(base + pb.dtheta);
    const long long PP = (long long)align_up((size_t)4 * p->H * p->N + p->N + 2 * p->H, 4);
    if (hipMemsetAsync(dth, 0, sizeof(float) * (size_t)PP * pb.KS, st) != hipSuccess) return PHX_ERR_LAUNCH;
    if (!set_lds(k2_hidden_partials<HT, true>, pb.ldsA)) return PHX_ERR_LAUNCH;
    for (int ch = 0; ch < pb.d.HC; ++ch) {      // hidden chunks are independent slices of the same gradient
        const int hb = ch * pb.d.Hc, hc = std::min(pb.d.Hc, p->H - hb);
        for (int t0 = 0; t0 < pb.ntiles && !hsaved; t0 += pb.chunk_tiles) {
            const int nt = std::min(pb.chunk_tiles, pb.ntiles - t0);
            hipLaunchKernelGGL((k2_hidden_partials<HT, true>), dim3(pb.d.TG * pb.d.G), dim3(batch_threads<HT>()),
                               pb.ldsA, st, batch_net(pb, p), pb.d, y, cot, part, t0, nt, hb, hc);
            const int tasks = nt * HT * 4;
            hipLaunchKernelGGL((k2_hidden_reduce<HT, true>), dim3((tasks + 3) / 4), dim3(256), 0, st, to_net(p), part, hdt,
                               pb.d.G, t0, nt, pb.Kp, hb, hc, (float *)nullptr, (float *)nullptr);
        }
        launch_pgrad_contract<HT>(pb, st, to_net(p), y, cot, hsaved ? hsaved : hdt, dth, PP, hb, hc, 0);
    }
    if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
    return launch_reduce_grads(dth, pb.KS, PP, p->N, p->H, grads, grads->overwrite, st) ? PHX_OK : PHX_ERR_LAUNCH;
}

// Full VJP of the RHS (or of prior_only_forward) on a batch: per hidden chunk A -> R per row chunk, E (input VJP, f, dg
// partials) per row chunk, C (parameter gradients) once per hidden chunk.  Any of vjp_y / grads / f_out may be null.
template <int HT>
int launch_batch_vjp(const PlanBatch &pb, const phx_params *p, const float *y, const float *cot, float *vjp_y,
                     const phx_grads *grads, float *f_out, int prior_only, char *base, hipStream_t st)
{
    float *part = (float *)(base + pb.part), *hdt = (float *)(base + pb.hdt), *dth = (float *)(base + pb.dtheta);
    float *zl4 = (float *)(base + pb.zl4), *dgp = (float *)(base + pb.dgpart);
    const long long PP = (long long)align_up((size_t)4 * p->H * p->N + p->N + 2 * p->H, 4);
    const int full = prior_only ? 0 : 1;
    const bool want_e = vjp_y || f_out || (grads && full);
    if (grads && hipMemsetAsync(dth, 0, sizeof(float) * (size_t)PP * pb.KS, st) != hipSuccess) return PHX_ERR_LAUNCH;
    if (!set_lds(k2_hidden_partials<HT, true>, pb.ldsA) || !set_lds(k2_expand_vjp<HT>, pb.ldsA)) return PHX_ERR_LAUNCH;
    for (int ch = 0; ch < pb.d.HC; ++ch) {   // hidden chunks: independent slices whose contributions add up (kernel E)
        const int hb = ch * pb.d.Hc, hc = std::min(pb.d.Hc, p->H - hb);
        for (int t0 = 0; t0 < pb.ntiles; t0 += pb.chunk_tiles) {
            const int nt = std::min(pb.chunk_tiles, pb.ntiles - t0);
            hipLaunchKernelGGL((k2_hidden_partials<HT, true>), dim3(pb.d.TG * pb.d.G), dim3(batch_threads<HT>()), pb.ldsA,
                               st, batch_net(pb, p), pb.d, y, cot, part, t0, nt, hb, hc, full);
            const int tasks = nt * HT * 4;
            hipLaunchKernelGGL((k2_hidden_reduce<HT, true>), dim3((tasks + 3) / 4), dim3(256), 0, st, to_net(p), part, hdt,
                               pb.d.G, t0, nt, pb.Kp, hb, hc, want_e ? zl4 : (float *)nullptr);
            if (want_e)
                hipLaunchKernelGGL((k2_expand_vjp<HT>), dim3(pb.d.TG * pb.d.G), dim3(256), pb.ldsA, st, batch_net(pb, p), pb.d, y,
                                   cot, zl4, vjp_y, f_out, (grads && full) ? dgp : (float *)nullptr, prior_only, t0, nt,
                                   (t0 == 0 && ch == 0) ? 1 : 0, hb, hc);
        }
        if (grads)
            launch_pgrad_contract<HT>(pb, st, to_net(p), y, cot, hdt, dth, PP, hb, hc, full);
    }
    if (grads) {
        if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
        if (!launch_reduce_grads(dth, pb.KS, PP, p->N, p->H, grads, grads->overwrite, st)) return PHX_ERR_LAUNCH;
        if (full)
            hipLaunchKernelGGL(k2_dg_finish, dim3((p->N + 255) / 256), dim3(256), 0, st, dgp, pb.d.TG * 4, p->N, p->g,
                               grads->g);
    }
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

// prior_only_forward / ODENet.forward on a large batch: A (u, v partials) -> R (z rows) -> D (expansion) per chunk.
// hsave (phx_prior_mse_save; unchunked hidden layer): D also contracts the loss cotangent with the Wa rows (dz partials, in
// the partial buffer A has just been reduced from) and a second R writes the four hidden sections du | dv | z_u | z_p of
// the chunk's rows into `hsave` ([4][16 HT][Kp], kernel C's A operand): the backward of the step is kernel C alone.
template <int HT>
int launch_batch_forward(const PlanBatch &pb, const phx_params *p, const float *y, float *out, int prior_only, char *base,
                         hipStream_t st, const float *target = nullptr, float *loss = nullptr, float *hsave = nullptr)
{
    if (pb.d.HC != 1 || !target) hsave = nullptr;   // (the saved rows are those of an unchunked hidden layer)
    float *part = (float *)(base + pb.part), *zl = (float *)(base + pb.zl);
    double *loss_part = (double *)(base + pb.losspart);
    const float cot_scale = (float)(2.0 / ((double)pb.d.B * (double)p->N));
    if (target && hipMemsetAsync(loss_part, 0, sizeof(double) * (size_t)pb.d.TG * pb.d.G * 16, st) != hipSuccess)
        return PHX_ERR_LAUNCH;
    if (!set_lds(k2_hidden_partials<HT, false>, pb.ldsA) || !set_lds(k2_expand<HT, false>, pb.ldsA) ||
        !set_lds(k2_expand<HT, true>, pb.ldsA))
        return PHX_ERR_LAUNCH;
    const dim3 grid(pb.d.TG * pb.d.G), blk(batch_threads<HT>()), blkD(batch_threads<HT>(true));
    for (int t0 = 0; t0 < pb.ntiles; t0 += pb.chunk_tiles2) {
        const int nt = std::min(pb.chunk_tiles2, pb.ntiles - t0);
        for (int ch = 0; ch < pb.d.HC; ++ch) {
            const int hb = ch * pb.d.Hc, hc = std::min(pb.d.Hc, p->H - hb);
            hipLaunchKernelGGL((k2_hidden_partials<HT, false>), grid, blk, pb.ldsA, st, batch_net(pb, p), pb.d, y,
                               (const float *)nullptr, part, t0, nt, hb, hc);
            const int tasks = nt * HT * 4;
            hipLaunchKernelGGL((k2_hidden_reduce<HT, false>), dim3((tasks + 3) / 4), dim3(256), 0, st, to_net(p), part, zl,
                               pb.d.G, t0, nt, pb.Kp, hb, hc, (float *)nullptr, (float *)nullptr);
            if (hsave) {
                hipLaunchKernelGGL((k2_expand<HT, true>), grid, blkD, pb.ldsA, st, batch_net(pb, p), pb.d, y, zl, out,
                                   prior_only, t0, nt, hb, hc, 1, 1, target, cot_scale, loss_part, part);
                hipLaunchKernelGGL((k2_hidden_reduce<HT, true>), dim3((tasks + 3) / 4), dim3(256), 0, st, to_net(p), part,
                                   hsave, pb.d.G, t0, nt, pb.Kp, hb, hc, (float *)nullptr, zl, 1);
            } else {
                hipLaunchKernelGGL((k2_expand<HT, false>), grid, blkD, pb.ldsA, st, batch_net(pb, p), pb.d, y, zl, out,
                                   prior_only, t0, nt, hb, hc, ch == 0 ? 1 : 0, ch == pb.d.HC - 1 ? 1 : 0, target, cot_scale,
                                   loss_part);
            }
        }
    }
    if (target) {
        const int nslots = pb.d.TG * pb.d.G * (batch_threads<HT>(true) / 64);
        hipLaunchKernelGGL(k2_loss_finish, dim3(1), dim3(64), 0, st, loss_part, nslots,
                           1.0 / ((double)pb.d.B * (double)p->N), loss);
    }
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

}  // namespace

extern "C" {

int phx_abi_version(void) { return PHX_ABI_VERSION; }

const char *phx_status_string(int s)
{
    switch (s) {
        case PHX_OK: return "ok";
        case PHX_ERR_MAX_STEPS: return "max_num_steps exceeded";
        case PHX_ERR_DT_UNDERFLOW: return "underflow in dt";
        case PHX_ERR_NONFINITE: return "non-finite values in state `y`";
        case PHX_ERR_BAD_ARG: return "bad argument";
        case PHX_ERR_WORKSPACE: return "workspace too small";
        case PHX_ERR_LAUNCH: return "HIP launch failure";
        case PHX_ERR_SYNC_TIMEOUT: return "in-kernel grid barrier timed out";
        default: return "unknown status";
    }
}

int phx_device_cus(void) { return num_cus(); }

int phx_hill_rhs(const int *code, const int *off, const int *len, const float *consts, const float *x, float *out, int B,
                 int N, void *stream)
{
    if (!code || !off || !len || !consts || !x || !out || B <= 0 || N <= 0) return PHX_ERR_BAD_ARG;
    const HillProg p{reinterpret_cast<const int2 *>(code), off, len, consts};
    hipLaunchKernelGGL(k_hill_rhs, dim3((N + 255) / 256, B), dim3(256), 0, (hipStream_t)stream, p, x, out, B, N);
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

int phx_hill_simulate(const int *code, const int *off, const int *len, const float *consts, const float *x0,
                      const double *times, int T, double dt_max, float *out, int B, int N, void *stream)
{
    if (!code || !off || !len || !consts || !x0 || !times || !out || B <= 0 || N <= 0 || T < 1 || !(dt_max > 0.0) ||
        N > HILL_GPT * 1024)
        return PHX_ERR_BAD_ARG;
    const HillProg p{reinterpret_cast<const int2 *>(code), off, len, consts};
    const int threads = std::min(1024, ((N + 63) / 64) * 64);
    hipLaunchKernelGGL(k_hill_simulate, dim3(B), dim3(threads), sizeof(float) * (size_t)N, (hipStream_t)stream, p, x0, times, T,
                       dt_max, out, B, N);
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

int phx_prior_mse(const phx_params *p, const float *X, const float *target, int B, float *cot, float *loss,
                  void *workspace, size_t workspace_bytes, void *stream)
{
    if (bad_params(p) || !X || !target || !cot || !loss || B <= 0 || !workspace) return PHX_ERR_BAD_ARG;
    PlanBatch pb;
    if (!plan_batch(p->N, p->H, B, &pb)) return PHX_ERR_BAD_ARG;   // callers fall back to the unfused formula
    if (workspace_bytes < pb.total_fwd) return PHX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    return pb.d.HT == 3 ? launch_batch_forward<3>(pb, p, X, cot, 1, (char *)workspace, st, target, loss)
                        : pb.d.HT == 7 ? launch_batch_forward<7>(pb, p, X, cot, 1, (char *)workspace, st, target, loss)
                                       : launch_batch_forward<8>(pb, p, X, cot, 1, (char *)workspace, st, target, loss);
}

size_t phx_prior_z_bytes(int N, int H, int B)
{
    PlanBatch pb;
    if (N <= 0 || H <= 0 || B <= 0 || !plan_batch(N, H, B, &pb) || pb.d.HC != 1) return 0;
    return (size_t)4 * 16 * pb.d.HT * pb.Kp * sizeof(float);   // du | dv | z_u | z_p, [16 HT] rows x Kp trajectories each
}

int phx_prior_mse_save(const phx_params *p, const float *X, const float *target, int B, float *cot, float *loss,
                       float *z_save, void *workspace, size_t workspace_bytes, void *stream)
{
    if (bad_params(p) || !X || !target || !cot || !loss || !z_save || B <= 0 || !workspace) return PHX_ERR_BAD_ARG;
    PlanBatch pb;
    if (!plan_batch(p->N, p->H, B, &pb) || pb.d.HC != 1) return PHX_ERR_BAD_ARG;
    if (workspace_bytes < pb.total_fwd) return PHX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    return pb.d.HT == 3 ? launch_batch_forward<3>(pb, p, X, cot, 1, (char *)workspace, st, target, loss, z_save)
                        : pb.d.HT == 7 ? launch_batch_forward<7>(pb, p, X, cot, 1, (char *)workspace, st, target, loss, z_save)
                                       : launch_batch_forward<8>(pb, p, X, cot, 1, (char *)workspace, st, target, loss, z_save);
}

int phx_prior_vjp_saved(const phx_params *p, const float *X, const float *cot, const float *z_saved, const phx_grads *grads,
                        int B, void *workspace, size_t workspace_bytes, void *stream)
{
    if (bad_params(p) || !X || !cot || !z_saved || !grads || B <= 0 || !workspace) return PHX_ERR_BAD_ARG;
    if (!grads->Ws || !grads->bs || !grads->Wp || !grads->bp || (!grads->WaT && !grads->Wa) || !grads->g) return PHX_ERR_BAD_ARG;
    PlanBatch pb;
    if (!plan_batch(p->N, p->H, B, &pb) || pb.d.HC != 1) return PHX_ERR_BAD_ARG;
    if (workspace_bytes < pb.total) return PHX_ERR_WORKSPACE;
    hipStream_t st = (hipStream_t)stream;
    return pb.d.HT == 3 ? launch_batch_pgrad<3>(pb, p, X, cot, grads, (char *)workspace, st, z_saved)
                        : pb.d.HT == 7 ? launch_batch_pgrad<7>(pb, p, X, cot, grads, (char *)workspace, st, z_saved)
                                       : launch_batch_pgrad<8>(pb, p, X, cot, grads, (char *)workspace, st, z_saved);
}

int phx_prior_targets(const int *colptr, const int *rowidx, const float *vals, const float *X, float *out, int K, int N,
                       void *stream)
{
    if (!colptr || !rowidx || !vals || !X || !out || K <= 0 || N <= 0) return PHX_ERR_BAD_ARG;
    const dim3 grid((N + 255) / 256, std::min(K, 2048));
    hipLaunchKernelGGL(k_prior_spmm, grid, dim3(256), 0, (hipStream_t)stream, colptr, rowidx, vals, X, out, K, N);
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

int phx_prior_targets_sell(const long long *sptr, const int *width, const int *ridx, const float *vals, const float *X,
                           float *out, int K, int N, void *stream)
{
    if (!sptr || !width || !ridx || !vals || !X || !out || K <= 0 || N <= 0) return PHX_ERR_BAD_ARG;
    const int nslices = (N + 63) / 64;
    const size_t rowbytes = (size_t)N * 4;
    const int RT = (int)std::min<size_t>(4, LDS_BUDGET / rowbytes);
    if (RT < 1) return PHX_ERR_BAD_ARG;   // a row of X does not fit LDS: use phx_prior_targets
    const int grid = std::max(1, std::min((K + RT - 1) / RT, 4 * std::max(1, num_cus())));
    const size_t lds = (size_t)RT * rowbytes;
    hipStream_t st = (hipStream_t)stream;
    auto go = [&](auto kern) -> int {
        if (!set_lds_fn(reinterpret_cast<const void *>(kern), lds)) return PHX_ERR_LAUNCH;
        hipLaunchKernelGGL(kern, dim3(grid), dim3(1024), lds, st, sptr, width, ridx, vals, X, out, K, N, nslices);
        return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
    };
    switch (RT) {
        case 1: return go(k_prior_spmm_sell<1>);
        case 2: return go(k_prior_spmm_sell<2>);
        case 3: return go(k_prior_spmm_sell<3>);
        default: return go(k_prior_spmm_sell<4>);
    }
}

void phx_debug_queue_kernel_events(void *ev_start, void *ev_stop)
{
    std::lock_guard<std::mutex> lk(g_evq_mu);
    if (!ev_start && !ev_stop) g_evq.clear();
    else g_evq.emplace_back((hipEvent_t)ev_start, (hipEvent_t)ev_stop);
}

void phx_debug_set_kernel_events(void *ev_start, void *ev_stop)
{
    g_ev_start = (hipEvent_t)ev_start;
    g_ev_stop = (hipEvent_t)ev_stop;
}

int phx_debug_profile_region(int op, int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups,
                             int *plan /* [NW, TPW, NB, G, TG, HT] */)
{
    D1 d1;
    if (op != PHX_OP_ODEINT && op != PHX_OP_ADJOINT) return PHX_ERR_BAD_ARG;
    const bool adj = op == PHX_OP_ADJOINT;
    if (!adj && fwd3_profile_region(N, H, B, T, control, offset, n_workgroups, plan) == PHX_OK) return PHX_OK;
    if (!adj && fwd3c_profile_region(N, H, B, T, control, offset, n_workgroups, plan) == PHX_OK) return PHX_OK;
    if (adj && adj3_profile_region(N, H, B, T, control, offset, n_workgroups, plan) == PHX_OK) return PHX_OK;
    if (adj && adj3c_profile_region(N, H, B, T, control, offset, n_workgroups, plan) == PHX_OK) return PHX_OK;
    if (adj && adj2_profile_region(N, H, B, T, control, offset, n_workgroups, plan) == PHX_OK) return PHX_OK;
    if (!plan_v1(N, H, B, T, control, adj ? NVEC_ADJ : NVEC_FWD, 2, adj ? ADJ_LDS_EXTRA : 0, &d1, adj ? 40 : 0,
                 adj ? ADJ_NW_CAP : 8))
        return PHX_ERR_BAD_ARG;
    const Layout1 L1 = make_layout1(d1, adj ? 4 * d1.HT : 2 * d1.HT, adj, adj ? 7 : 1);
    *offset = L1.prof;
    *n_workgroups = d1.TG * d1.G;
    if (plan) { plan[0] = d1.NW; plan[1] = d1.TPW; plan[2] = d1.NB; plan[3] = d1.G; plan[4] = d1.TG; plan[5] = d1.HT; }
    return PHX_OK;
}

int phx_debug_adjoint_kernel(int N, int H, int B, int T, int control)
{
    return phx_debug_adjoint_kernel_m(N, H, B, T, control, PHX_DOPRI5);
}

int phx_debug_adjoint_kernel_m(int N, int H, int B, int T, int control, int method)
{
    if (adj3_chunk(N, H, B, T, control, method) > 0) return 3;
    if (adj3c_chunk(N, H, B, T, control, method) > 0) return 4;   // third generation, chunked hidden layer (H > 48)
    if (adj2_chunk(N, H, B, T, control) > 0) return 2;
    return pick_chunk_v1(N, H, B, T, control, true) > 0 ? 1 : 0;
}

int phx_debug_forward_kernel_m(int N, int H, int B, int T, int control, int method)
{
    if (fwd3_chunk(N, H, B, T, control, method) > 0) return 3;
    if (fwd3c_chunk(N, H, B, T, control, method) > 0) return 4;   // third generation, chunked hidden layer (H > 48)
    return pick_chunk_v1(N, H, B, T, control, false) > 0 ? 1 : 0;
}

int phx_debug_solve_launches(int op, int N, int H, int B, int T, int control, int method)
{
    if (N <= 0 || H <= 0 || B <= 0 || T < 0 || (op != PHX_OP_ODEINT && op != PHX_OP_ADJOINT)) return 0;
    int chunk = 0;
    if (op == PHX_OP_ADJOINT) {
        if ((chunk = adj3_chunk(N, H, B, T, control, method)) <= 0 && (chunk = adj3c_chunk(N, H, B, T, control, method)) <= 0 &&
            (chunk = adj2_chunk(N, H, B, T, control)) <= 0)
            chunk = pick_chunk_v1(N, H, B, T, control, true);
    } else {
        if ((chunk = fwd3_chunk(N, H, B, T, control, method)) <= 0 && (chunk = fwd3c_chunk(N, H, B, T, control, method)) <= 0)
            chunk = pick_chunk_v1(N, H, B, T, control, false);
    }
    if (chunk <= 0) return 1;   // the VALU engine takes any batch in one launch
    return (B + chunk - 1) / chunk;
}

size_t phx_workspace_bytes(int op, int N, int H, int B, int T)
{
    if (N <= 0 || H <= 0 || B <= 0 || T < 0) return 0;
    const Dims d = make_dims(N, H, B, T, PHX_CTRL_PER_TRAJECTORY);
    size_t need = make_layout(d, op).total;
    if (op == PHX_OP_ODEINT) {
        need = std::max(need, fwd3_workspace_bytes(N, H, B, T));
        need = std::max(need, fwd3c_workspace_bytes(N, H, B, T));
        D1 d1;
        for (int ctl = 0; ctl < 2; ++ctl) {
            const int bc = pick_chunk_v1(N, H, B, T, ctl, false);
            if (bc > 0 && plan_v1(N, H, bc, T, ctl, NVEC_FWD, 2, 0, &d1))
                need = std::max(need, make_layout1(d1, 2 * d1.HT, false).total);
        }
    }
    if (op == PHX_OP_RHS_FORWARD) {
        PlanBatch pb;
        if (plan_batch(N, H, B, &pb)) need = std::max(need, pb.total_fwd);   // (whether the chain runs depends on the call's images)
        PlanEval pe;
        if (plan_eval(N, H, B, 8, &pe)) need = std::max(need, make_layout_eval(pe, false).total);
    }
    if (op == PHX_OP_RHS_VJP) {
        PlanBatch pb;
        if (plan_batch(N, H, B, &pb)) need = std::max(need, std::max(pb.total, pb.total_vjp));
        PlanEval pe;
        for (int nbc = (H <= 48 ? 2 : EVAL_NBC_HT8); nbc <= (H <= 48 ? EVAL_NBC_HT3_MAX : EVAL_NBC_HT8); ++nbc)
            if (plan_eval(N, H, B, nbc, &pe)) need = std::max(need, make_layout_eval(pe, true).total);
    }
    if (op == PHX_OP_ADJOINT) {
        need = std::max(need, adj2_workspace_bytes(N, H, B, T));
        need = std::max(need, adj3_workspace_bytes(N, H, B, T));
        need = std::max(need, adj3c_workspace_bytes(N, H, B, T));
        D1 d1;
        for (int ctl = 0; ctl < 2; ++ctl) {
            const int bc = pick_chunk_v1(N, H, B, T, ctl, true);
            if (bc > 0 && plan_v1(N, H, bc, T, ctl, NVEC_ADJ, 4, ADJ_LDS_EXTRA, &d1, 40, ADJ_NW_CAP))
                need = std::max(need, make_layout1(d1, 4 * d1.HT, true, 7).total);
        }
    }
    return need;
}

size_t phx_weight_image_bytes(int N, int H)
{
    if (N <= 0 || H <= 0 || H > 256 || force_v0()) return 0;
    const int HC = (H + 127) / 128, Hc = (H + HC - 1) / HC, HT = solve_ht(HC, Hc);
    return (size_t)((N + 31) / 32) * HC * blk_floats_ch(HT, Hc) * 4;
}

int phx_pack_weight_images(const phx_params *p, void *wimg, void *stream)
{
    if (bad_params(p) || !wimg || phx_weight_image_bytes(p->N, p->H) == 0) return PHX_ERR_BAD_ARG;
    const int HC = (p->H + 127) / 128, Hc = (p->H + HC - 1) / HC, HT = solve_ht(HC, Hc);
    hipLaunchKernelGGL(k1_pack_images, dim3(((p->N + 31) / 32) * HC), dim3(256), 0, (hipStream_t)stream, to_net(p),
                       (float *)wimg, HT, HC, Hc, blk_floats_ch(HT, Hc));
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

int phx_layout_params(const float *Ws, const float *Wp, const float *Wa, const float *g, int N, int H, float *WaT_out,
                      void *wimg_out, void *stream)
{
    if (!Ws || !Wp || !Wa || !g || !WaT_out || N <= 0 || H <= 0) return PHX_ERR_BAD_ARG;
    if (wimg_out && phx_weight_image_bytes(N, H) == 0) return PHX_ERR_BAD_ARG;
    const int HC = (H + 127) / 128, Hc = (H + HC - 1) / HC, HT = solve_ht(HC, Hc);
    hipLaunchKernelGGL(k1_layout_params, dim3((N + 31) / 32, 2 + (2 * H + 63) / 64), dim3(256), 0, (hipStream_t)stream, Ws, Wp, Wa, g, N, H,
                       WaT_out, (float *)wimg_out, HT, HC, Hc, blk_floats_ch(HT, Hc));
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

size_t phx_odeint_calls_workspace_bytes(int N, int H, int B, int T, int calls)
{
    if (N <= 0 || H <= 0 || B <= 0 || T < 0 || calls < 1 || B % calls != 0) return 0;
    if (calls == 1) return phx_workspace_bytes(PHX_OP_ODEINT, N, H, B, T);
    const int n = pick_calls_v1(N, H, B, T, calls);
    D1 d1;
    if (n < 1 || !plan_v1(N, H, n * (B / calls), T, PHX_CTRL_SHARED, NVEC_FWD, 2, 0, &d1, 0, 8, n)) return 0;
    return make_layout1(d1, 2 * d1.HT, false).total;
}

int phx_rhs_forward(const phx_params *p, const float *y, float *out, int B, int prior_only, void *workspace,
                    size_t workspace_bytes, void *stream)
{
    if (bad_params(p) || !y || !out || B <= 0 || !workspace) return PHX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    {   // large batches (the prior branch): exchange-free A -> R -> D chain (phx_mfma_batch.inc)
        PlanBatch pb;
        if (plan_batch(p->N, p->H, B, &pb) && B >= batch_fwd_min_rows(batch_net(pb, p).wimg != nullptr)) {
            if (workspace_bytes < pb.total_fwd) return PHX_ERR_WORKSPACE;
            return pb.d.HT == 3 ? launch_batch_forward<3>(pb, p, y, out, prior_only, (char *)workspace, st)
                                : pb.d.HT == 7 ? launch_batch_forward<7>(pb, p, y, out, prior_only, (char *)workspace, st)
                                               : launch_batch_forward<8>(pb, p, y, out, prior_only, (char *)workspace, st);
        }
    }
    {   // v1: MFMA, weights staged once for all passes over the batch
        PlanEval pe;
        if (plan_eval(p->N, p->H, B, 8, &pe)) {
            const LayoutE L1 = make_layout_eval(pe, false);
            if (workspace_bytes < L1.total) return PHX_ERR_WORKSPACE;
            char *base = (char *)workspace;
            W1 w1{};
            w1.cnt = (unsigned long long *)(base + L1.cnt);
            w1.abort_flag = (unsigned int *)(base + L1.cnt + 2048);
            w1.part = (unsigned long long *)(base + L1.part);
            w1.zbuf = (unsigned long long *)(base + L1.zbuf);
            // counters + granule buffers are contiguous: one fill
            if (hipMemsetAsync(w1.cnt, 0, L1.part - L1.cnt + L1.xbytes, st) != hipSuccess) return PHX_ERR_LAUNCH;
            const dim3 grid1(pe.d.TG * pe.d.G), blk1(64 * pe.d.NW);
            if (pe.d.HT == 3) {
                if (!set_lds_resident(k1_eval_fwd<3, 256>, pe.lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_eval_fwd<3, 256>), grid1, blk1, pe.lds, st, to_net(p), pe.d, w1, y, out, prior_only,
                                   pe.npass);
            } else {
                if (!set_lds_resident(k1_eval_fwd<8, 256>, pe.lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_eval_fwd<8, 256>), grid1, blk1, pe.lds, st, to_net(p), pe.d, w1, y, out, prior_only,
                                   pe.npass);
            }
            return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
        }
    }
    const Dims d = make_dims(p->N, p->H, B, 0, PHX_CTRL_PER_TRAJECTORY);
    const Layout L = make_layout(d, PHX_OP_RHS_FORWARD);
    if (workspace_bytes < L.total) return PHX_ERR_WORKSPACE;
    const WS w = make_ws(workspace, L, d);
    const int grid = grid_for(d.items);
    if (grid <= 0) return PHX_ERR_LAUNCH;
    if (!fits_resident(reinterpret_cast<const void *>(k_eval<false>), NT, 0, grid)) return PHX_ERR_LAUNCH;   // grid barriers inside
    if (hipMemsetAsync(w.sync, 0, sizeof(SyncBlock), st) != hipSuccess) return PHX_ERR_LAUNCH;
    hipLaunchKernelGGL(k_eval<false>, dim3(grid), dim3(NT), 0, st, to_net(p), d, w, y, (const float *)nullptr, out,
                       (float *)nullptr, prior_only, 0, (int *)nullptr);
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

int phx_rhs_vjp(const phx_params *p, const float *y, const float *cot, float *vjp_y, const phx_grads *grads,
                float *f_out, int B, int prior_only, void *workspace, size_t workspace_bytes, void *stream)
{
    if (bad_params(p) || !y || !cot || B <= 0 || !workspace) return PHX_ERR_BAD_ARG;
    if (grads && (!grads->Ws || !grads->bs || !grads->Wp || !grads->bp || (!grads->WaT && !grads->Wa) || !grads->g))
        return PHX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    if (prior_only && grads && !vjp_y && !f_out) {   // the prior branch's backward: MFMA parameter gradients
        PlanBatch pb;
        if (plan_batch(p->N, p->H, B, &pb)) {       // exchange-free three-kernel path (phx_mfma_batch.inc)
            if (workspace_bytes < pb.total) return PHX_ERR_WORKSPACE;
            return pb.d.HT == 3 ? launch_batch_pgrad<3>(pb, p, y, cot, grads, (char *)workspace, st)
                                : pb.d.HT == 7 ? launch_batch_pgrad<7>(pb, p, y, cot, grads, (char *)workspace, st)
                                               : launch_batch_pgrad<8>(pb, p, y, cot, grads, (char *)workspace, st);
        }
        PlanEval pe;
        const int nbc3 = eval_nbc_ht3();
        if (plan_eval(p->N, p->H, B, p->H <= 48 ? nbc3 : EVAL_NBC_HT8, &pe)) {
            const LayoutE L1 = make_layout_eval(pe, true);
            if (workspace_bytes < L1.total) return PHX_ERR_WORKSPACE;
            char *base = (char *)workspace;
            W1 w1{};
            w1.cnt = (unsigned long long *)(base + L1.cnt);
            w1.abort_flag = (unsigned int *)(base + L1.cnt + 2048);
            w1.part = (unsigned long long *)(base + L1.part);
            w1.zbuf = (unsigned long long *)(base + L1.zbuf);
            w1.dtheta = (float *)(base + L1.dtheta);
            const long long PP = (long long)align_up((size_t)4 * p->H * p->N + p->N + 2 * p->H, 4);
            // counters + granule buffers are contiguous: one fill
            if (hipMemsetAsync(w1.cnt, 0, L1.part - L1.cnt + L1.xbytes, st) != hipSuccess) return PHX_ERR_LAUNCH;
            const dim3 grid1(pe.d.TG * pe.d.G), blk1(64 * pe.d.NW);
            if (pe.d.HT == 3 && nbc3 == 2) {
                if (!set_lds_resident(k1_eval_pgrad<3, 2>, pe.lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_eval_pgrad<3, 2>), grid1, blk1, pe.lds, st, to_net(p), pe.d, w1, y, cot, pe.npass, PP);
            } else if (pe.d.HT == 3 && nbc3 == 3) {
                if (!set_lds_resident(k1_eval_pgrad<3, 3>, pe.lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_eval_pgrad<3, 3>), grid1, blk1, pe.lds, st, to_net(p), pe.d, w1, y, cot, pe.npass, PP);
            } else if (pe.d.HT == 3) {
                if (!set_lds_resident(k1_eval_pgrad<3, 4>, pe.lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_eval_pgrad<3, 4>), grid1, blk1, pe.lds, st, to_net(p), pe.d, w1, y, cot, pe.npass, PP);
            } else {
                if (!set_lds_resident(k1_eval_pgrad<8, EVAL_NBC_HT8>, pe.lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_eval_pgrad<8, EVAL_NBC_HT8>), grid1, blk1, pe.lds, st, to_net(p), pe.d, w1, y, cot,
                                   pe.npass, PP);
            }
            if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
            return launch_reduce_grads(w1.dtheta, pe.d.TG * pe.d.NW, PP, p->N, p->H, grads, grads->overwrite, st)
                       ? PHX_OK : PHX_ERR_LAUNCH;
        }
    }
    {   // everything else (dL/dy wanted, f wanted, or the full RHS): MFMA kernel chain A -> R -> E (+ C), one pass per
        // hidden chunk (H > 128: two chunks whose contributions add up)
        PlanBatch pb;
        if (plan_batch(p->N, p->H, B, &pb) && pb.d.NB <= 6) {
            if (workspace_bytes < pb.total_vjp) return PHX_ERR_WORKSPACE;
            return pb.d.HT == 3 ? launch_batch_vjp<3>(pb, p, y, cot, vjp_y, grads, f_out, prior_only, (char *)workspace, st)
                                : pb.d.HT == 7 ? launch_batch_vjp<7>(pb, p, y, cot, vjp_y, grads, f_out, prior_only, (char *)workspace, st)
                                               : launch_batch_vjp<8>(pb, p, y, cot, vjp_y, grads, f_out, prior_only, (char *)workspace, st);
        }
    }
    const Dims d = make_dims(p->N, p->H, B, 0, PHX_CTRL_PER_TRAJECTORY);
    const Layout L = make_layout(d, PHX_OP_RHS_VJP);
    if (workspace_bytes < L.total) return PHX_ERR_WORKSPACE;
    const WS w = make_ws(workspace, L, d);
    const int grid = grid_for(d.items);
    if (grid <= 0) return PHX_ERR_LAUNCH;
    if (!fits_resident(reinterpret_cast<const void *>(k_eval<true>), NT, 0, grid)) return PHX_ERR_LAUNCH;   // grid barriers inside
    if (hipMemsetAsync(w.sync, 0, sizeof(SyncBlock), st) != hipSuccess) return PHX_ERR_LAUNCH;
    if (grads && hipMemsetAsync(w.dtheta, 0, sizeof(float) * (size_t)d.PP * d.GB, st) != hipSuccess)
        return PHX_ERR_LAUNCH;
    hipLaunchKernelGGL(k_eval<true>, dim3(grid), dim3(NT), 0, st, to_net(p), d, w, y, cot, f_out, vjp_y, prior_only,
                       grads ? 1 : 0, (int *)nullptr);
    if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
    if (grads) return launch_reduce(d, w, grads, st);
    return PHX_OK;
}

int phx_odeint(const phx_params *p, const float *y0_all, const double *t_all, int B, int T, const phx_solve_opts *o,
               float *sol_all, int *status_all, int *nfe_all, int *nsteps_all, void *workspace, size_t workspace_bytes,
               void *stream)
{
    if (bad_params(p) || !y0_all || !t_all || !o || !sol_all || !status_all || !nfe_all || !nsteps_all || B <= 0 ||
        T < 1 || !workspace)
        return PHX_ERR_BAD_ARG;
    if (o->method < PHX_EULER || o->method > PHX_DOPRI5) return PHX_ERR_BAD_ARG;
    if (o->control == PHX_CTRL_SHARED && o->t_per_sample) return PHX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    SolveCfg cfg;
    cfg.method = o->method; cfg.control = o->control; cfg.t_per_sample = o->t_per_sample; cfg.t_is_f32 = o->t_is_f32;
    cfg.rtol = (float)o->rtol; cfg.atol = (float)o->atol;
    cfg.max_steps = o->max_num_steps > 0 ? o->max_num_steps : 2147483647LL;
    // several calls in one batch (analysis callers): every call keeps its own shared controller, a launch takes as many
    // calls as there are batch groups to run them (MFMA kernels only)
    const int calls = o->calls > 1 ? o->calls : 1;
    int Bcall = 0;
    if (calls > 1) {
        if (o->control != PHX_CTRL_SHARED || B % calls != 0) return PHX_ERR_BAD_ARG;
        Bcall = B / calls;
    }
    // third-generation forward kernel (dopri5, narrow hidden layer; phx_fwd3.hip)
    if (!Bcall && fwd3_chunk(p->N, p->H, B, T, o->control, o->method) > 0)
        return fwd3_run(p, y0_all, t_all, B, T, o, sol_all, status_all, nfe_all, nsteps_all, workspace, workspace_bytes, st);
    // ... and its hidden-chunked form for wide hidden layers (H > 48: the yeast and B-cell shapes; phx_fwd3c.hip)
    if (!Bcall && fwd3c_chunk(p->N, p->H, B, T, o->control, o->method) > 0)
        return fwd3c_run(p, y0_all, t_all, B, T, o, sol_all, status_all, nfe_all, nsteps_all, workspace, workspace_bytes, st);
    // v1: MFMA kernels with LDS-resident weights, when the shape fits (large batches: in chunks)
    const int chunk_f = Bcall ? pick_calls_v1(p->N, p->H, B, T, calls) * Bcall : pick_chunk_v1(p->N, p->H, B, T, o->control, false);
    if (Bcall && chunk_f == 0) return PHX_ERR_BAD_ARG;
    for (int b0 = 0; chunk_f > 0 && b0 < B; b0 += chunk_f) {
        D1 d1;
        const int bc = std::min(chunk_f, B - b0);
        if (!plan_v1(p->N, p->H, bc, T, o->control, NVEC_FWD, 2, 0, &d1, 0, 8, Bcall ? bc / Bcall : 1)) return PHX_ERR_BAD_ARG;
        d1.BN = (long long)B * p->N;   // time stride of the caller's [T,B,N] arrays
        {
            const float *y0 = y0_all + (long long)b0 * p->N;
            const double *t = !o->t_per_sample ? t_all   // rows of b0 onward; the buffer holds floats when t_is_f32 == 2
                              : reinterpret_cast<const double *>(reinterpret_cast<const char *>(t_all) +
                                                                 (size_t)b0 * T * (o->t_is_f32 == 2 ? 4 : 8));
            float *sol = sol_all + (long long)b0 * p->N;
            int *status = status_all + b0, *nfe = nfe_all + b0, *nsteps = nsteps_all + b0;
            const Layout1 L1 = make_layout1(d1, 2 * d1.HT, false);
            if (workspace_bytes < L1.total) return PHX_ERR_WORKSPACE;
            W1 w1 = make_w1(workspace, L1);
            const size_t lds = lds_bytes_v1(d1, 0);
            // counters + granule buffers are contiguous: one fill
            if (hipMemsetAsync(w1.cnt, 0, L1.part - L1.cnt + L1.xbytes, st) != hipSuccess) return PHX_ERR_LAUNCH;
            const dim3 grid1(d1.TG * d1.G), blk1(64 * d1.NW);
            if (p->wimg) w1.wimg = (const float *)p->wimg;   // packed once by the caller for these parameter values
            else
                hipLaunchKernelGGL(k1_pack_images, dim3(d1.nblk * d1.HC), dim3(256), 0, st, to_net(p), (float *)w1.wimg, d1.HT,
                                   d1.HC, d1.Hc, blk_floats_ch(d1.HT, d1.Hc));
            ev_begin(st);
            if (d1.HC > 1 && d1.HT == 7) {
                if (!set_lds_resident(k1_solve_fwd<7, 256, true>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_fwd<7, 256, true>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, y0, t,
                                   sol, status, nfe, nsteps);
            } else if (d1.HC > 1) {
                if (!set_lds_resident(k1_solve_fwd<8, 256, true>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_fwd<8, 256, true>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, y0, t,
                                   sol, status, nfe, nsteps);
            } else if (d1.HT == 3 && d1.NW == 8) {
                if (!set_lds_resident(k1_solve_fwd<3, 512, false>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_fwd<3, 512, false>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, y0, t,
                                   sol, status, nfe, nsteps);
            } else if (d1.HT == 3) {
                if (!set_lds_resident(k1_solve_fwd<3, 256, false>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_fwd<3, 256, false>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, y0, t,
                                   sol, status, nfe, nsteps);
            } else {
                if (!set_lds_resident(k1_solve_fwd<8, 256, false>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_fwd<8, 256, false>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, y0, t,
                                   sol, status, nfe, nsteps);
            }
            ev_end(st);
            if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
        }
    }
    if (chunk_f > 0) return PHX_OK;
    const float *y0 = y0_all;
    const double *t = t_all;
    float *sol = sol_all;
    int *status = status_all, *nfe = nfe_all, *nsteps = nsteps_all;
    const Dims d = make_dims(p->N, p->H, B, T, o->control);
    const Layout L = make_layout(d, PHX_OP_ODEINT);
    if (workspace_bytes < L.total) return PHX_ERR_WORKSPACE;
    const WS w = make_ws(workspace, L, d);
    const int grid = grid_for(d.items);
    if (grid <= 0) return PHX_ERR_LAUNCH;
    if (!fits_resident(reinterpret_cast<const void *>(k_solve_fwd), NT, 0, grid)) return PHX_ERR_LAUNCH;   // grid barriers inside
    if (hipMemsetAsync(w.sync, 0, sizeof(SyncBlock), st) != hipSuccess) return PHX_ERR_LAUNCH;
    ev_begin(st);
    hipLaunchKernelGGL(k_solve_fwd, dim3(grid), dim3(NT), 0, st, to_net(p), d, w, cfg, y0, t, sol, status, nfe,
                       nsteps);
    ev_end(st);
    return hipGetLastError() == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
}

int phx_odeint_adjoint_backward(const phx_params *p, const double *t_all, int B, int T, const phx_solve_opts *o,
                                const float *y_saved_all, const float *grad_y_all, float *adj_y0_all,
                                const phx_grads *grads, int *status_all, int *nfe_all, int *nsteps_all,
                                void *workspace, size_t workspace_bytes, void *stream)
{
    if (bad_params(p) || !t_all || !o || !y_saved_all || !grad_y_all || !adj_y0_all || !status_all || !nfe_all ||
        !nsteps_all || B <= 0 || T < 1 || !workspace)
        return PHX_ERR_BAD_ARG;
    if (grads && (!grads->Ws || !grads->bs || !grads->Wp || !grads->bp || (!grads->WaT && !grads->Wa) || !grads->g))
        return PHX_ERR_BAD_ARG;
    if (o->method < PHX_EULER || o->method > PHX_DOPRI5) return PHX_ERR_BAD_ARG;
    if (o->control == PHX_CTRL_SHARED && o->t_per_sample) return PHX_ERR_BAD_ARG;
    hipStream_t st = (hipStream_t)stream;
    SolveCfg cfg;
    cfg.method = o->method; cfg.control = o->control; cfg.t_per_sample = o->t_per_sample; cfg.t_is_f32 = o->t_is_f32;
    cfg.rtol = (float)o->rtol; cfg.atol = (float)o->atol;
    cfg.max_steps = o->max_num_steps > 0 ? o->max_num_steps : 2147483647LL;
    // third kernel (dopri5, narrow hidden layer exchanged among many gene tiles: the breast-cancer shape; phx_adj3.hip)
    if (adj3_chunk(p->N, p->H, B, T, o->control, o->method) > 0)
        return adj3_run(p, t_all, B, T, o, y_saved_all, grad_y_all, adj_y0_all, grads, status_all, nfe_all, nsteps_all,
                        workspace, workspace_bytes, st);
    // ... and its hidden-chunked form for wide hidden layers (H > 48: the yeast and B-cell shapes; phx_adj3c.hip)
    if (adj3c_chunk(p->N, p->H, B, T, o->control, o->method) > 0)
        return adj3c_run(p, t_all, B, T, o, y_saved_all, grad_y_all, adj_y0_all, grads, status_all, nfe_all, nsteps_all,
                         workspace, workspace_bytes, st);
    // second-generation MFMA kernel (wave pairs, fused sweeps): hidden layers that stay LDS resident (phx_adj2.hip)
    if (adj2_chunk(p->N, p->H, B, T, o->control) > 0)
        return adj2_run(p, t_all, B, T, o, y_saved_all, grad_y_all, adj_y0_all, grads, status_all, nfe_all, nsteps_all,
                        workspace, workspace_bytes, st);
    // v1: MFMA kernels (large batches: in chunks; every chunk's partials are added into `grads`)
    const int chunk_a = pick_chunk_v1(p->N, p->H, B, T, o->control, true);
    for (int b0 = 0; chunk_a > 0 && b0 < B; b0 += chunk_a) {
        D1 d1;
        const int bc = std::min(chunk_a, B - b0);
        if (!plan_v1(p->N, p->H, bc, T, o->control, NVEC_ADJ, 4, ADJ_LDS_EXTRA, &d1, 40, ADJ_NW_CAP)) return PHX_ERR_BAD_ARG;
        d1.BN = (long long)B * p->N;
        {
            const double *t = !o->t_per_sample ? t_all   // rows of b0 onward; the buffer holds floats when t_is_f32 == 2
                              : reinterpret_cast<const double *>(reinterpret_cast<const char *>(t_all) +
                                                                 (size_t)b0 * T * (o->t_is_f32 == 2 ? 4 : 8));
            const float *y_saved = y_saved_all + (long long)b0 * p->N, *grad_y = grad_y_all + (long long)b0 * p->N;
            float *adj_y0 = adj_y0_all + (long long)b0 * p->N;
            int *status = status_all + b0, *nfe = nfe_all + b0, *nsteps = nsteps_all + b0;
            const Layout1 L1 = make_layout1(d1, 4 * d1.HT, true, 7);
            if (workspace_bytes < L1.total) return PHX_ERR_WORKSPACE;
            W1 w1 = make_w1(workspace, L1);
            const size_t lds = lds_bytes_v1(d1, ADJ_LDS_EXTRA) + (size_t)40 * d1.Bt;
            const long long PP = (long long)align_up((size_t)4 * p->H * p->N + p->N + 2 * p->H, 4);
            // counters + granule buffers are contiguous: one fill
            if (hipMemsetAsync(w1.cnt, 0, L1.part - L1.cnt + L1.xbytes, st) != hipSuccess) return PHX_ERR_LAUNCH;
            // the quadrature pass first-touches every element of every partial (plain stores) when T >= 2
            if (grads && T < 2 && hipMemsetAsync(w1.dtheta, 0, sizeof(float) * (size_t)PP * d1.TG * d1.NW, st) != hipSuccess)
                return PHX_ERR_LAUNCH;
            const dim3 grid1(d1.TG * d1.G), blk1(64 * d1.NW);
            if (p->wimg) w1.wimg = (const float *)p->wimg;   // packed once by the caller for these parameter values
            else
                hipLaunchKernelGGL(k1_pack_images, dim3(d1.nblk * d1.HC), dim3(256), 0, st, to_net(p), (float *)w1.wimg, d1.HT,
                                   d1.HC, d1.Hc, blk_floats_ch(d1.HT, d1.Hc));
            ev_begin(st);
            if (d1.HC > 1 && d1.HT == 7) {
                if (!set_lds_resident(k1_solve_adj<7, 256, true>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_adj<7, 256, true>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, t,
                                   y_saved, grad_y, adj_y0, status, nfe, nsteps, grads ? 1 : 0, PP);
            } else if (d1.HC > 1) {
                if (!set_lds_resident(k1_solve_adj<8, 256, true>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_adj<8, 256, true>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, t,
                                   y_saved, grad_y, adj_y0, status, nfe, nsteps, grads ? 1 : 0, PP);
            } else if (d1.HT == 3) {   // NW <= ADJ_NW_CAP = 4 (the kernel's LDS combine layout relies on it)
                if (!set_lds_resident(k1_solve_adj<3, 256, false>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_adj<3, 256, false>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, t,
                                   y_saved, grad_y, adj_y0, status, nfe, nsteps, grads ? 1 : 0, PP);
            } else {
                if (!set_lds_resident(k1_solve_adj<8, 256, false>, lds, (int)blk1.x, (int)grid1.x)) return PHX_ERR_LAUNCH;
                hipLaunchKernelGGL((k1_solve_adj<8, 256, false>), grid1, blk1, lds, st, to_net(p), d1, w1, cfg, t,
                                   y_saved, grad_y, adj_y0, status, nfe, nsteps, grads ? 1 : 0, PP);
            }
            ev_end(st);
            if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
            if (grads) {
                // partials of waves that own tiles (helper waves of a single small group write none); later chunks of a
                // large batch add
                if (!launch_reduce_grads(w1.dtheta, d1.TG == 1 ? (d1.ntg + d1.TPW - 1) / d1.TPW : d1.TG * d1.NW, PP, p->N,
                                         p->H, grads, (grads->overwrite && b0 == 0) ? 1 : 0, st))
                    return PHX_ERR_LAUNCH;
            }
        }
    }
    if (chunk_a > 0) return PHX_OK;
    const double *t = t_all;
    const float *y_saved = y_saved_all, *grad_y = grad_y_all;
    float *adj_y0 = adj_y0_all;
    int *status = status_all, *nfe = nfe_all, *nsteps = nsteps_all;
    const Dims d = make_dims(p->N, p->H, B, T, o->control);
    const Layout L = make_layout(d, PHX_OP_ADJOINT);
    if (workspace_bytes < L.total) return PHX_ERR_WORKSPACE;
    const WS w = make_ws(workspace, L, d);
    const int grid = grid_for(d.items);
    if (grid <= 0) return PHX_ERR_LAUNCH;
    if (!fits_resident(reinterpret_cast<const void *>(k_solve_adj), NT, 0, grid)) return PHX_ERR_LAUNCH;   // grid barriers inside
    if (hipMemsetAsync(w.sync, 0, sizeof(SyncBlock), st) != hipSuccess) return PHX_ERR_LAUNCH;
    if (hipMemsetAsync(w.dtheta, 0, sizeof(float) * (size_t)d.PP * d.GB, st) != hipSuccess) return PHX_ERR_LAUNCH;
    ev_begin(st);
    hipLaunchKernelGGL(k_solve_adj, dim3(grid), dim3(NT), 0, st, to_net(p), d, w, cfg, t, y_saved, grad_y, adj_y0,
                       status, nfe, nsteps, grads ? 1 : 0);
    ev_end(st);
    if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
    if (grads) return launch_reduce(d, w, grads, st);
    return PHX_OK;
}

}  // extern "C"
