// phx_adj3.hip -- translation unit of the third backward kernel (k1_solve_adj3, phx_mfma_adj3.inc): launch planning,
// workspace layout and the host entry points the C ABI (phx_engine.hip) calls.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "phx_solver.hpp"
#include "phx_host.hpp"

using namespace phxh;

#include "phx_mfma_common.inc"
#include "phx_mfma_v3common.inc"
#include "phx_mfma_adj3.inc"

namespace {

int adj3_mode()   // -1: disabled (another kernel forced), 1: forced for every shape it supports, 0: by shape
{
    if (force_v0()) return -1;
    const char *e = getenv("PHX_ADJ");
    if (!e) return 0;
    if (strcmp(e, "v3") == 0) return 1;
    if (strcmp(e, "v1") == 0 || strcmp(e, "v2") == 0) return -1;
    return 0;
}

// picks (NW, TPW, NB) like plan_v1 does for the first backward kernel: one wave per SIMD (the augmented sweeps need the
// 512-register budget), the smallest gene tile that keeps TG x G workgroups resident, the smallest batch group
bool plan_adj3(int N, int H, int B, int T, int control, int method, D1 *out)
{
    const int cus = num_cus();
    const int mode = adj3_mode();
    if (cus <= 0 || mode < 0 || method != PHX_DOPRI5 || H > 48) return false;
    const int HT = 3;
    const size_t blkbytes = (size_t)blk_floats_ch(HT, H) * 4;
    const int nblk = (N + 31) / 32, ntt = (B + 15) / 16;
    long long best_cost = -1;
    D1 best{};
    for (int NW = 4; NW >= 1; NW >>= 1)
        for (int TPW = 1; TPW <= 4; TPW <<= 1) {
            const int slots = NW * TPW, TG = (ntt + slots - 1) / slots;
            const int ntg = TG == 1 ? std::min(slots, ntt) : slots, Bt = 16 * ntg;
            const bool helpers = ntg < slots;
            if (control == PHX_CTRL_SHARED && TG != 1) continue;
            // (+ the LDS of the block-split combine where the batch is that small: v3_split_parts)
            const size_t cb = ((ctl3_bytes(Bt, ntg) + 15) & ~(size_t)15) +
                              (v3_split_parts(TG, NW, TPW, ntg) > 1 ? v3_comb_bytes(NW, 4 * HT) : 0);
            if (cb + blkbytes > LDS_BUDGET) continue;
            const int NBmax = (int)std::min<size_t>((LDS_BUDGET - cb) / blkbytes, 8);
            const char *enb = getenv("PHX_V3_NB");   // experiment: smallest gene tile to consider
            for (int NB = enb ? std::max(1, atoi(enb)) : 1; NB <= NBmax; ++NB) {
                const int G = (nblk + NB - 1) / NB;
                if ((long long)TG * G > cus) continue;
                const long long cost = (long long)TPW * NB * 1000 + Bt / 4 - (helpers && TPW == 1 ? 50 : 0);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    best.N = N; best.H = H; best.B = B; best.T = T; best.HT = HT; best.NB = NB; best.NW = NW;
                    best.TPW = TPW; best.G = G; best.TG = TG; best.nblk = nblk; best.ntg = ntg; best.Bt = Bt;
                    best.nvec = NVEC_ADJ3; best.BN = (long long)B * N; best.HC = 1; best.Hc = H;
                    best.Bcall = 0; best.cntN = (long long)B * N;
                }
                break;  // smallest feasible NB for this (NW, TPW) is the cheapest
            }
        }
    if (best_cost < 0) return false;
    // by shape: the many-member exchange of a narrow hidden layer (the breast-cancer shape) is this kernel's; groups of
    // up to 32 gene tiles stay with the wave-pair kernel (k1_solve_adj2), which is faster there.  PHX_ADJ=v3 forces it.
    if (mode == 0 && best.G <= 32) return false;
    *out = best;
    return true;
}

int pick_chunk_adj3(int N, int H, int B, int T, int control, int method)
{
    D1 d1;
    if (plan_adj3(N, H, B, T, control, method, &d1)) return B;
    if (control != PHX_CTRL_PER_TRAJECTORY) return 0;
    for (int bc = 4096; bc >= 16; bc >>= 1)
        if (bc < B && plan_adj3(N, H, bc, T, control, method, &d1)) return bc;
    return 0;
}

// floats of one batch group's partial: accumulator-native [gene block][4 HT x 2][64] float4, then dg [N], dbs [H], dbp [H]
size_t pp_adj3(const D1 &d) { return align_up((size_t)d.nblk * (4 * d.HT * 2 * 256) + d.N + 2 * d.H, 64); }

struct Layout3 {
    size_t total, cnt, part, zbuf, part1, zbuf1, scratch, dtheta, prof, xbytes, wimg, hq;
};

Layout3 make_layout3(const D1 &d, bool grads)
{
    Layout3 L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t R = (size_t)d.ntg * 4 * d.HT * 4 + d.ntg;   // hidden rows + norm rows per group
    L.cnt = take(4096);
    L.part = take((size_t)d.TG * d.G * R * 64 * 8);
    L.zbuf = take((size_t)d.TG * R * 64 * 8);
    L.xbytes = off - L.part;                                 // header + set 0: what a fill covers (phx_mfma_v3common.inc: XSet)
    L.part1 = take((size_t)d.TG * d.G * R * 64 * 8);         // set 1: cleaned by the launch that works in set 0
    L.zbuf1 = take((size_t)d.TG * R * 64 * 8);
    L.scratch = take((size_t)d.TG * d.G * NVEC_ADJ3 * d.ntg * d.NB * 512 * 4);
    L.dtheta = take(grads ? pp_adj3(d) * 4 * d.TG : 0);
    L.prof = take((size_t)d.TG * d.G * 16 * 8);
    L.wimg = take((size_t)d.nblk * blk_floats_ch(d.HT, d.H) * 4);
    // transposed hidden rows of the seven ring slots, shared by a group's workgroups: [group][tile][7][4 HT][64] float4
    L.hq = take(grads ? (size_t)d.TG * d.ntg * 7 * 4 * d.HT * 1024 : 0);
    L.total = off;
    return L;
}

size_t lds_bytes_adj3(const D1 &d)
{
    return (size_t)blk_floats_ch(d.HT, d.H) * 4 * d.NB + ((ctl3_bytes(d.Bt, d.ntg) + 15) & ~(size_t)15) +
           (v3_split_parts(d.TG, d.NW, d.TPW, d.ntg) > 1 ? v3_comb_bytes(d.NW, 4 * d.HT) : 0);
}

}  // namespace

namespace phxh {

int adj3_chunk(int N, int H, int B, int T, int control, int method) { return pick_chunk_adj3(N, H, B, T, control, method); }

size_t adj3_workspace_bytes(int N, int H, int B, int T)
{
    size_t need = 0;
    for (int ctl = 0; ctl < 2; ++ctl) {
        D1 d1;
        const int bc = pick_chunk_adj3(N, H, B, T, ctl, PHX_DOPRI5);
        if (bc > 0 && plan_adj3(N, H, bc, T, ctl, PHX_DOPRI5, &d1)) need = std::max(need, make_layout3(d1, true).total);
    }
    return need;
}

int adj3_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6)
{
    D1 d1;
    if (!plan_adj3(N, H, B, T, control, PHX_DOPRI5, &d1)) return PHX_ERR_BAD_ARG;
    *offset = make_layout3(d1, true).prof;
    *n_workgroups = d1.TG * d1.G;
    if (plan6) { plan6[0] = d1.NW; plan6[1] = d1.TPW; plan6[2] = d1.NB; plan6[3] = d1.G; plan6[4] = d1.TG; plan6[5] = d1.HT; }
    return PHX_OK;
}

int adj3_run(const phx_params *p, const double *t_all, int B, int T, const phx_solve_opts *o, const float *y_saved_all,
             const float *grad_y_all, float *adj_y0_all, const phx_grads *grads, int *status_all, int *nfe_all,
             int *nsteps_all, void *workspace, size_t workspace_bytes, hipStream_t st)
{
    SolveCfg cfg;
    cfg.method = o->method; cfg.control = o->control; cfg.t_per_sample = o->t_per_sample; cfg.t_is_f32 = o->t_is_f32;
    cfg.rtol = (float)o->rtol; cfg.atol = (float)o->atol;
    cfg.max_steps = o->max_num_steps > 0 ? o->max_num_steps : 2147483647LL;
    const int chunk = pick_chunk_adj3(p->N, p->H, B, T, o->control, o->method);
    if (chunk <= 0) return PHX_ERR_BAD_ARG;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        D1 d1;
        const int bc = std::min(chunk, B - b0);
        if (!plan_adj3(p->N, p->H, bc, T, o->control, o->method, &d1)) return PHX_ERR_BAD_ARG;
        d1.BN = (long long)B * p->N;   // time stride of the caller's [T,B,N] arrays
        const double *t = !o->t_per_sample ? t_all   // rows of b0 onward; the buffer holds floats when t_is_f32 == 2
                          : reinterpret_cast<const double *>(reinterpret_cast<const char *>(t_all) +
                                                             (size_t)b0 * T * (o->t_is_f32 == 2 ? 4 : 8));
        const float *y_saved = y_saved_all + (long long)b0 * p->N, *grad_y = grad_y_all + (long long)b0 * p->N;
        float *adj_y0 = adj_y0_all + (long long)b0 * p->N;
        int *status = status_all + b0, *nfe = nfe_all + b0, *nsteps = nsteps_all + b0;
        const Layout3 L = make_layout3(d1, grads != nullptr);
        if (workspace_bytes < L.total) return PHX_ERR_WORKSPACE;
        char *base = (char *)workspace;
        W1 w1{};
        w1.cnt = (unsigned long long *)(base + L.cnt);
        w1.abort_flag = (unsigned int *)(base + L.cnt + 2048);
        w1.part = (unsigned long long *)(base + L.part);
        w1.zbuf = (unsigned long long *)(base + L.zbuf);
        w1.part1 = (unsigned long long *)(base + L.part1);
        w1.zbuf1 = (unsigned long long *)(base + L.zbuf1);
        w1.scratch = (float *)(base + L.scratch);
        w1.dtheta = (float *)(base + L.dtheta);
        const char *pe = getenv("PHX_PROF");   // 1: segment timers, 2: + per-block timers of the sweeps, 3: + of the quadrature
        const int plevel = pe ? atoi(pe) : 0;
        w1.prof = plevel >= 1 ? (unsigned long long *)(base + L.prof) : nullptr;
        const int prof_flags = (plevel == 2 ? 2 : 0) | (plevel == 3 ? 4 : 0);
        w1.wimg = (const float *)(base + L.wimg);
        w1.hq = (float *)(base + L.hq);
        const size_t lds = lds_bytes_adj3(d1);
        const long long PP = (long long)pp_adj3(d1);
        const int npart = d1.TG;   // one partial per workgroup of a gene tile = per batch group (k1_solve_adj3: quad_accept)
        // header + set 0 are contiguous: one fill -- unless the caller vouches for the workspace (ws_keep: the previous call
        // on it was this one, same shape, same options) and the batch is one launch: the kernels then alternate between
        // the two sets and clean the idle one themselves
        const bool fill = !(o->ws_keep && chunk >= B);
        if (fill && hipMemsetAsync(w1.cnt, 0, L.part - L.cnt + L.xbytes, st) != hipSuccess) return PHX_ERR_LAUNCH;
        // every wave that owns a tile first-touches its whole partial (plain stores) in its first quadrature visit, or
        // zero-fills it at the end of the launch when its group never stepped; with T < 2 nobody runs either
        if (grads && T < 2 && hipMemsetAsync(w1.dtheta, 0, sizeof(float) * (size_t)PP * npart, st) != hipSuccess)
            return PHX_ERR_LAUNCH;
        const dim3 grid1(d1.TG * d1.G), blk1(64 * d1.NW);
        if (p->wimg) w1.wimg = (const float *)p->wimg;   // packed once by the caller for these parameter values
        else
            hipLaunchKernelGGL(k1_pack_images, dim3(d1.nblk), dim3(256), 0, st, to_net(p), (float *)w1.wimg, d1.HT, 1, p->H,
                               blk_floats_ch(d1.HT, p->H));
        // HALF: the last hidden tile has at most 8 live rows (H <= 40 with three tiles; rho16 in phx_mfma_v3common.inc)
        const char *eh = getenv("PHX_V3_HALF");   // diagnostic: 0 = full last tile also where half of it is padding
        const bool half = p->H <= 16 * (d1.HT - 1) + 8 && !(eh && eh[0] == '0');
        const bool split = v3_split_parts(d1.TG, d1.NW, d1.TPW, d1.ntg) > 1;   // small batch: the waves of a tile split its blocks
        const void *fn = split ? (half ? reinterpret_cast<const void *>(k1_solve_adj3<3, true, true>)
                                       : reinterpret_cast<const void *>(k1_solve_adj3<3, false, true>))
                               : (half ? reinterpret_cast<const void *>(k1_solve_adj3<3, true, false>)
                                       : reinterpret_cast<const void *>(k1_solve_adj3<3, false, false>));
        if (!set_lds_fn(fn, lds)) return PHX_ERR_LAUNCH;
        // the workgroups of a launch wait for each other's rows: refuse a grid the device cannot hold at once
        if (!fits_resident(fn, 64 * d1.NW, lds, d1.TG * d1.G)) return PHX_ERR_LAUNCH;
        ev_begin(st);
        const hipError_t lerr = launch_persistent(fn, grid1, blk1, lds, st, to_net(p), d1, w1, cfg, t, y_saved, grad_y, adj_y0,
                                                  status, nfe, nsteps, (int)((grads ? 1 : 0) | prof_flags), PP);
        ev_end(st);
        if (lerr != hipSuccess || hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
        if (grads) {
            const long long total = (long long)d1.nblk * (4 * d1.HT * 2 * 64) + p->N + 2 * p->H;
            hipLaunchKernelGGL((k3_reduce_grads<3>), dim3((unsigned int)((total + 255) / 256)), dim3(256), 0, st, w1.dtheta, npart, PP,
                               p->N, p->H, d1.nblk, grads->Ws, grads->Wp, grads->WaT, grads->g, grads->bs, grads->bp,
                               (grads->overwrite && b0 == 0) ? 1 : 0, grads->Wa);   // later chunks of a large batch add
            if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
        }
    }
    return PHX_OK;
}

}  // namespace phxh
