// phx_adj2.hip -- translation unit of the second-generation adjoint kernel (k1_solve_adj2, phx_mfma_adj2.inc):
// launch planning, workspace layout and the host entry points the C ABI (phx_engine.hip) calls.
#include <hip/hip_runtime.h>

#include <algorithm>
#include <cstdint>
#include <cstdlib>
#include <cstring>

#include "phx_solver.hpp"
#include "phx_host.hpp"

using namespace phxh;

#include "phx_mfma_common.inc"
#include "phx_mfma_adj2.inc"

namespace {

bool adj2_disabled()
{
    const char *e = getenv("PHX_ADJ");
    return force_v0() || (e && strcmp(e, "v1") == 0);
}
bool adj2_forced()
{
    const char *e = getenv("PHX_ADJ");
    return e && strcmp(e, "v2") == 0;
}

// picks (TPW, NB): minimise the per-wave MFMA work TPW*NB subject to LDS and residency (one workgroup per CU)
bool plan_adj2(int N, int H, int B, int T, int control, D1 *out)
{
    const int cus = num_cus();
    if (cus <= 0 || H > 128 || adj2_disabled()) return false;
    const int HT = H <= 48 ? 3 : 8;
    const size_t blkbytes = (size_t)blk_floats_ch(HT, H) * 4;
    const int nblk = (N + 31) / 32, ntt = (B + 15) / 16;
    long long best_cost = -1;
    D1 best{};
    // NP pairs per workgroup: 2 (four waves, one per SIMD, 512 registers each; a wave alternates between its TPW >= 2
    // tiles, so one tile's exchange is in flight while the other tile computes) or 4 (eight waves, 256 registers)
    // Measured on MI355X (profiles/r2_adjoint_variants.txt): wide hidden layers (HT = 8) want the 512-register
    // form (NP = 2: no spills), narrow ones (HT = 3) the eight-wave form (NP = 4); PHX_ADJ2_NP overrides (diagnostic)
    int np_only = HT == 8 ? 2 : 4;
    if (const char *e = getenv("PHX_ADJ2_NP")) np_only = atoi(e);
    for (int NP = 2; NP <= 4; NP += 2)
    for (int TPW = 1; TPW <= 4; TPW <<= 1) {
        if (np_only && NP != np_only) continue;
        const int slots = NP * TPW, TG = (ntt + slots - 1) / slots;
        const int ntg = TG == 1 ? std::min(slots, ntt) : slots, Bt = 16 * ntg;
        if (control == PHX_CTRL_SHARED && TG != 1) continue;
        const size_t cb = adj2_ctl_bytes(NP, TPW);
        if (cb + blkbytes > LDS_BUDGET) continue;
        const int NBmax = (int)std::min<size_t>((LDS_BUDGET - cb) / blkbytes, 8);
        for (int NB = 1; NB <= NBmax; ++NB) {
            const int G = (nblk + NB - 1) / NB;
            if ((long long)TG * G > cus) continue;
            // per-SIMD MFMA work ~ TPW * NB * (waves per SIMD); at equal work one wave per SIMD with two tiles wins
            const long long cost = (long long)TPW * NB * (NP / 2) * 1000 + Bt / 4 + (NP == 2 && TPW >= 2 ? 0 : 40);
            if (best_cost < 0 || cost < best_cost) {
                best_cost = cost;
                best.N = N; best.H = H; best.B = B; best.T = T; best.HT = HT; best.NB = NB; best.NW = 2 * NP;
                best.TPW = TPW; best.G = G; best.TG = TG; best.nblk = nblk; best.ntg = ntg; best.Bt = Bt;
                best.nvec = NVEC_ADJ2; best.BN = (long long)B * N; best.HC = 1; best.Hc = H;
            }
            break;  // smallest feasible NB for this TPW is the cheapest
        }
    }
    if (best_cost < 0) return false;
    // Narrow hidden layer exchanged among many gene tiles (the breast-cancer shape: H = 40, 59 members per group): the
    // first-generation kernel (one wave per trajectory tile doing both halves of the augmented state: twice the MFMA
    // work per gene-block visit) is still the faster one there -- 0.73 against 0.90 ms.  PHX_ADJ=v2 forces this kernel.
    if (best.HT == 3 && best.G > 32 && !adj2_forced()) return false;
    *out = best;
    return true;
}

int pick_chunk_adj2(int N, int H, int B, int T, int control)
{
    D1 d1;
    if (plan_adj2(N, H, B, T, control, &d1)) return B;
    if (control != PHX_CTRL_PER_TRAJECTORY) return 0;
    for (int bc = 4096; bc >= 16; bc >>= 1)
        if (bc < B && plan_adj2(N, H, bc, T, control, &d1)) return bc;
    return 0;
}

struct Layout2 {
    size_t total, cnt, part, zbuf, scratch, dtheta, prof, xbytes, wimg, hq;
};

Layout2 make_layout2(const D1 &d, bool grads)
{
    Layout2 L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t RH = (size_t)d.ntg * 4 * d.HT * 4;
    const size_t R = RH + 2 * (size_t)d.ntg;    // partial rows per workgroup: hidden rows + one norm row per tile and side
    const size_t RZ = RH + 4 * (size_t)d.ntg;   // reduced rows per group: the norm rows are double buffered
    L.cnt = take(4096);
    L.part = take((size_t)d.TG * d.G * R * 64 * 8);
    L.zbuf = take((size_t)d.TG * RZ * 64 * 8);
    L.xbytes = off - L.part;                                // granule buffers are zeroed before every launch
    L.scratch = take((size_t)d.TG * d.G * NVEC_ADJ2 * d.ntg * d.NB * 512 * 4);
    const size_t PP = align_up((size_t)4 * d.H * d.N + d.N + 2 * d.H, 4);
    L.dtheta = take(grads ? PP * 4 * d.TG * (d.NW / 2) : 0);
    L.prof = take((size_t)d.TG * d.G * 16 * 8);
    L.wimg = take((size_t)d.nblk * blk_floats_ch(d.HT, d.H) * 4);
    L.hq = take(grads ? (size_t)d.TG * d.G * d.NW * d.TPW * 7 * 2 * d.HT * 256 * 4 : 0);
    L.total = off;
    return L;
}

size_t lds_bytes_adj2(const D1 &d) { return (size_t)blk_floats_ch(d.HT, d.H) * 4 * d.NB + adj2_ctl_bytes(d.NW / 2, d.TPW); }

// start delay of the second half of the wave pairs (100 MHz ticks); PHX_STAGGER_US overrides (diagnostic)
int prof_wave()
{
    if (const char *e = getenv("PHX_PROF_WAVE")) return atoi(e) & 7;
    return 0;
}
int stagger_ticks()
{
    if (const char *e = getenv("PHX_STAGGER_US")) return atoi(e) < 0 ? -1 : atoi(e) * 100;   // < 0: generic sweeps only
    return 0;
}

}  // namespace

namespace phxh {

int adj2_chunk(int N, int H, int B, int T, int control) { return pick_chunk_adj2(N, H, B, T, control); }

size_t adj2_workspace_bytes(int N, int H, int B, int T)
{
    size_t need = 0;
    for (int ctl = 0; ctl < 2; ++ctl) {
        D1 d1;
        const int bc = pick_chunk_adj2(N, H, B, T, ctl);
        if (bc > 0 && plan_adj2(N, H, bc, T, ctl, &d1)) need = std::max(need, make_layout2(d1, true).total);
    }
    return need;
}

int adj2_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6)
{
    D1 d1;
    if (!plan_adj2(N, H, B, T, control, &d1)) return PHX_ERR_BAD_ARG;
    *offset = make_layout2(d1, true).prof;
    *n_workgroups = d1.TG * d1.G;
    if (plan6) { plan6[0] = d1.NW; plan6[1] = d1.TPW; plan6[2] = d1.NB; plan6[3] = d1.G; plan6[4] = d1.TG; plan6[5] = d1.HT; }
    return PHX_OK;
}

int adj2_run(const phx_params *p, const double *t_all, int B, int T, const phx_solve_opts *o, const float *y_saved_all,
             const float *grad_y_all, float *adj_y0_all, const phx_grads *grads, int *status_all, int *nfe_all,
             int *nsteps_all, void *workspace, size_t workspace_bytes, hipStream_t st)
{
    SolveCfg cfg;
    cfg.method = o->method; cfg.control = o->control; cfg.t_per_sample = o->t_per_sample; cfg.t_is_f32 = o->t_is_f32;
    cfg.rtol = (float)o->rtol; cfg.atol = (float)o->atol;
    cfg.max_steps = o->max_num_steps > 0 ? o->max_num_steps : 2147483647LL;
    const int chunk = pick_chunk_adj2(p->N, p->H, B, T, o->control);
    if (chunk <= 0) return PHX_ERR_BAD_ARG;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        D1 d1;
        const int bc = std::min(chunk, B - b0);
        if (!plan_adj2(p->N, p->H, bc, T, o->control, &d1)) return PHX_ERR_BAD_ARG;
        d1.BN = (long long)B * p->N;   // time stride of the caller's [T,B,N] arrays
        const double *t = !o->t_per_sample ? t_all   // rows of b0 onward; the buffer holds floats when t_is_f32 == 2
                          : reinterpret_cast<const double *>(reinterpret_cast<const char *>(t_all) +
                                                             (size_t)b0 * T * (o->t_is_f32 == 2 ? 4 : 8));
        const float *y_saved = y_saved_all + (long long)b0 * p->N, *grad_y = grad_y_all + (long long)b0 * p->N;
        float *adj_y0 = adj_y0_all + (long long)b0 * p->N;
        int *status = status_all + b0, *nfe = nfe_all + b0, *nsteps = nsteps_all + b0;
        const Layout2 L = make_layout2(d1, grads != nullptr);
        if (workspace_bytes < L.total) return PHX_ERR_WORKSPACE;
        char *base = (char *)workspace;
        W1 w1{};
        w1.cnt = (unsigned long long *)(base + L.cnt);
        w1.abort_flag = (unsigned int *)(base + L.cnt + 2048);
        w1.part = (unsigned long long *)(base + L.part);
        w1.zbuf = (unsigned long long *)(base + L.zbuf);
        w1.scratch = (float *)(base + L.scratch);
        w1.dtheta = (float *)(base + L.dtheta);
        const char *pe = getenv("PHX_PROF");
        w1.prof = (pe && pe[0] == '1') ? (unsigned long long *)(base + L.prof) : nullptr;
        w1.wimg = (const float *)(base + L.wimg);
        w1.hq = (float *)(base + L.hq);
        const size_t lds = lds_bytes_adj2(d1);
        const long long PP = (long long)align_up((size_t)4 * p->H * p->N + p->N + 2 * p->H, 4);
        const int npart = d1.TG == 1 ? (d1.ntg + d1.TPW - 1) / d1.TPW : d1.TG * (d1.NW / 2);   // pairs that own tiles
        // counters + granule buffers are contiguous: one fill
        if (hipMemsetAsync(w1.cnt, 0, L.part - L.cnt + L.xbytes, st) != hipSuccess) return PHX_ERR_LAUNCH;
        // The quadrature first-touches every element of a pair's partial (plain stores) -- but only pairs that own at
        // least one real trajectory ever run it: with several groups the last group's trailing pairs can hold padding
        // tiles only (their controllers start "done"), with T < 2 nobody steps, and a pair whose trajectories all
        // failed at once never gets there either.  Every partial must read as zero then: always cleared (a few us).
        if (grads && hipMemsetAsync(w1.dtheta, 0, sizeof(float) * (size_t)PP * npart, st) != hipSuccess)
            return PHX_ERR_LAUNCH;
        const dim3 grid1(d1.TG * d1.G), blk1(64 * d1.NW);
        if (p->wimg) w1.wimg = (const float *)p->wimg;   // packed once by the caller for these parameter values
        else
            hipLaunchKernelGGL(k1_pack_images, dim3(d1.nblk), dim3(256), 0, st, to_net(p), (float *)w1.wimg, d1.HT, 1, p->H,
                               blk_floats_ch(d1.HT, p->H));
        auto launch = [&](auto kern) -> int {
            const void *fn = reinterpret_cast<const void *>(kern);
            if (!set_lds_fn(fn, lds)) return PHX_ERR_LAUNCH;
            // the workgroups of a launch wait for each other's rows: refuse a grid the device cannot hold at once
            if (!fits_resident(fn, 64 * d1.NW, lds, d1.TG * d1.G)) return PHX_ERR_LAUNCH;
            ev_begin(st);
            hipLaunchKernelGGL(kern, grid1, blk1, lds, st, to_net(p), d1, w1, cfg, t, y_saved, grad_y, adj_y0, status, nfe,
                               nsteps, grads ? 1 : 0, PP, stagger_ticks(), prof_wave());
            return PHX_OK;
        };
        int lrc;
        if (d1.HT == 3) lrc = d1.NW == 8 ? launch(k1_solve_adj2<3, 4>) : launch(k1_solve_adj2<3, 2>);
        else lrc = d1.NW == 8 ? launch(k1_solve_adj2<8, 4>) : launch(k1_solve_adj2<8, 2>);
        if (lrc != PHX_OK) return lrc;
        ev_end(st);
        if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
        if (grads) {
            // later chunks of a large batch add
            if (!launch_reduce_grads(w1.dtheta, npart, PP, p->N, p->H, grads, (grads->overwrite && b0 == 0) ? 1 : 0, st))
                return PHX_ERR_LAUNCH;
        }
    }
    return PHX_OK;
}

}  // namespace phxh
