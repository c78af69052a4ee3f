// phx_adj3c.hip -- translation unit of the third-generation backward kernel for wide hidden layers (k1_solve_adj3c,
// phx_mfma_adj3c.inc: hidden chunks of <= 48 rows): launch planning, workspace layout and the host entry points the C ABI
// (phx_engine.hip) calls.
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
#include "phx_mfma_adj3c.inc"

namespace {

bool adj3c_disabled()
{
    if (force_v0()) return true;
    const char *e = getenv("PHX_ADJ");   // another backward kernel forced
    if (e && (strcmp(e, "v1") == 0 || strcmp(e, "v2") == 0)) return true;
    const char *c = getenv("PHX_V3C");   // diagnostic: PHX_V3C=0 switches the chunked third-generation kernels off
    return c && c[0] == '0';
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// as plan_fwd3c (phx_fwd3c.hip): HC = ceil(H / 48) chunks of Hc = ceil(H / HC) rows; (NW, TPW, NB) with NB in {1, 2, 4, 8} and at
// most `maxslots` (tile, block) slots per wave
bool plan_adj3c(int N, int H, int B, int T, int control, int method, D1 *out, int *nbt_out)
{
    const int cus = num_cus();
    if (cus <= 0 || adj3c_disabled() || method != PHX_DOPRI5 || H <= 48 || H > 256) return false;
    const int HT = 3, HC = (H + 47) / 48, Hc = (H + HC - 1) / HC;
    const size_t blkbytes = (size_t)blk_floats_ch(HT, Hc) * 4;
    const int nblk = (N + 31) / 32, ntt = (B + 15) / 16;
    const int fnb = env_int("PHX_V3C_NB", 0), ftpw = env_int("PHX_V3C_TPW", 0), fres = env_int("PHX_V3C_RES", -1);
    const int fhb = env_int("PHX_V3C_HB", -1);
    // at most FOUR (tile, block) slots per wave: the eight-slot instantiation needs more than the 512 registers (3 460 spilled,
    // and this compiler fails on its <8, false> form); a batch that needs more slots runs as several launches (pick_chunk)
    const int maxslots = std::min(4, std::max(1, env_int("PHX_V3C_SLOTS", 4)));
    long long best_cost = -1;
    D1 best{};
    int best_nbt = 0;
    for (int NW = 4; NW >= 1; NW >>= 1)
        for (int TPW = 1; TPW <= 8; TPW <<= 1) {
            if (ftpw > 0 && TPW != ftpw) continue;
            const int slots = NW * TPW;
            int TG = (ntt + slots - 1) / slots;
            int ntg = TG == 1 ? std::min(slots, ntt) : slots;
            // More batch groups of FEWER trajectory tiles where the chip has room for them (one tile per wave plans, twice
            // the gene blocks -- half-block tiles -- still resident): the groups' exchanges carry fewer rows and a tile gets
            // helper waves (measured, round 5: -8 ... -32 % of a launch; 23 yeast pairs as two groups of one tile: neutral).
            // PHX_V3C_NTG forces the tiles per group (4: the plan before this rule).
            const int fntg = env_int("PHX_V3C_NTG", 0);
            if (fntg > 0 && fntg <= slots) { ntg = std::min(fntg, ntt); TG = (ntt + ntg - 1) / ntg; }
            else if (fntg == 0 && TPW == 1 && control != PHX_CTRL_SHARED) {
                // (one-tile groups may take three quarters of the chip -- 23 yeast pairs as two groups on 252 workgroups
                // measured +3 % --, two-tile groups all of it)
                for (int c = 1; c < ntg; c <<= 1) {
                    const int tgc = (ntt + c - 1) / c;
                    if ((long long)tgc * nblk * 2 * 4 <= (long long)cus * (c == 1 ? 3 : 4)) { ntg = c; TG = tgc; break; }
                }
            }
            const int Bt = 16 * ntg;
            const bool helpers = ntg < slots;
            if (control == PHX_CTRL_SHARED && TG != 1) continue;
            const size_t cb = ctl3c_bytes(Bt, ntg);
            for (int NB = 1; NB <= 8 && TPW * NB <= maxslots; NB <<= 1) {
                if (fnb > 0 && NB != fnb) continue;
                if (cb + blkbytes * NB > LDS_BUDGET) break;
                const int G = (nblk + NB - 1) / NB;
                if ((long long)TG * G > cus) continue;
                bool res = cb + blkbytes * NB * HC <= LDS_BUDGET;
                if (fres == 0) res = false;
                const long long cost = (long long)TPW * NB * 1000 + (res ? 0 : 150) + G - (helpers && TPW == 1 ? 50 : 0);
                if (best_cost < 0 || cost < best_cost) {
                    best_cost = cost;
                    best.N = N; best.H = H; best.B = B; best.T = T; best.HT = HT; best.NB = NB; best.NW = NW;
                    best.TPW = TPW; best.G = G; best.TG = TG; best.nblk = nblk; best.ntg = ntg; best.Bt = Bt;
                    best.nvec = NVEC_ADJ3C; best.BN = (long long)B * N; best.HC = HC; best.Hc = Hc;
                    best.Bcall = 0; best.cntN = (long long)B * N; best.res = res ? 1 : 0;
                    const int nslot = TPW * NB;
                    // half-block gene tiles (plan_fwd3c): one slot per wave and room for twice the workgroups
                    // (measured: -4..-37 % of a launch wherever twice the workgroups fit the chip)
                    const bool hb_fits = nslot == 1 && res && (long long)TG * G * 2 <= cus && cb + HSA_BYTES + 16 + blkbytes * HC <= LDS_BUDGET;
                    best.hb = (hb_fits && fhb != 0) ? 1 : 0;
                    best_nbt = nslot <= 1 ? 1 : 4;
                    // block split: a tile's gene blocks on the workgroup's spare waves (plan_fwd3c)
                    best.split = 1;
                    if (!best.hb && TPW == 1 && NB > 1 && ntg * NB <= NW && env_int("PHX_V3C_SPLIT", 1) != 0) {
                        const int parts = std::min(NW / ntg, NB);
                        if (hsa_offset(Bt, ntg) + spa_bytes(ntg, parts) + blkbytes * NB * (res ? HC : 1) <= LDS_BUDGET) {
                            best.split = parts;
                            best_nbt = 1;
                        }
                    }
                }
            }
        }
    if (best_cost < 0) return false;
    if (best.hb) best.G = 2 * best.nblk;
    *out = best;
    if (nbt_out) *nbt_out = best_nbt;
    return true;
}

int pick_chunk_adj3c(int N, int H, int B, int T, int control, int method)
{
    D1 d1;
    if (plan_adj3c(N, H, B, T, control, method, &d1, nullptr)) return B;
    if (control != PHX_CTRL_PER_TRAJECTORY) return 0;
    for (int bc = 4096; bc >= 16; bc >>= 1)
        if (bc < B && plan_adj3c(N, H, bc, T, control, method, &d1, nullptr)) return bc;
    return 0;
}

// floats of one batch group's partial: accumulator-native [gene block][chunk][4 HT x 2][64] float4, then dg [N], dbs [H], dbp [H]
size_t pp_adj3c(const D1 &d) { return align_up((size_t)d.nblk * d.HC * (4 * d.HT * 2 * 256) + d.N + 2 * d.H, 64); }

struct Layout3C {
    size_t total, cnt, part, zbuf, part1, zbuf1, scratch, dtheta, prof, xbytes, wimg, hq;
};

Layout3C make_layout3c(const D1 &d, bool grads)
{
    Layout3C L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t R = (size_t)d.ntg * d.HC * 4 * d.HT * 4 + d.ntg;   // hidden rows of every chunk + norm rows per group
    L.cnt = take(4096);
    L.part = take((size_t)d.TG * d.G * R * 64 * 8);
    L.zbuf = take((size_t)d.TG * R * 64 * 8);
    L.xbytes = off - L.part;                                 // header + set 0: what a fill covers (phx_mfma_v3common.inc: XSet)
    L.part1 = take((size_t)d.TG * d.G * R * 64 * 8);         // set 1: cleaned by the launch that works in set 0
    L.zbuf1 = take((size_t)d.TG * R * 64 * 8);
    L.scratch = take((size_t)d.TG * d.G * NVEC_ADJ3C * d.ntg * d.NB * 512 * 4);
    L.dtheta = take(grads ? pp_adj3c(d) * 4 * d.TG : 0);
    L.prof = take((size_t)d.TG * d.G * 16 * 8);
    L.wimg = take((size_t)d.nblk * d.HC * blk_floats_ch(d.HT, d.Hc) * 4);
    // transposed hidden rows of the seven ring slots, shared by a group's workgroups: [group][tile][7][chunk][4 HT][64] float4
    L.hq = take(grads ? (size_t)d.TG * d.ntg * 7 * d.HC * 4 * d.HT * 1024 : 0);
    L.total = off;
    return L;
}

size_t lds_bytes_adj3c(const D1 &d)
{
    return (size_t)blk_floats_ch(d.HT, d.Hc) * 4 * d.NB * (d.res ? d.HC : 1) +
           (d.hb ? hsa_offset(d.Bt, d.ntg) + HSA_BYTES
                 : (d.split > 1 ? hsa_offset(d.Bt, d.ntg) + spa_bytes(d.ntg, d.split) : ctl3c_bytes(d.Bt, d.ntg)));
}

}  // namespace

namespace phxh {

int adj3c_chunk(int N, int H, int B, int T, int control, int method) { return pick_chunk_adj3c(N, H, B, T, control, method); }

size_t adj3c_workspace_bytes(int N, int H, int B, int T)
{
    size_t need = 0;
    for (int ctl = 0; ctl < 2; ++ctl) {
        D1 d1;
        const int bc = pick_chunk_adj3c(N, H, B, T, ctl, PHX_DOPRI5);
        if (bc > 0 && plan_adj3c(N, H, bc, T, ctl, PHX_DOPRI5, &d1, nullptr)) need = std::max(need, make_layout3c(d1, true).total);
    }
    return need;
}

int adj3c_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6)
{
    D1 d1;
    if (!plan_adj3c(N, H, B, T, control, PHX_DOPRI5, &d1, nullptr)) return PHX_ERR_BAD_ARG;
    *offset = make_layout3c(d1, true).prof;
    *n_workgroups = d1.TG * d1.G;
    if (plan6) { plan6[0] = d1.NW; plan6[1] = d1.TPW; plan6[2] = d1.NB; plan6[3] = d1.G; plan6[4] = d1.TG; plan6[5] = d1.HC * 10 + d1.res + 2 * d1.hb + (d1.split > 1 ? 4 : 0); }
    return PHX_OK;
}

int adj3c_run(const phx_params *p, const double *t_all, int B, int T, const phx_solve_opts *o, const float *y_saved_all,
              const float *grad_y_all, float *adj_y0_all, const phx_grads *grads, int *status_all, int *nfe_all,
              int *nsteps_all, void *workspace, size_t workspace_bytes, hipStream_t st)
{
    SolveCfg cfg;
    cfg.method = o->method; cfg.control = o->control; cfg.t_per_sample = o->t_per_sample; cfg.t_is_f32 = o->t_is_f32;
    cfg.rtol = (float)o->rtol; cfg.atol = (float)o->atol;
    cfg.max_steps = o->max_num_steps > 0 ? o->max_num_steps : 2147483647LL;
    const int chunk = pick_chunk_adj3c(p->N, p->H, B, T, o->control, o->method);
    if (chunk <= 0) return PHX_ERR_BAD_ARG;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        D1 d1;
        int nbt = 0;
        const int bc = std::min(chunk, B - b0);
        if (!plan_adj3c(p->N, p->H, bc, T, o->control, o->method, &d1, &nbt)) return PHX_ERR_BAD_ARG;
        d1.BN = (long long)B * p->N;   // time stride of the caller's [T,B,N] arrays
        const double *t = !o->t_per_sample ? t_all   // rows of b0 onward; the buffer holds floats when t_is_f32 == 2
                          : reinterpret_cast<const double *>(reinterpret_cast<const char *>(t_all) +
                                                             (size_t)b0 * T * (o->t_is_f32 == 2 ? 4 : 8));
        const float *y_saved = y_saved_all + (long long)b0 * p->N, *grad_y = grad_y_all + (long long)b0 * p->N;
        float *adj_y0 = adj_y0_all + (long long)b0 * p->N;
        int *status = status_all + b0, *nfe = nfe_all + b0, *nsteps = nsteps_all + b0;
        const Layout3C L = make_layout3c(d1, grads != nullptr);
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
        const char *pe = getenv("PHX_PROF");   // 1: segment timers
        const int plevel = pe ? atoi(pe) : 0;
        w1.prof = plevel >= 1 ? (unsigned long long *)(base + L.prof) : nullptr;
        w1.wimg = (const float *)(base + L.wimg);
        w1.hq = (float *)(base + L.hq);
        const size_t lds = lds_bytes_adj3c(d1);
        const long long PP = (long long)pp_adj3c(d1);
        const int npart = d1.TG;   // one partial per batch group (the quadrature's items: every slot has one owner wave)
        const bool fill = !(o->ws_keep && chunk >= B);
        if (fill && hipMemsetAsync(w1.cnt, 0, L.part - L.cnt + L.xbytes, st) != hipSuccess) return PHX_ERR_LAUNCH;
        // every item's owner first-touches its slots (plain stores) in the first quadrature visit, or zero-fills them at the
        // end of the launch when its group never stepped; with T < 2 nobody runs either
        if (grads && T < 2 && hipMemsetAsync(w1.dtheta, 0, sizeof(float) * (size_t)PP * npart, st) != hipSuccess)
            return PHX_ERR_LAUNCH;
        const dim3 grid1(d1.TG * d1.G), blk1(64 * d1.NW);
        // the chunk images of this kernel family are not the caller's phx_params.wimg format for H > 48: packed per launch
        hipLaunchKernelGGL(k1_pack_images, dim3(d1.nblk * d1.HC), dim3(256), 0, st, to_net(p), (float *)w1.wimg, d1.HT, d1.HC,
                           d1.Hc, blk_floats_ch(d1.HT, d1.Hc));
        const bool half = d1.Hc <= 40;   // every chunk's last tile has at most 8 live rows (rho16, phx_mfma_v3common.inc)
        const void *fn;
        switch (nbt * 2 + (half ? 1 : 0) + (d1.hb ? 32 : 0)) {
        case 35: fn = reinterpret_cast<const void *>(k1_solve_adj3c<1, true, true>); break;
        case 34: fn = reinterpret_cast<const void *>(k1_solve_adj3c<1, false, true>); break;
        case 3: fn = reinterpret_cast<const void *>(k1_solve_adj3c<1, true, false>); break;
        case 2: fn = reinterpret_cast<const void *>(k1_solve_adj3c<1, false, false>); break;
        case 9: fn = reinterpret_cast<const void *>(k1_solve_adj3c<4, true, false>); break;
        case 8: fn = reinterpret_cast<const void *>(k1_solve_adj3c<4, false, false>); break;
        default: return PHX_ERR_BAD_ARG;
        }
        if (!set_lds_fn(fn, lds)) return PHX_ERR_LAUNCH;
        // the workgroups of a launch wait for each other's rows: refuse a grid the device cannot hold at once
        if (!fits_resident(fn, 64 * d1.NW, lds, d1.TG * d1.G)) return PHX_ERR_LAUNCH;
        // (diagnostic kernel events: ONE pair around all launches of a batch that runs in several)
        if (b0 == 0) ev_begin(st);
        const hipError_t lerr = launch_persistent(fn, grid1, blk1, lds, st, to_net(p), d1, w1, cfg, t, y_saved, grad_y, adj_y0,
                                                  status, nfe, nsteps, (int)(grads ? 1 : 0), PP);
        if (b0 + chunk >= B) ev_end(st);
        if (lerr != hipSuccess || hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
        if (grads) {
            const long long total = (long long)d1.nblk * d1.HC * (4 * d1.HT * 2 * 64) + p->N + 2 * p->H;
            hipLaunchKernelGGL(k3c_reduce_grads, dim3((unsigned int)((total + 255) / 256)), dim3(256), 0, st, w1.dtheta, npart, PP,
                               p->N, p->H, d1.nblk, d1.HC, d1.Hc, grads->Ws, grads->Wp, grads->WaT, grads->g, grads->bs,
                               grads->bp, (grads->overwrite && b0 == 0) ? 1 : 0, grads->Wa);   // later chunks of a large batch add
            if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
        }
    }
    return PHX_OK;
}

}  // namespace phxh
