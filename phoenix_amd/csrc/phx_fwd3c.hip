// phx_fwd3c.hip -- translation unit of the third-generation forward solve for wide hidden layers (k1_solve_fwd3c,
// phx_mfma_fwd3c.inc: hidden chunks of <= 48 rows): launch planning, workspace layout and the host entry points the C ABI
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
#include "phx_mfma_fwd3c.inc"

namespace {

bool fwd3c_disabled()
{
    const char *e = getenv("PHX_FWD");   // diagnostic: PHX_FWD=v1 keeps every forward solve on k1_solve_fwd
    const char *c = getenv("PHX_V3C");   // diagnostic: PHX_V3C=0 switches the chunked third-generation kernels off
    return force_v0() || (e && strcmp(e, "v1") == 0) || (c && c[0] == '0');
}

int env_int(const char *name, int dflt)
{
    const char *e = getenv(name);
    return e ? atoi(e) : dflt;
}

// Hidden chunks: HC = ceil(H / 48) chunks of Hc = ceil(H / HC) rows, three 16-row tiles each (120 -> 3 x 40, 200 -> 5 x 40).
// (NW, TPW, NB): one wave per SIMD, gene tiles of 1 / 2 / 4 / 8 blocks, at most 8 (tile, block) slots per wave (the kernel
// keeps their partial sums in registers); all chunks LDS resident when they fit, else one re-staged slot.  Cost: the
// per-wave MFMA work first, then the exchange volume (members per group).  A plan of ONE slot per wave whose workgroups
// leave half the chip idle takes HALF-BLOCK gene tiles (hb: twice the workgroups, half the sweep work each).
bool plan_fwd3c(int N, int H, int B, int T, int control, int method, D1 *out, int *nbt_out)
{
    const int cus = num_cus();
    if (cus <= 0 || fwd3c_disabled() || method != PHX_DOPRI5 || H <= 48 || H > 256) return false;
    const int HT = 3, HC = (H + 47) / 48, Hc = (H + HC - 1) / HC;
    const size_t blkbytes = (size_t)blk_floats_ch(HT, Hc) * 4;
    const int nblk = (N + 31) / 32, ntt = (B + 15) / 16;
    const int fnb = env_int("PHX_V3C_NB", 0), ftpw = env_int("PHX_V3C_TPW", 0), fres = env_int("PHX_V3C_RES", -1);
    const int fhb = env_int("PHX_V3C_HB", -1);
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
                // (one-tile groups may take three quarters of the chip, two-tile groups half of it: beyond, the forward
                // launch measured +8 ... +18 % -- 252 half-block workgroups at two tiles per group -- where the backward gains)
                for (int c = 1; c < ntg; c <<= 1) {
                    const int tgc = (ntt + c - 1) / c;
                    if ((long long)tgc * nblk * 2 * 4 <= (long long)cus * (c == 1 ? 3 : 2)) { ntg = c; TG = tgc; break; }
                }
            }
            const int Bt = 16 * ntg;
            const bool helpers = ntg < slots;
            if (control == PHX_CTRL_SHARED && TG != 1) continue;
            const size_t cb = ctlf3c_bytes(Bt, ntg);
            for (int NB = 1; NB <= 8 && TPW * NB <= 8; NB <<= 1) {
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
                    best.nvec = NVEC_FWD3C; best.BN = (long long)B * N; best.HC = HC; best.Hc = Hc;
                    best.Bcall = 0; best.cntN = (long long)B * N; best.res = res ? 1 : 0;
                    const int nslot = TPW * NB;
                    // measured (round 5, H = 120 / 200, 8..220 gene blocks, 16..128 trajectories): -9..-33 % of a launch while
                    // the groups' rows times their workgroups stay small (four-tile groups on at most half the chip, one-tile
                    // groups on all of it) and a group has at most 160 members; +5..+24 % beyond (tools/v3c_check.py hbgrid)
                    const bool hb_fits = nslot == 1 && res && (long long)TG * G * 2 <= cus && cb + HSF_BYTES + 16 + blkbytes * HC <= LDS_BUDGET;
                    best.hb = (hb_fits && fhb != 0 && (fhb == 1 || ((long long)TG * G * ntg <= cus && 2 * nblk <= 160))) ? 1 : 0;
                    best_nbt = nslot <= 1 ? 1 : 8;   // (unused slots of the eight-slot form cost a scalar branch each)
                    // block split: a tile's gene blocks on the workgroup's spare waves (small batches of multi-block tiles)
                    best.split = 1;
                    if (!best.hb && TPW == 1 && NB > 1 && ntg * NB <= NW && env_int("PHX_V3C_SPLIT", 1) != 0) {
                        const int parts = std::min(NW / ntg, NB);
                        if (hsf_offset(Bt, ntg) + spf_bytes(ntg, parts) + blkbytes * NB * (res ? HC : 1) <= LDS_BUDGET) {
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

int pick_chunk_fwd3c(int N, int H, int B, int T, int control, int method)
{
    D1 d1;
    if (plan_fwd3c(N, H, B, T, control, method, &d1, nullptr)) return B;
    if (control != PHX_CTRL_PER_TRAJECTORY) return 0;
    for (int bc = 4096; bc >= 16; bc >>= 1)
        if (bc < B && plan_fwd3c(N, H, bc, T, control, method, &d1, nullptr)) return bc;
    return 0;
}

struct LayoutF3C {
    size_t total, cnt, part, zbuf, part1, zbuf1, scratch, prof, xbytes, wimg;
};

LayoutF3C make_layout_f3c(const D1 &d)
{
    LayoutF3C L;
    size_t off = 0;
    auto take = [&](size_t bytes) { size_t o = off; off = align_up(off + bytes, 256); return o; };
    const size_t R = (size_t)d.ntg * d.HC * 2 * d.HT * 4 + d.ntg;   // hidden rows of every chunk + norm rows per group
    L.cnt = take(4096);
    L.part = take((size_t)d.TG * d.G * R * 64 * 8);
    L.zbuf = take((size_t)d.TG * R * 64 * 8);
    L.xbytes = off - L.part;                                 // header + set 0: what a fill covers (phx_mfma_v3common.inc: XSet)
    L.part1 = take((size_t)d.TG * d.G * R * 64 * 8);         // set 1: cleaned by the launch that works in set 0
    L.zbuf1 = take((size_t)d.TG * R * 64 * 8);
    L.scratch = take((size_t)d.TG * d.G * NVEC_FWD3C * d.ntg * d.NB * 512 * 4);
    L.prof = take((size_t)d.TG * d.G * 16 * 8);
    L.wimg = take((size_t)d.nblk * d.HC * blk_floats_ch(d.HT, d.Hc) * 4);
    L.total = off;
    return L;
}

size_t lds_bytes_fwd3c(const D1 &d)
{
    return (size_t)blk_floats_ch(d.HT, d.Hc) * 4 * d.NB * (d.res ? d.HC : 1) +
           (d.hb ? hsf_offset(d.Bt, d.ntg) + HSF_BYTES
                 : (d.split > 1 ? hsf_offset(d.Bt, d.ntg) + spf_bytes(d.ntg, d.split) : ctlf3c_bytes(d.Bt, d.ntg)));
}

}  // namespace

namespace phxh {

int fwd3c_chunk(int N, int H, int B, int T, int control, int method) { return pick_chunk_fwd3c(N, H, B, T, control, method); }

size_t fwd3c_workspace_bytes(int N, int H, int B, int T)
{
    size_t need = 0;
    for (int ctl = 0; ctl < 2; ++ctl) {
        D1 d1;
        const int bc = pick_chunk_fwd3c(N, H, B, T, ctl, PHX_DOPRI5);
        if (bc > 0 && plan_fwd3c(N, H, bc, T, ctl, PHX_DOPRI5, &d1, nullptr)) need = std::max(need, make_layout_f3c(d1).total);
    }
    return need;
}

int fwd3c_profile_region(int N, int H, int B, int T, int control, size_t *offset, int *n_workgroups, int *plan6)
{
    D1 d1;
    if (!plan_fwd3c(N, H, B, T, control, PHX_DOPRI5, &d1, nullptr)) return PHX_ERR_BAD_ARG;
    *offset = make_layout_f3c(d1).prof;
    *n_workgroups = d1.TG * d1.G;
    if (plan6) { plan6[0] = d1.NW; plan6[1] = d1.TPW; plan6[2] = d1.NB; plan6[3] = d1.G; plan6[4] = d1.TG; plan6[5] = d1.HC * 10 + d1.res + 2 * d1.hb + (d1.split > 1 ? 4 : 0); }
    return PHX_OK;
}

int fwd3c_run(const phx_params *p, const float *y0_all, const double *t_all, int B, int T, const phx_solve_opts *o,
              float *sol_all, int *status_all, int *nfe_all, int *nsteps_all, void *workspace, size_t workspace_bytes,
              hipStream_t st)
{
    SolveCfg cfg;
    cfg.method = o->method; cfg.control = o->control; cfg.t_per_sample = o->t_per_sample; cfg.t_is_f32 = o->t_is_f32;
    cfg.rtol = (float)o->rtol; cfg.atol = (float)o->atol;
    cfg.max_steps = o->max_num_steps > 0 ? o->max_num_steps : 2147483647LL;
    const int chunk = pick_chunk_fwd3c(p->N, p->H, B, T, o->control, o->method);
    if (chunk <= 0) return PHX_ERR_BAD_ARG;
    for (int b0 = 0; b0 < B; b0 += chunk) {
        D1 d1;
        int nbt = 0;
        const int bc = std::min(chunk, B - b0);
        if (!plan_fwd3c(p->N, p->H, bc, T, o->control, o->method, &d1, &nbt)) return PHX_ERR_BAD_ARG;
        d1.BN = (long long)B * p->N;   // time stride of the caller's [T,B,N] arrays
        const float *y0 = y0_all + (long long)b0 * p->N;
        const double *t = !o->t_per_sample ? t_all   // rows of b0 onward; the buffer holds floats when t_is_f32 == 2
                          : reinterpret_cast<const double *>(reinterpret_cast<const char *>(t_all) +
                                                             (size_t)b0 * T * (o->t_is_f32 == 2 ? 4 : 8));
        float *sol = sol_all + (long long)b0 * p->N;
        int *status = status_all + b0, *nfe = nfe_all + b0, *nsteps = nsteps_all + b0;
        const LayoutF3C L = make_layout_f3c(d1);
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
        const char *pe = getenv("PHX_PROF");
        w1.prof = (pe && atoi(pe) >= 1) ? (unsigned long long *)(base + L.prof) : nullptr;
        w1.prof_level = pe ? atoi(pe) : 0;
        w1.wimg = (const float *)(base + L.wimg);
        const size_t lds = lds_bytes_fwd3c(d1);
        const bool fill = !(o->ws_keep && chunk >= B);
        if (fill && hipMemsetAsync(w1.cnt, 0, L.part - L.cnt + L.xbytes, st) != hipSuccess) return PHX_ERR_LAUNCH;
        const dim3 grid1(d1.TG * d1.G), blk1(64 * d1.NW);
        // the chunk images of this kernel family (three-tile chunks) are not the caller's phx_params.wimg format for H > 48
        // (one eight-tile chunk / seven-tile 100-row chunks): packed per launch from the parameter tensors
        hipLaunchKernelGGL(k1_pack_images, dim3(d1.nblk * d1.HC), dim3(256), 0, st, to_net(p), (float *)w1.wimg, d1.HT, d1.HC,
                           d1.Hc, blk_floats_ch(d1.HT, d1.Hc));
        auto launch = [&](auto kern) -> int {
            const void *fn = reinterpret_cast<const void *>(kern);
            if (!set_lds_fn(fn, lds)) return PHX_ERR_LAUNCH;
            // the workgroups of a launch wait for each other's rows: refuse a grid the device cannot hold at once
            if (!fits_resident(fn, 64 * d1.NW, lds, d1.TG * d1.G)) return PHX_ERR_LAUNCH;
            if (b0 == 0) ev_begin(st);   // (ONE event pair around all launches of a batch that runs in several)
            const hipError_t lerr = launch_persistent(fn, grid1, blk1, lds, st, to_net(p), d1, w1, cfg, y0, t, sol, status, nfe, nsteps);
            if (b0 + chunk >= B) ev_end(st);
            return lerr == hipSuccess ? PHX_OK : PHX_ERR_LAUNCH;
        };
        const bool half = d1.Hc <= 40;   // every chunk's last tile has at most 8 live rows (rho16, phx_mfma_v3common.inc)
        int lrc;
        switch (nbt * 2 + (half ? 1 : 0) + (d1.hb ? 32 : 0)) {
        case 35: lrc = launch(k1_solve_fwd3c<1, true, true>); break;
        case 34: lrc = launch(k1_solve_fwd3c<1, false, true>); break;
        case 3: lrc = launch(k1_solve_fwd3c<1, true, false>); break;
        case 2: lrc = launch(k1_solve_fwd3c<1, false, false>); break;
        case 17: lrc = launch(k1_solve_fwd3c<8, true, false>); break;
        default: lrc = launch(k1_solve_fwd3c<8, false, false>); break;
        }
        if (lrc != PHX_OK) return lrc;
        if (hipGetLastError() != hipSuccess) return PHX_ERR_LAUNCH;
    }
    return PHX_OK;
}

}  // namespace phxh
