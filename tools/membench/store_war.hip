// Reproducer for the "trajectories 12..15 of every tile" wrong-result signature of the third-generation solve kernels
// ("one signature").
//
// What the bad builds of k1_solve_adj3 / k1_solve_fwd3 had in common (ISA of a bad instantiation, round 4):
//     buffer_store_dwordx4 v[124:127], v41, s[8:11], s1 offen      ; k_y tile, genes 0..3 of every lane
//     v_mov_b32_e32 v124, v117                                      ; the NEXT instruction re-uses the data registers
//     v_mov_b32_e32 v125, v115 ...
// i.e. a 128-bit MUBUF store whose data VGPRs are overwritten by VALU instructions with ZERO wait states in between.
// LLVM's hazard recognizer knows this write-after-read hazard ("VALU writes the data VGPRs of a VMEM store wider than
// 64 bits": 1 wait state, 2 on gfx940+) but exempts MUBUF / MTBUF stores that take their soffset from an SGPR
// (GCNHazardRecognizer::createsVALUHazard: "this hazard only exists if the instruction is not using a register in the
// soffset field") -- and the third-generation kernels address every private tile as (buffer resource, SGPR tile offset,
// one lane offset).  This program checks what the hardware does: every wave stores a tuple holding A, overwrites the
// tuple with B in the very next instruction, and the host counts the lanes whose stored value is B (or a mix).
//   variant 0: soffset = SGPR, 0 wait states   (what the compiler emits for the kernels)
//   variant 1: soffset = SGPR, s_nop 1 between (the wait states LLVM would insert without the exemption)
//   variant 2: soffset = immediate-zero form (`off`), 0 wait states (the case LLVM does protect)
// usage: hipcc --offload-arch=gfx950 -O2 tools/membench/store_war.hip -o /tmp/store_war && /tmp/store_war
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <cstring>

typedef int i32x4 __attribute__((ext_vector_type(4)));

template <int VAR>
__global__ __launch_bounds__(256) void k(float *out, int tiles_per_wave, int soff_words)
{
    const int lane = threadIdx.x & 63;
    const int wave = __builtin_amdgcn_readfirstlane((int)((blockIdx.x * blockDim.x + threadIdx.x) >> 6));   // uniform
    float *base = out + (long long)wave * tiles_per_wave * 512;          // [tile][64 lanes][8 floats]: the kernels' tile layout
    const unsigned long long ba = reinterpret_cast<unsigned long long>(base);
    const i32x4 rs = {__builtin_amdgcn_readfirstlane((int)(unsigned int)ba),
                      __builtin_amdgcn_readfirstlane((int)(unsigned int)(ba >> 32)), tiles_per_wave * 2048, 0x00020000};
    const int voff = lane * 32;
    const float a0 = 1.0f, a1 = 2.0f, a2 = 3.0f, a3 = 4.0f, b0 = -1.0f, b1 = -2.0f, b2 = -3.0f, b3 = -4.0f;
    for (int t = 0; t < tiles_per_wave; ++t) {
        const int soff = __builtin_amdgcn_readfirstlane(t * 2048 + soff_words * 0);
        // a few stores in flight first (the kernels' sweeps keep the memory pipeline busy), then the critical pair
        if (VAR == 0)
            asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\ts_nop 7\n\t"
                         "buffer_store_dwordx4 v[10:13], %8, %9, %10 offen offset:16\n\t"
                         "buffer_store_dwordx4 v[10:13], %8, %9, %10 offen\n\t"
                         "v_mov_b32 v10, %4\n\tv_mov_b32 v11, %5\n\tv_mov_b32 v12, %6\n\tv_mov_b32 v13, %7\n\t"
                         "s_waitcnt vmcnt(0)"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(voff), "s"(rs), "s"(soff)
                         : "v10", "v11", "v12", "v13", "memory");
        else if (VAR == 1)
            asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\ts_nop 7\n\t"
                         "buffer_store_dwordx4 v[10:13], %8, %9, %10 offen offset:16\n\t"
                         "buffer_store_dwordx4 v[10:13], %8, %9, %10 offen\n\t"
                         "s_nop 1\n\t"
                         "v_mov_b32 v10, %4\n\tv_mov_b32 v11, %5\n\tv_mov_b32 v12, %6\n\tv_mov_b32 v13, %7\n\t"
                         "s_waitcnt vmcnt(0)"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(voff), "s"(rs), "s"(soff)
                         : "v10", "v11", "v12", "v13", "memory");
        else {
            const int voff2 = voff + soff;
            asm volatile("v_mov_b32 v10, %0\n\tv_mov_b32 v11, %1\n\tv_mov_b32 v12, %2\n\tv_mov_b32 v13, %3\n\ts_nop 7\n\t"
                         "buffer_store_dwordx4 v[10:13], %8, %9, 0 offen offset:16\n\t"
                         "buffer_store_dwordx4 v[10:13], %8, %9, 0 offen\n\t"
                         "v_mov_b32 v10, %4\n\tv_mov_b32 v11, %5\n\tv_mov_b32 v12, %6\n\tv_mov_b32 v13, %7\n\t"
                         "s_waitcnt vmcnt(0)"
                         :: "v"(a0), "v"(a1), "v"(a2), "v"(a3), "v"(b0), "v"(b1), "v"(b2), "v"(b3), "v"(voff2), "s"(rs)
                         : "v10", "v11", "v12", "v13", "memory");
        }
    }
}

int main()
{
    const int waves_per_wg = 4, wgs = 1024, tiles = 64;
    const size_t n = (size_t)wgs * waves_per_wg * tiles * 512;
    float *d, *h = (float *)malloc(n * 4);
    hipMalloc(&d, n * 4);
    const char *names[3] = {"SGPR soffset, 0 wait states", "SGPR soffset, s_nop 1", "soffset = 0 (off), 0 wait states"};
    for (int var = 0; var < 3; ++var) {
        hipMemset(d, 0, n * 4);
        for (int rep = 0; rep < 4; ++rep) {
            if (var == 0) hipLaunchKernelGGL(k<0>, dim3(wgs), dim3(64 * waves_per_wg), 0, 0, d, tiles, 0);
            if (var == 1) hipLaunchKernelGGL(k<1>, dim3(wgs), dim3(64 * waves_per_wg), 0, 0, d, tiles, 0);
            if (var == 2) hipLaunchKernelGGL(k<2>, dim3(wgs), dim3(64 * waves_per_wg), 0, 0, d, tiles, 0);
        }
        hipDeviceSynchronize();
        hipMemcpy(h, d, n * 4, hipMemcpyDeviceToHost);
        long long bad = 0, by_lane[64];
        memset(by_lane, 0, sizeof(by_lane));
        for (size_t tile = 0; tile < n / 512; ++tile)
            for (int lane = 0; lane < 64; ++lane)
                for (int half = 0; half < 2; ++half)
                    for (int c = 0; c < 4; ++c) {
                        const float v = h[tile * 512 + lane * 8 + half * 4 + c];
                        if (v != (float)(c + 1)) { ++bad; ++by_lane[lane]; }
                    }
        printf("%-34s: %lld of %zu stored values are not the value the store was issued with\n", names[var], bad, n);
        if (bad) {
            printf("   bad values by lane:");
            for (int lane = 0; lane < 64; ++lane)
                if (by_lane[lane]) printf(" %d:%lld", lane, by_lane[lane]);
            printf("\n");
        }
    }
    return 0;
}
