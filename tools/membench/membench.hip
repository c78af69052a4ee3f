// Diagnostic: achievable tile-stream rate of the backward kernel's access pattern without any compute.
// 236 workgroups x NW waves; a wave walks NBLK (tile, block) slots; per slot it reads NL vectors' 2 KB tiles
// (two dwordx4 per lane) and writes NS tiles; software prefetch distance 1 (like the hot sweeps).
// build: hipcc --offload-arch=gfx950 -O3 tools/membench/membench.hip -o tools/membench/membench ; run on the GPU box
#include <hip/hip_runtime.h>
#ifndef SLEEPN
#define SLEEPN 0   // s_sleep between a slot's loads and stores (stand-in for compute)
#endif
#include <cstdio>
#include <cstdlib>
template <int NL, int NS>
__global__ __launch_bounds__(512) void k(float *base, int nvec, int slots_per_wave, int reps, float *sink)
{
    const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6, nw = blockDim.x >> 6;
    float *wg = base + (long long)blockIdx.x * nvec * slots_per_wave * nw * 512;
    auto tp = [&](int v, int slot) { return wg + ((long long)(v * nw * slots_per_wave) + wv * slots_per_wave + slot) * 512 + lane * 8; };
    float acc = 0.f;
    for (int r = 0; r < reps; ++r) {
        float4 a[2][NL][2];
#pragma unroll
        for (int v = 0; v < NL; ++v) { a[0][v][0] = *(float4 *)tp((v + r) % nvec, 0); a[0][v][1] = *(float4 *)(tp((v + r) % nvec, 0) + 4); }
        for (int s = 0; s < slots_per_wave; s += 2) {
#pragma unroll
            for (int h = 0; h < 2; ++h) {
                const int sl = s + h, sn = min(sl + 1, slots_per_wave - 1);
#pragma unroll
                for (int v = 0; v < NL; ++v) { a[1 - h][v][0] = *(float4 *)tp((v + r) % nvec, sn); a[1 - h][v][1] = *(float4 *)(tp((v + r) % nvec, sn) + 4); }
                float4 o0 = a[h][0][0], o1 = a[h][0][1];
#pragma unroll
                for (int v = 1; v < NL; ++v) { o0.x += a[h][v][0].x; o0.y += a[h][v][0].y; o1.x += a[h][v][1].x; o1.w += a[h][v][1].w; }
                __builtin_amdgcn_s_sleep(SLEEPN);
#pragma unroll
                for (int q = 0; q < NS; ++q) { *(float4 *)tp((NL + q + r) % nvec, sl) = o0; *(float4 *)(tp((NL + q + r) % nvec, sl) + 4) = o1; }
                acc += o0.x;
            }
        }
    }
    if (acc == 12345.678f) sink[0] = acc;
}
int main(int argc, char **argv)
{
    const int nwg = 236, nvec = 32;
    for (int nw : {4, 8}) {
        const int slots = 48 / nw;   // 24 tile-blocks per workgroup like C4 (4 tiles x 6 blocks), split over the waves
        const size_t bytes = (size_t)nwg * nvec * slots * nw * 512 * 4;
        float *buf, *sink;
        hipMalloc(&buf, bytes); hipMalloc(&sink, 4); hipMemset(buf, 0, bytes);
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        const int reps = 40;
        auto run = [&](auto kern, int nl, int ns, const char *name) {
            hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * nw), 0, 0, buf, nvec, slots, 2, sink);
            hipEventRecord(e0); hipLaunchKernelGGL(kern, dim3(nwg), dim3(64 * nw), 0, 0, buf, nvec, slots, reps, sink); hipEventRecord(e1);
            hipEventSynchronize(e1); float ms; hipEventElapsedTime(&ms, e0, e1);
            const double b = (double)nwg * nw * slots * reps * (nl + ns) * 2048.0;
            printf("waves/WG %d  %s: %.3f ms  %.2f TB/s (reads+writes), footprint %.0f MB\n", nw, name, ms, b / ms / 1e9, bytes / 1e6);
        };
        run(k<6, 2>, 6, 2, "6 loads + 2 stores per slot");
        run(k<8, 3>, 8, 3, "8 loads + 3 stores per slot");
        run(k<8, 0>, 8, 0, "8 loads, no stores");
        run(k<1, 3>, 1, 3, "1 load + 3 stores");
        hipFree(buf); hipFree(sink);
    }
    return 0;
}
