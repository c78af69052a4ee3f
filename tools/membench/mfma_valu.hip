// Diagnostic microbenchmark: does independent f32 VALU work hide under v_mfma_f32_16x16x4_f32 on gfx950 (one wave per
// SIMD)?  Per iteration: NM dependent-free MFMAs (4 accumulators) + NV independent v_fma_f32.  Prints cycles/iteration.
// build: hipcc --offload-arch=gfx950 -O3 mfma_valu.hip -o mfma_valu
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f32x4 __attribute__((ext_vector_type(4)));

template <int NM, int NV, int TR>
__global__ __launch_bounds__(256) void k(float *out, int iters, unsigned long long *cyc)
{
    f32x4 a0 = {0, 0, 0, 0}, a1 = a0, a2 = a0, a3 = a0;
    float x = threadIdx.x * 1e-3f, y = 1.0001f;
    float v[8];
#pragma unroll
    for (int i = 0; i < 8; ++i) v[i] = x + i;
    const unsigned long long t0 = clock64();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int m = 0; m < NM; m += 4) {
            a0 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a0, 0, 0, 0);
            a1 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a1, 0, 0, 0);
            a2 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a2, 0, 0, 0);
            a3 = __builtin_amdgcn_mfma_f32_16x16x4f32(x, y, a3, 0, 0, 0);
#pragma unroll
            for (int i = 0; i < (NV * 4) / (NM > 0 ? NM : 4); ++i) {
                if (TR) v[i & 7] = __builtin_amdgcn_rcpf(v[i & 7]);
                else v[i & 7] = __builtin_fmaf(v[i & 7], y, x);
            }
        }
        if (NM == 0) {
#pragma unroll
            for (int i = 0; i < NV; ++i) v[i & 7] = TR ? __builtin_amdgcn_rcpf(v[i & 7]) : __builtin_fmaf(v[i & 7], y, x);
        }
    }
    const unsigned long long t1 = clock64();
    float s = 0;
#pragma unroll
    for (int i = 0; i < 8; ++i) s += v[i];
    out[blockIdx.x * blockDim.x + threadIdx.x] = s + a0[0] + a1[1] + a2[2] + a3[3];
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
}

template <int NM, int NV, int TR>
void run(const char *name)
{
    float *out;
    unsigned long long *cyc, h[256];
    hipMalloc(&out, 256 * 256 * 4);
    hipMalloc(&cyc, 256 * 8);
    const int iters = 2000;
    for (int rep = 0; rep < 2; ++rep) hipLaunchKernelGGL((k<NM, NV, TR>), dim3(256), dim3(256), 0, 0, out, iters, cyc);
    hipDeviceSynchronize();
    hipMemcpy(h, cyc, sizeof(h), hipMemcpyDeviceToHost);
    double s = 0;
    for (int i = 0; i < 256; ++i) s += (double)h[i];
    printf("%-34s NM=%3d NV=%3d: %8.1f cycles / iteration  (MFMA alone %d, VALU alone ~%d)\n", name, NM, NV, s / 256 / iters, NM * 32,
           NV * 4);
    hipFree(out); hipFree(cyc);
}

int main()
{
    run<16, 0, 0>("16 mfma");
    run<0, 64, 0>("64 v_fma");
    run<16, 32, 0>("16 mfma + 32 v_fma (2/mfma)");
    run<16, 64, 0>("16 mfma + 64 v_fma (4/mfma)");
    run<16, 96, 0>("16 mfma + 96 v_fma (6/mfma)");
    run<16, 128, 0>("16 mfma + 128 v_fma (8/mfma)");
    run<0, 32, 1>("32 v_rcp");
    run<16, 32, 1>("16 mfma + 32 v_rcp (2/mfma)");
    return 0;
}
