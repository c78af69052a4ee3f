// Diagnostic: v_pk_fma_f32 with a 64-bit SGPR source and op_sel_hi = 0 for it ("use the LOW half for both lanes", the
// form the compiler emits for a broadcast constant): which half does the HIGH lane really use on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(float *out, float lo, float hi)
{
    f2 a = {2.0f, 3.0f}, b = {5.0f, 7.0f}, r;
    // constant pair {lo, hi} in SGPRs, src2, op_sel_hi bit for src2 = 0
    asm volatile("s_mov_b32 s8, %1\n\ts_mov_b32 s9, %2\n\tv_pk_fma_f32 %0, %3, %4, s[8:9] op_sel_hi:[1,1,0]"
                 : "=v"(r) : "s"(lo), "s"(hi), "v"(a), "v"(b) : "s8", "s9");
    out[0] = r.x; out[1] = r.y;
    // the same as src1
    asm volatile("s_mov_b32 s8, %1\n\ts_mov_b32 s9, %2\n\tv_pk_fma_f32 %0, %3, s[8:9], %4 op_sel_hi:[1,0,1]"
                 : "=v"(r) : "s"(lo), "s"(hi), "v"(a), "v"(b) : "s8", "s9");
    out[2] = r.x; out[3] = r.y;
}
int main()
{
    float *d, h[4];
    hipMalloc(&d, 16);
    hipLaunchKernelGGL(k, dim3(1), dim3(64), 0, 0, d, 100.0f, 1000.0f);
    hipMemcpy(h, d, 16, hipMemcpyDeviceToHost);
    printf("src2 = s[lo=100, hi=1000], op_sel_hi 0: lanes give %g (2*5+c) and %g (3*7+c): high lane used c = %g\n", h[0], h[1], h[1] - 21.f);
    printf("src1 = s[lo=100, hi=1000], op_sel_hi 0: lanes give %g (2*c+5) and %g (3*c+7): high lane used c = %g\n", h[2], h[3], (h[3] - 7.f) / 3.f);
    return 0;
}
