// Diagnostic: is v_pk_fma_f32 bit-identical to v_fma_f32 (same operands) on gfx950?
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
typedef float f2 __attribute__((ext_vector_type(2)));
__global__ void k(const float *a, const float *b, const float *c, unsigned int *bad, int n)
{
    const int i = (blockIdx.x * blockDim.x + threadIdx.x) * 2;
    if (i + 2 > n) return;
    f2 va = {a[i], a[i + 1]}, vb = {b[i], b[i + 1]}, vc = {c[i], c[i + 1]}, r;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3" : "=v"(r) : "v"(va), "v"(vb), "v"(vc));
    float s0, s1;
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s0) : "v"(va.x), "v"(vb.x), "v"(vc.x));
    asm volatile("v_fma_f32 %0, %1, %2, %3" : "=v"(s1) : "v"(va.y), "v"(vb.y), "v"(vc.y));
    if (__float_as_uint(r.x) != __float_as_uint(s0)) atomicAdd(bad, 1u);
    if (__float_as_uint(r.y) != __float_as_uint(s1)) atomicAdd(bad, 1u);
    // the same with a negated operand (modifier form used by the compiler)
    f2 r2;
    asm volatile("v_pk_fma_f32 %0, %1, %2, %3 neg_lo:[0,1,0] neg_hi:[0,1,0]" : "=v"(r2) : "v"(va), "v"(vb), "v"(vc));
    float t0, t1;
    asm volatile("v_fma_f32 %0, %1, -%2, %3" : "=v"(t0) : "v"(va.x), "v"(vb.x), "v"(vc.x));
    asm volatile("v_fma_f32 %0, %1, -%2, %3" : "=v"(t1) : "v"(va.y), "v"(vb.y), "v"(vc.y));
    if (__float_as_uint(r2.x) != __float_as_uint(t0)) atomicAdd(bad + 1, 1u);
    if (__float_as_uint(r2.y) != __float_as_uint(t1)) atomicAdd(bad + 1, 1u);
}
int main()
{
    const int n = 1 << 22;
    float *h = (float *)malloc(3 * n * 4), *d;
    srand(3);
    for (int i = 0; i < 3 * n; ++i) h[i] = ((float)rand() / RAND_MAX - 0.5f) * ((i % 7 == 0) ? 1e-3f : 2.f);
    unsigned int *b;
    hipMalloc(&d, 3 * n * 4); hipMalloc(&b, 8); hipMemset(b, 0, 8);
    hipMemcpy(d, h, 3 * n * 4, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 2 / 256), dim3(256), 0, 0, d, d + n, d + 2 * n, b, n);
    unsigned int r[2];
    hipMemcpy(r, b, 8, hipMemcpyDeviceToHost);
    printf("v_pk_fma_f32 vs v_fma_f32 on %d triples: plain differs %u, with neg modifier differs %u\n", n, r[0], r[1]);
    return 0;
}
