// Diagnostic: do more than 1024 co-resident waves each keep their own private (scratch) memory?
// Persistent grid of `nwg` workgroups x 512 threads, 150 KB of dynamic LDS (one workgroup per CU); every lane fills a
// private array that the compiler must keep in scratch (dynamic indexing), all workgroups meet at a counter barrier,
// then every lane verifies its array.  Prints the number of lanes that read back something else.
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>

__global__ __launch_bounds__(512) void k(unsigned int *counter, unsigned int *bad, int nwg, int rounds)
{
    extern __shared__ float lds[];
    volatile unsigned int priv[48];   // dynamically indexed below => scratch
    const unsigned int id = blockIdx.x * blockDim.x + threadIdx.x;
    unsigned int errors = 0;
    for (int r = 0; r < rounds; ++r) {
        for (int i = 0; i < 48; ++i) priv[(i * 7 + r) % 48] = id * 977u + (unsigned)((i * 7 + r) % 48) * 31u + r;
        lds[threadIdx.x] = (float)r;
        __syncthreads();
        if (threadIdx.x == 0) {
            __threadfence();
            atomicAdd(counter, 1u);
            while (__hip_atomic_load(counter, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT) < (unsigned)nwg * (r + 1)) __builtin_amdgcn_s_sleep(4);
        }
        __syncthreads();
        for (int i = 0; i < 48; ++i)
            if (priv[i] != id * 977u + (unsigned)i * 31u + r) errors++;
    }
    if (errors) atomicAdd(bad, 1u);
}

int main(int argc, char **argv)
{
    const int nwg = argc > 1 ? atoi(argv[1]) : 256, rounds = 50;
    unsigned int *d;
    hipMalloc(&d, 8);
    hipMemset(d, 0, 8);
    hipFuncSetAttribute(reinterpret_cast<const void *>(k), hipFuncAttributeMaxDynamicSharedMemorySize, 150 * 1024);
    hipLaunchKernelGGL(k, dim3(nwg), dim3(512), 150 * 1024, 0, d, d + 1, nwg, rounds);
    hipError_t e = hipDeviceSynchronize();
    unsigned int h[2];
    hipMemcpy(h, d, 8, hipMemcpyDeviceToHost);
    printf("workgroups %d (waves %d): %s, lanes with a wrong private value: %u\n", nwg, nwg * 8, hipGetErrorString(e), h[1]);
    return 0;
}
