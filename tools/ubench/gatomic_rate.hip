// Microbenchmark: device-scope 64-bit atomic adds to RANDOM addresses of a global array (M elements), all compute
// units at once — the rate a "push" SpMV pass of the whole-device solver (k_solve_wide) would see if it scattered
// fixed-point contributions into a global accumulator instead of pulling rows.  Also: plain 8-byte gathers from the
// same array (the pull pass's access pattern).
// Build: hipcc --offload-arch=gfx950 -O3 -o gatomic_rate gatomic_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>

template <int MODE>
__global__ void __launch_bounds__(512) k(unsigned long long* acc, unsigned M, int iters, double* out)
{
    unsigned s = (blockIdx.x * 512u + threadIdx.x) * 2654435761u + 12345u;
    double a = 0.0;
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            s = s * 1664525u + 1013904223u;
            const unsigned q = (s >> 8) % M;
            if (MODE == 0) __hip_atomic_fetch_add(acc + q, (unsigned long long)(j + 1), __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
            else a += reinterpret_cast<const double*>(acc)[q];
        }
    }
    out[blockIdx.x * 512 + threadIdx.x] = a;
}

template <int MODE> void run(const char* name, unsigned M, int ncu, unsigned long long* dAcc, double* dOut)
{
    const int iters = 400;
    k<MODE><<<ncu, 512>>>(dAcc, M, 4, dOut);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE><<<ncu, 512>>>(dAcc, M, iters, dOut); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double ops = (double)ncu * 512 * iters * 8;
    printf("%-10s M=%8u (%7.1f KB): %.3f ms for %.1f M ops -> %.1f G ops/s\n", name, M, M * 8.0 / 1024, ms, ops / 1e6, ops / ms / 1e6);
}

int main()
{
    hipDeviceProp_t p; hipGetDeviceProperties(&p, 0);
    const int ncu = p.multiProcessorCount;
    unsigned long long* dAcc; double* dOut;
    hipMalloc(&dAcc, sizeof(unsigned long long) * (1u << 24)); hipMemset(dAcc, 0, sizeof(unsigned long long) * (1u << 24));
    hipMalloc(&dOut, sizeof(double) * ncu * 512);
    for (unsigned M : {2048u, 40000u, 80000u, 1u << 20, 1u << 24}) { run<0>("atomic u64", M, ncu, dAcc, dOut); run<1>("gather f64", M, ncu, dAcc, dOut); }
    return 0;
}
