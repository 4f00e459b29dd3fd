// Microbenchmark: LDS gather (ds_read_b64) and LDS atomic scatter (ds_add_u64 / ds_add_f64, no return) throughput
// with random addresses inside a vector of L 8-byte elements, per CU, for 8 or 16 waves per workgroup.
// Models the inner step of a symmetric SpMV that stores only the upper triangle: per matrix entry one gather of
// x[q] and two scattered adds (M and C contributions of row p to column q).
// Build: hipcc --offload-arch=gfx950 -O3 -o lds_scatter lds_scatter.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>

template <int MODE, int NT>
__global__ void __launch_bounds__(NT) k(const unsigned* idx, int L, int iters, double* out, unsigned long long* cyc)
{
    extern __shared__ unsigned long long sm[];           // [L] x vector, [L] M accumulators, [L] C accumulators
    double* x = reinterpret_cast<double*>(sm);
    unsigned long long* am = sm + L; unsigned long long* ac = sm + 2 * L;
    for (int i = threadIdx.x; i < L; i += NT) { x[i] = 1.0 / (1 + i); am[i] = 0; ac[i] = 0; }
    __syncthreads();
    // per-thread index stream (random columns), held in registers: 8 per iteration group
    unsigned q[8];
    for (int j = 0; j < 8; ++j) q[j] = idx[(blockIdx.x * NT + threadIdx.x) * 8 + j] % (unsigned)L;
    double acc = 0.0;
    const unsigned long long t0 = __builtin_readcyclecounter();
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int j = 0; j < 8; ++j) {
            const unsigned qq = (q[j] + (unsigned)it * 97u) % (unsigned)L;
            if (MODE == 0 || MODE == 2 || MODE == 3) acc += x[qq];                                  // gather
            if (MODE == 1 || MODE == 2) { atomicAdd(&am[qq], (unsigned long long)(it + j + 1)); atomicAdd(&ac[qq], 3ull); }   // 2 u64 scatters
            if (MODE == 3) { atomicAdd(reinterpret_cast<double*>(&am[qq]), 0.25); atomicAdd(reinterpret_cast<double*>(&ac[qq]), 0.5); }   // 2 f64 scatters
            if (MODE == 4) { atomicAdd(&am[qq], (unsigned long long)(it + j + 1)); }                // 1 u64 scatter
        }
    }
    const unsigned long long t1 = __builtin_readcyclecounter();
    __syncthreads();
    if (threadIdx.x == 0) cyc[blockIdx.x] = t1 - t0;
    out[blockIdx.x * NT + threadIdx.x] = acc + (double)am[threadIdx.x % L] + (double)ac[threadIdx.x % L];
}

template <int MODE, int NT> void run(const char* name, int L, const unsigned* dIdx, double* dOut, unsigned long long* dCyc, int ncu)
{
    const int iters = 2000;
    const size_t lds = (size_t)3 * L * 8;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE, NT>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    k<MODE, NT><<<ncu, NT, lds>>>(dIdx, L, 10, dOut, dCyc);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    hipEventRecord(e0); k<MODE, NT><<<ncu, NT, lds>>>(dIdx, L, iters, dOut, dCyc); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    std::vector<unsigned long long> h(ncu); hipMemcpy(h.data(), dCyc, sizeof(unsigned long long) * ncu, hipMemcpyDeviceToHost);
    double c = 0; for (auto v : h) c += (double)v; c /= ncu;
    const double entries = (double)iters * 8 * NT;             // "matrix entries" processed per CU
    printf("%-22s NT=%4d L=%5d: %.3f ms, %.0f cycles/CU -> %.2f cycles per 64 entries (one wave-step), %.3f entries/clk/CU\n",
           name, NT, L, ms, c, c / (entries / 64.0), entries / c);
}

int main()
{
    int dev; hipGetDevice(&dev); hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int ncu = p.multiProcessorCount;
    std::vector<unsigned> hidx((size_t)ncu * 1024 * 8);
    srand(1); for (auto& v : hidx) v = (unsigned)rand();
    unsigned* dIdx; double* dOut; unsigned long long* dCyc;
    hipMalloc(&dIdx, hidx.size() * 4); hipMemcpy(dIdx, hidx.data(), hidx.size() * 4, hipMemcpyHostToDevice);
    hipMalloc(&dOut, (size_t)ncu * 1024 * 8); hipMalloc(&dCyc, (size_t)ncu * 8);
    for (int L : {128, 2100}) {
        run<0, 512>("gather", L, dIdx, dOut, dCyc, ncu);        run<0, 1024>("gather", L, dIdx, dOut, dCyc, ncu);
        run<4, 512>("1x add_u64", L, dIdx, dOut, dCyc, ncu);    run<4, 1024>("1x add_u64", L, dIdx, dOut, dCyc, ncu);
        run<1, 512>("2x add_u64", L, dIdx, dOut, dCyc, ncu);    run<1, 1024>("2x add_u64", L, dIdx, dOut, dCyc, ncu);
        run<2, 512>("gather+2x add_u64", L, dIdx, dOut, dCyc, ncu); run<2, 1024>("gather+2x add_u64", L, dIdx, dOut, dCyc, ncu);
        run<3, 512>("gather+2x add_f64", L, dIdx, dOut, dCyc, ncu); run<3, 1024>("gather+2x add_f64", L, dIdx, dOut, dCyc, ncu);
    }
    return 0;
}
