// Microbenchmark: issue cost of v_mfma_f64_16x16x4_f64 (4 independent accumulators, chains of 4 dependent ones, with v_fma_f64
// mixed in: the f64 vector FMA runs on the units the f64 MFMA occupies) vs v_fma_f64 with an SGPR operand.  Cycles per 16 MFMAs / 16.  Build: hipcc --offload-arch=gfx950 -O3 -o mfma_rate mfma_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, double s, int iters)
{
    double4_t acc[4];
    for (int i = 0; i < 4; ++i) acc[i] = double4_t{0, 0, 0, 0};
    double a = out[threadIdx.x], b = out[threadIdx.x + 256];
    double f[16];
    for (int i = 0; i < 16; ++i) f[i] = (OP == 3) ? out[512 + 16 * threadIdx.x + i] : a + i;
    for (int it = 0; it < iters; ++it) {
        if (OP == 0) {
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        } else if (OP == 3) {                   // 8 distinct operand registers holding random values (data-dependent power)
#pragma unroll
            for (int r = 0; r < 4; ++r)
#pragma unroll
                for (int i = 0; i < 4; ++i) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(f[r + 4 * (i >> 1)], f[8 + r + 4 * (i & 1)], acc[i], 0, 0, 0);
        } else if (OP == 4) {                   // chains of 4 DEPENDENT MFMAs (one accumulator at a time: the cosine kernels' block-by-block order)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
        } else if (OP == 5) {                   // 16 MFMAs (chains of 4), then a burst of 4 dependent v_fma_f64 (the norms of k_cos_deal: 14 per 52 MFMAs)
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[0]) : "s"(s), "v"(b));
        } else if (OP == 6) {                   // the same 4 v_fma_f64, one behind every chain of 4 MFMAs
#pragma unroll
            for (int i = 0; i < 4; ++i) {
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
                asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[0]) : "s"(s), "v"(b));
            }
        } else if (OP == 7) {                   // 16 MFMAs, then 4 INDEPENDENT v_fma_f64
#pragma unroll
            for (int i = 0; i < 4; ++i)
#pragma unroll
                for (int r = 0; r < 4; ++r) acc[i] = __builtin_amdgcn_mfma_f64_16x16x4f64(a, b, acc[i], 0, 0, 0);
#pragma unroll
            for (int i = 0; i < 4; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[i]) : "s"(s), "v"(b));
        } else if (OP == 1) {
#pragma unroll
            for (int i = 0; i < 16; ++i) asm volatile("v_fma_f64 %0, %1, %2, %0" : "+v"(f[i]) : "s"(s), "v"(b));
        } else {
#pragma unroll
            for (int i = 0; i < 16; ++i) { asm volatile("v_mul_f64 %0, %1, %2" : "=v"(a) : "s"(s), "v"(b)); asm volatile("v_add_f64 %0, %0, %1" : "+v"(f[i]) : "v"(a)); }
        }
    }
    double r = 0;
    for (int i = 0; i < 4; ++i) r += acc[i][0] + acc[i][1] + acc[i][2] + acc[i][3];
    for (int i = 0; i < 16; ++i) r += f[i];
    out[threadIdx.x] = r;
}
template <int OP> void run(const char* name, double* d, int wavesPerSimd, double flop_per_inst)
{
    int dev; hipGetDevice(&dev); hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int iters = 20000;
    dim3 grid(p.multiProcessorCount * wavesPerSimd), block(256);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, block>>>(d, 1.0000001, 10);
    hipEventRecord(e0); k<OP><<<grid, block>>>(d, 1.0000001, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts = (double)iters * 16 * wavesPerSimd;        // per SIMD
    const double clk = p.clockRate * 1e3;
    printf("%-22s waves/SIMD=%d  %.3f ms -> %.1f cycles/inst/SIMD, %.1f TFLOP/s chip\n", name, wavesPerSimd, ms, ms * 1e-3 * clk / insts,
           insts * p.multiProcessorCount * 4 * flop_per_inst / (ms * 1e-3) / 1e12);
}
int main()
{
    const int N = 512 + 16 * 256;
    double* d; hipMalloc(&d, N * sizeof(double)); hipMemset(d, 0, N * sizeof(double));
    {   // random operands in [-1, 1) for the data-dependent variant
        double* h = new double[N]; unsigned long long x = 88172645463325252ull;
        for (int i = 0; i < N; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; h[i] = i < 512 ? 0.0 : (double)(x >> 11) / 4503599627370496.0 - 1.0; }
        hipMemcpy(d, h, N * sizeof(double), hipMemcpyHostToDevice); delete[] h;
    }
    for (int w : {1, 2, 4}) {
        run<0>("mfma_f64_16x16x4", d, w, 2048.0);
        run<3>("mfma_f64 random data", d, w, 2048.0);
        run<4>("mfma chains of 4", d, w, 2048.0);
        run<5>("16 mfma + 4 dep. v_fma burst", d, w, 2048.0);
        run<6>("4 x (4 mfma + v_fma)", d, w, 2048.0);
        run<7>("16 mfma + 4 indep. v_fma", d, w, 2048.0);
        run<1>("v_fma_f64 (sgpr src)", d, w, 128.0);
        run<2>("v_mul_f64+v_add_f64", d, w, 128.0);
    }
    return 0;
}
