// Microbenchmark: issue cost (cycles per wave64 instruction per SIMD) of the f64 / int VALU ops
// the pair-test kernel uses.  Build: hipcc --offload-arch=gfx950 -O3 -o valu_rate valu_rate.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
template <int OP>
__global__ void __launch_bounds__(256) k(double* out, double s, int iters)
{
    double a[8];
    for (int i = 0; i < 8; ++i) a[i] = out[threadIdx.x + i * 256] ;
    unsigned u[8];
    for (int i = 0; i < 8; ++i) u[i] = (unsigned)a[i];
    for (int it = 0; it < iters; ++it) {
#pragma unroll
        for (int r = 0; r < 8; ++r)
#pragma unroll
        for (int i = 0; i < 8; ++i) {
            if (OP == 0) asm volatile("v_add_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (OP == 1) asm volatile("v_mul_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (OP == 2) asm volatile("v_fma_f64 %0, %0, %1, %1" : "+v"(a[i]) : "v"(s));
            if (OP == 3) asm volatile("v_max_f64 %0, %0, %1" : "+v"(a[i]) : "v"(s));
            if (OP == 4) asm volatile("v_cmp_gt_f64 vcc, %0, %1" :: "v"(a[i]), "v"(s) : "vcc");
            if (OP == 5) asm volatile("v_add_u32 %0, %0, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (OP == 6) asm volatile("v_fma_f32 %0, %0, %1, %1" : "+v"(u[i]) : "v"(u[(i + 1) & 7]));
            if (OP == 7) asm volatile("v_cndmask_b32 %0, %0, %1, vcc" : "+v"(u[i]) : "v"(u[(i + 1) & 7]) : "vcc");
            if (OP == 8) asm volatile("v_writelane_b32 %0, s0, 3" : "+v"(u[i]));
        }
    }
    double r = 0; for (int i = 0; i < 8; ++i) r += a[i] + u[i];
    out[threadIdx.x] = r;
}
template <int OP> void run(const char* name, double* d, int wavesPerSimd)
{
    int dev; hipGetDevice(&dev); hipDeviceProp_t p; hipGetDeviceProperties(&p, dev);
    const int iters = 20000;
    dim3 grid(p.multiProcessorCount * wavesPerSimd), block(256);       // 256 threads = 1 wave per SIMD per block
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<OP><<<grid, block>>>(d, 1.0000001, 10);
    hipEventRecord(e0); k<OP><<<grid, block>>>(d, 1.0000001, iters); hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double insts_per_simd = (double)iters * 64 * wavesPerSimd;
    const double clk = p.clockRate * 1e3;      // Hz
    printf("%-14s waves/SIMD=%d  %.3f ms  -> %.2f cycles/inst/SIMD (at %.0f MHz nominal)\n", name, wavesPerSimd, ms, ms * 1e-3 * clk / insts_per_simd, clk / 1e6);
}
int main()
{
    double* d; hipMalloc(&d, 256 * 8 * sizeof(double)); hipMemset(d, 0, 256 * 8 * sizeof(double));
    for (int w : {1, 4}) {
        run<0>("v_add_f64", d, w); run<1>("v_mul_f64", d, w); run<2>("v_fma_f64", d, w); run<3>("v_max_f64", d, w);
        run<4>("v_cmp_gt_f64", d, w); run<5>("v_add_u32", d, w); run<6>("v_fma_f32", d, w); run<7>("v_cndmask_b32", d, w); run<8>("v_writelane", d, w);
    }
    return 0;
}
