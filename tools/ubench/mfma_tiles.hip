// Microbenchmark: what the launch geometry of the cosine kernel costs on the f64 matrix core, without any memory traffic.
// 256 problems x 16 tiles, one workgroup of 4 waves per tile, 32 stages x 16 MFMAs per full wave (config 3's shape).
//   MODE 0  every wave multiplies 2x2 blocks, no branches around the MFMAs
//   MODE 1  the same behind wave-uniform `if (onA[x] && onB[y])` branches
//   MODE 2  blocks as k_cos_tile's first tiling deals them at n = 200 (13 blocks per dimension: tiles of 4+4+4+1)
//   MODE 3  balanced tiling (3+3+3+4)
// Build: hipcc --offload-arch=gfx950 -O3 -o mfma_tiles mfma_tiles.hip
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double double4_t __attribute__((ext_vector_type(4)));
template <int MODE>
__global__ void __launch_bounds__(256) k(double* out, int stages, int ldsUse)
{
    extern __shared__ double smem[];
    const int tile = (blockIdx.x >> 3) & 15, w = threadIdx.x >> 6;
    const int ti = tile >> 2, tj = tile & 3;
    int nbx = 4, nby = 4;
    if (MODE == 2) { nbx = ti == 3 ? 1 : 4; nby = tj == 3 ? 1 : 4; }
    if (MODE == 3) { nbx = ti == 3 ? 4 : 3; nby = tj == 3 ? 4 : 3; }
    const int ws = (w + tile) & 3, wy = ws >> 1, wx = ws & 1;
    bool onA[2], onB[2];
    for (int h = 0; h < 2; ++h) {
        onA[h] = __builtin_amdgcn_readfirstlane(2 * wy + h < nbx) != 0;
        onB[h] = __builtin_amdgcn_readfirstlane(2 * wx + h < nby) != 0;
    }
    double4_t acc[2][2];
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) acc[x][y] = double4_t{0, 0, 0, 0};
    double a[2], b[2];
    a[0] = out[threadIdx.x]; a[1] = a[0] + 1.0; b[0] = out[threadIdx.x + 256]; b[1] = b[0] + 2.0;
    if (ldsUse) smem[threadIdx.x] = a[0];
    for (int s = 0; s < stages; ++s) {
#pragma unroll
        for (int t = 0; t < 4; ++t)
#pragma unroll
            for (int x = 0; x < 2; ++x)
#pragma unroll
                for (int y = 0; y < 2; ++y) {
                    if (MODE == 0) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
                    else if (onA[x] && onB[y]) acc[x][y] = __builtin_amdgcn_mfma_f64_16x16x4f64(a[x], b[y], acc[x][y], 0, 0, 0);
                }
    }
    double r = 0;
    for (int x = 0; x < 2; ++x) for (int y = 0; y < 2; ++y) r += acc[x][y][0] + acc[x][y][1] + acc[x][y][2] + acc[x][y][3];
    if (r == 12345.678) out[threadIdx.x] = r;
}
template <int MODE> void run(const char* name, double* d, size_t lds, double blocksPerProblem)
{
    const int grid = 256 * 16, stages = 32, reps = 20;
    hipFuncSetAttribute(reinterpret_cast<const void*>(k<MODE>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds);
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    k<MODE><<<grid, 256, lds>>>(d, stages, 1);
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) k<MODE><<<grid, 256, lds>>>(d, stages, 1);
    hipEventRecord(e1); hipEventSynchronize(e1);
    float ms; hipEventElapsedTime(&ms, e0, e1);
    const double us = ms * 1e3 / reps;
    const double mfmas = 256.0 * blocksPerProblem * stages * 4;
    printf("%-34s LDS %6zu B: %7.1f us per launch, %.2fM MFMAs -> %.1f cycles (2.4 GHz) per MFMA and SIMD, %.1f TFLOP/s\n", name, lds, us, mfmas / 1e6,
           us * 1e-6 * 2.4e9 * 1024 / mfmas, mfmas * 2048 / (us * 1e-6) / 1e12);
}
int main()
{
    double* d; hipMalloc(&d, 1024 * sizeof(double)); hipMemset(d, 0, 1024 * sizeof(double));
    for (size_t lds : {(size_t)36864, (size_t)1024}) {
        run<0>("uniform, no branches", d, lds, 256);
        run<1>("uniform, branches", d, lds, 256);
        run<2>("tiles 4+4+4+1 (169 blocks)", d, lds, 169);
        run<3>("tiles 3+3+3+4 (169 blocks)", d, lds, 169);
    }
    return 0;
}
