// Microbenchmark: how fast can ONE workgroup of 16 waves per compute unit read its problem's descriptor rows (config 3: 256 problems x
// 400 rows x 4 KB inside rows of 4160 B) — the read pattern of k_cos_sel's two passes, without any of their work?
//   mode 0  tile order: for each 1-KB quarter of the rows, wave w reads its rows w, w + 16, ... (DEPTH instructions in flight per wave)
//   mode 1  row order: wave w reads its rows one after the other, each whole (4 x 1 KB)
//   mode 2  chunk order: 128-B pieces, eight rows per instruction, all rows per 16-element chunk (pass 2)
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -o row_stream row_stream.hip
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
typedef double dbl2_t __attribute__((ext_vector_type(2)));
struct __attribute__((packed, aligned(8))) d2u_t { double v[2]; };

template <int MODE, int DEPTH>
__global__ void __launch_bounds__(1024) k_read(const double* __restrict__ feats, int rows, int F, int d, double* __restrict__ out)
{
    const int b = blockIdx.x, tid = threadIdx.x, lane = tid & 63, w = tid >> 6;
    const double* base = feats + (size_t)b * rows * F + 3;
    dbl2_t acc = {0.0, 0.0};
    if (MODE == 0) {
        const int nt = rows / 16;
        for (int k0 = 0; k0 < d; k0 += 128)
            for (int t0 = 0; t0 < nt; t0 += DEPTH) {
                dbl2_t x[DEPTH];
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) {
                    const int t = min(t0 + j, nt - 1);
                    const d2u_t v = *reinterpret_cast<const d2u_t*>(base + (size_t)(w + 16 * t) * F + k0 + 2 * lane);
                    x[j] = dbl2_t{v.v[0], v.v[1]};
                }
#pragma unroll
                for (int j = 0; j < DEPTH; ++j) acc += x[j];
            }
    } else if (MODE == 1) {
        const int nt = rows / 16;
        for (int t0 = 0; t0 < nt; t0 += DEPTH / 4) {
            dbl2_t x[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const int t = min(t0 + j / 4, nt - 1);
                const d2u_t v = *reinterpret_cast<const d2u_t*>(base + (size_t)(w + 16 * t) * F + 128 * (j & 3) + 2 * lane);
                x[j] = dbl2_t{v.v[0], v.v[1]};
            }
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) acc += x[j];
        }
    } else {
        const int pp = lane & 7, lr = 8 * w + (lane >> 3), nq = (rows + 127) / 128;
        for (int k0 = 0; k0 < d; k0 += 16 * (DEPTH / 4)) {
            dbl2_t x[DEPTH];
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) {
                const int q = j & 3, R = min(128 * q + lr, rows - 1);
                const d2u_t v = *reinterpret_cast<const d2u_t*>(base + (size_t)R * F + k0 + 16 * (j >> 2) + 2 * pp);
                x[j] = dbl2_t{v.v[0], v.v[1]};
            }
            (void)nq;
#pragma unroll
            for (int j = 0; j < DEPTH; ++j) acc += x[j];
        }
    }
    if (acc.x + acc.y == 1.2345e300) out[b * 1024 + tid] = acc.x;
}

int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, rows = 416, d = 512, F = 3 + d + 5;
    const size_t nf = (size_t)B * rows * F;
    double *feats, *out;
    CK(hipMalloc(&feats, nf * 8)); CK(hipMemset(feats, 0, nf * 8)); CK(hipMalloc(&out, (size_t)B * 1024 * 8));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    auto time_it = [&](const char* name, auto&& launch) {
        launch();
        hipEventRecord(e0);
        for (int r = 0; r < 10; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        const double us = ms * 1e3 / 10;
        printf("%-64s %7.1f us  %6.2f TB/s of the rows' %zu MB\n", name, us, (double)B * rows * d * 8 / us * 1e-6, (size_t)B * rows * d * 8 >> 20);
    };
#define RUN(MODE, DEPTH, label) time_it(label, [&]() { hipLaunchKernelGGL((k_read<MODE, DEPTH>), dim3(B), dim3(1024), 0, 0, feats, rows, F, d, out); })
    RUN(0, 2, "tile order, 1 KB per instruction, 2 in flight per wave");
    RUN(0, 4, "tile order, 1 KB per instruction, 4 in flight per wave");
    RUN(0, 8, "tile order, 1 KB per instruction, 8 in flight per wave");
    RUN(0, 13, "tile order, 1 KB per instruction, 13 in flight per wave");
    RUN(1, 4, "row order, whole rows, 4 in flight per wave");
    RUN(1, 8, "row order, whole rows, 8 in flight per wave");
    RUN(1, 16, "row order, whole rows, 16 in flight per wave");
    RUN(2, 4, "chunk order, 128 B x 8 rows per instruction, 4 in flight");
    RUN(2, 8, "chunk order, 128 B x 8 rows per instruction, 8 in flight");
    RUN(2, 16, "chunk order, 128 B x 8 rows per instruction, 16 in flight");
    CK(hipGetLastError()); CK(hipDeviceSynchronize());
    return 0;
}
