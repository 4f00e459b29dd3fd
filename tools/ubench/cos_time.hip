// Microbenchmark: the cosine kernels of the library on config 3's shape (256 problems of 200 x 200 objects, 512-d descriptors),
// alone on the device: k_cos_tile<16>, k_cos_deal, and k_cos_deal with parts stripped (template DBG) to see what the rest costs.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o cos_time cos_time.hip
#include "../../roman_amd/csrc/kernels.hip.h"
#include <cstdio>
#include <cstring>
#include <vector>
#include <algorithm>
using namespace roman;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
template <typename K, typename... A> static int run(const char* name, K kern, dim3 grid, size_t lds, A... a)
{
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a...);
    CK(hipGetLastError());
    hipEventRecord(e0);
    for (int r = 0; r < reps; ++r) hipLaunchKernelGGL(kern, grid, dim3(256), lds, 0, a...);
    hipEventRecord(e1); CK(hipEventSynchronize(e1));
    float ms; hipEventElapsedTime(&ms, e0, e1);
    printf("%-44s grid %5u  LDS %6zu B : %7.1f us per launch\n", name, grid.x, lds, ms * 1e3 / reps);
    return 0;
}
// Shader-clock probe: one wave per XCD samples clock64() (shader cycles) against wall_clock64() (constant rate) in windows while
// the kernel under test runs on another stream: the average shader clock the kernel runs at.
__global__ void k_probe(unsigned long long* out, int windows, unsigned long long ticksPerWindow)
{
    if (threadIdx.x != 0) return;
    for (int wdw = 0; wdw < windows; ++wdw) {
        const unsigned long long t0 = wall_clock64(), c0 = clock64();
        unsigned long long t1;
        do { __builtin_amdgcn_s_sleep(8); t1 = wall_clock64(); } while (t1 - t0 < ticksPerWindow);
        const unsigned long long c1 = clock64();
        out[(blockIdx.x * windows + wdw) * 2] = t1 - t0; out[(blockIdx.x * windows + wdw) * 2 + 1] = c1 - c0;
    }
}
template <typename K, typename... A> static int probe(const char* name, K kern, dim3 grid, size_t lds, A... a)
{
    int wallKhz = 100000; hipDeviceGetAttribute(&wallKhz, hipDeviceAttributeWallClockRate, 0);
    const int windows = 40; const unsigned long long tpw = (unsigned long long)wallKhz * 100 / 1000;     // 100 us windows
    unsigned long long* d; CK(hipMalloc(&d, 8 * windows * 2 * 8)); CK(hipMemset(d, 0, 8 * windows * 2 * 8));
    hipStream_t sa, sb; hipStreamCreateWithFlags(&sa, hipStreamNonBlocking); hipStreamCreateWithFlags(&sb, hipStreamNonBlocking);
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(kern), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds));
    CK(hipDeviceSynchronize());
    hipLaunchKernelGGL(k_probe, dim3(8), dim3(64), 0, sa, d, windows, tpw);
    for (int r = 0; r < 9; ++r) hipLaunchKernelGGL(kern, grid, dim3(256), lds, sb, a...);
    CK(hipDeviceSynchronize());
    unsigned long long h[8 * windows * 2]; CK(hipMemcpy(h, d, sizeof h, hipMemcpyDeviceToHost));
    printf("%-44s shader MHz per 100 us window (XCD of probe 0):", name);
    for (int wdw = 0; wdw < windows; wdw += 2) printf(" %.0f", (double)h[wdw * 2 + 1] / (double)h[wdw * 2] * wallKhz / 1000.0);
    double lo = 1e9; for (int x = 0; x < 8; ++x) for (int wdw = 5; wdw < 25; ++wdw) lo = std::min(lo, (double)h[(x * windows + wdw) * 2 + 1] / (double)h[(x * windows + wdw) * 2] * wallKhz / 1000.0);
    printf("  | min over XCDs, windows 5-24: %.0f\n", lo);
    hipFree(d); hipStreamDestroy(sa); hipStreamDestroy(sb);
    return 0;
}
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, n = argc > 2 ? atoi(argv[2]) : 200, d = argc > 3 ? atoi(argv[3]) : 512, F = 3 + d;
    DevParams D{}; D.p.cos_feature_dim = d; D.p.point_dim = 3; D.p.ratio_feature_dim = 0; D.F = F;
    std::vector<ProbDesc> hp(B);
    for (int b = 0; b < B; ++b) { hp[b] = ProbDesc{}; hp[b].off1 = (int64_t)2 * b * n; hp[b].off2 = hp[b].off1 + n; hp[b].n1 = n; hp[b].n2 = n; hp[b].cosOff = (int64_t)b * n * n; }
    ProbDesc* dP; double *feats, *cosPool;
    CK(hipMalloc(&dP, B * sizeof(ProbDesc))); CK(hipMemcpy(dP, hp.data(), B * sizeof(ProbDesc), hipMemcpyHostToDevice));
    const size_t nf = (size_t)2 * B * n * F;
    std::vector<double> hf(nf);
    uint64_t x = 88172645463325252ull;
    for (size_t i = 0; i < nf; ++i) { x ^= x << 13; x ^= x >> 7; x ^= x << 17; hf[i] = (double)(x >> 11) / 9007199254740992.0 - 0.5; }
    CK(hipMalloc(&feats, nf * 8)); CK(hipMemcpy(feats, hf.data(), nf * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&cosPool, (size_t)B * n * n * 8));
    auto ct = [](int n_) { return (((n_ + 15) >> 4) + 3) >> 2; };
    const int Gt = ct(n) * ct(n);
    const dim3 gt((unsigned)(Gt * ((B + 7) / 8) * 8));
    if (run("k_cos_tile<16>", k_cos_tile<16>, gt, (size_t)2 * 128 * (16 * 8 + 16), D, B, Gt, dP, feats, cosPool)) return 1;
    std::vector<double> ref((size_t)B * n * n), got(ref.size());
    CK(hipMemcpy(ref.data(), cosPool, ref.size() * 8, hipMemcpyDeviceToHost));
    auto check = [&](const char* name) {
        hipMemcpy(got.data(), cosPool, ref.size() * 8, hipMemcpyDeviceToHost);
        size_t bad = 0; for (size_t i = 0; i < ref.size(); ++i) bad += memcmp(&ref[i], &got[i], 8) != 0;
        printf("    %s vs k_cos_tile: %zu of %zu elements differ\n", name, bad, ref.size());
        hipMemset(cosPool, 0, ref.size() * 8);
    };
#define DEAL(T, DBG, label) { using CD = CosDeal<T>; const int Gd = CD::tiles(n) * CD::tiles(n); const dim3 gd((unsigned)(B >= 8 ? Gd * ((B + 7) / 8) * 8 : Gd * B)); \
        if (run(label, k_cos_deal<T, DBG>, gd, (size_t)CD::LDS, D, B, Gd, dP, feats, cosPool, (const int32_t*)nullptr)) return 1; }
    CK(hipMemset(cosPool, 0, ref.size() * 8));
    DEAL(7, 0, "k_cos_deal<7>"); check("k_cos_deal<7>");
    DEAL(5, 0, "k_cos_deal<5>"); check("k_cos_deal<5>");
    DEAL(6, 0, "k_cos_deal<6>"); check("k_cos_deal<6>");
    DEAL(4, 0, "k_cos_deal<4>"); check("k_cos_deal<4>");
    DEAL(7, 1, "k_cos_deal<7> no loads in the loop");
    DEAL(7, 3, "k_cos_deal<7> no loads, no stores/barriers");
    DEAL(7, 11, "k_cos_deal<7> no loads/stores/barriers/norms");
    DEAL(7, 8, "k_cos_deal<7> no norms");
    DEAL(7, 2, "k_cos_deal<7> no stores/barriers");
    DEAL(5, 1, "k_cos_deal<5> no loads in the loop");
    DEAL(5, 3, "k_cos_deal<5> no loads, no stores/barriers");
    DEAL(5, 11, "k_cos_deal<5> no loads/stores/barriers/norms");
    DEAL(5, 8, "k_cos_deal<5> no norms");
    DEAL(5, 2, "k_cos_deal<5> no stores/barriers");
#define PROBE(T, DBG, label) { using CD = CosDeal<T>; const int Gd = CD::tiles(n) * CD::tiles(n); const dim3 gd((unsigned)(B >= 8 ? Gd * ((B + 7) / 8) * 8 : Gd * B)); \
        if (probe(label, k_cos_deal<T, DBG>, gd, (size_t)CD::LDS, D, B, Gd, dP, feats, cosPool, (const int32_t*)nullptr)) return 1; }
    if (probe("k_cos_tile<16>", k_cos_tile<16>, gt, (size_t)2 * 128 * (16 * 8 + 16), D, B, Gt, dP, feats, cosPool)) return 1;
    PROBE(7, 0, "k_cos_deal<7>");
    PROBE(7, 11, "k_cos_deal<7> no loads/stores/barriers/norms");
    PROBE(5, 0, "k_cos_deal<5>");
    PROBE(5, 11, "k_cos_deal<5> no loads/stores/barriers/norms");
    return 0;
}
