// Microbenchmark + check: k_cos_sel (bf16 screen + exact f64 candidates) against k_cos_deal<5> on config 3's shape (256 problems of
// 200 x 200 objects, 512-d descriptors), alone on the device.  Descriptors: a shared direction plus noise, so that the cosines spread
// around a level the thresholds below cut at a few candidate densities.
// Checks: every element >= thr + 0.0042 (surely a candidate... in fact every candidate) is bit-identical to k_cos_deal's, every other
// element lies within 0.0042 of it; with thr = +inf the whole matrix is the screen's and lies within 0.0042.
// Build: hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -o cos_sel_time cos_sel_time.hip
#include "../../roman_amd/csrc/kernels.hip.h"
#include <cstdio>
#include <cstring>
#include <cmath>
#include <vector>
#include <algorithm>
using namespace roman;
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)
int main(int argc, char** argv)
{
    const int B = argc > 1 ? atoi(argv[1]) : 256, n = argc > 2 ? atoi(argv[2]) : 200, d = argc > 3 ? atoi(argv[3]) : 512;
    const int F = 3 + d + (argc > 4 ? atoi(argv[4]) : 5);
    DevParams D{}; D.p.cos_feature_dim = d; D.p.point_dim = 3; D.p.ratio_feature_dim = 0; D.F = F;
    std::vector<ProbDesc> hp(B);
    for (int b = 0; b < B; ++b) { hp[b] = ProbDesc{}; hp[b].off1 = (int64_t)2 * b * n; hp[b].off2 = hp[b].off1 + n; hp[b].n1 = n; hp[b].n2 = n; hp[b].cosOff = (int64_t)b * n * n; }
    ProbDesc* dP; double *feats, *cosPool; int32_t* dense;
    CK(hipMalloc(&dP, B * sizeof(ProbDesc))); CK(hipMemcpy(dP, hp.data(), B * sizeof(ProbDesc), hipMemcpyHostToDevice));
    CK(hipMalloc(&dense, B * 4));
    const size_t nf = (size_t)2 * B * n * F;
    std::vector<double> hf(nf), dir(d);
    uint64_t x = 88172645463325252ull;
    auto rnd = [&]() { x ^= x << 13; x ^= x >> 7; x ^= x << 17; return (double)(x >> 11) / 9007199254740992.0 - 0.5; };
    for (int k = 0; k < d; ++k) dir[k] = rnd();
    for (size_t row = 0; row < nf / F; ++row) {
        const double wgt = 0.6 + 0.8 * (rnd() + 0.5);          // share of the common direction: cosines between ~0.25 and ~0.65
        for (int k = 0; k < F; ++k) hf[row * F + k] = rnd() + (k >= 3 && k < 3 + d ? wgt * dir[k - 3] : 0.0);
    }
    CK(hipMalloc(&feats, nf * 8)); CK(hipMemcpy(feats, hf.data(), nf * 8, hipMemcpyHostToDevice));
    CK(hipMalloc(&cosPool, (size_t)B * n * n * 8));
    using CD = CosDeal<5>;
    const int Gd = CD::tiles(n) * CD::tiles(n);
    const dim3 gd((unsigned)(B >= 8 ? Gd * ((B + 7) / 8) * 8 : Gd * B));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cos_deal<5, 0>), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CD::LDS));
    CK(hipFuncSetAttribute(reinterpret_cast<const void*>(k_cos_sel), hipFuncAttributeMaxDynamicSharedMemorySize, (int)CSEL_LDS));
    hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
    const int reps = 10;
    auto time_it = [&](const char* name, auto&& launch) {
        launch();
        hipEventRecord(e0);
        for (int r = 0; r < reps; ++r) launch();
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        printf("%-52s %7.1f us per launch\n", name, ms * 1e3 / reps);
    };
    time_it("k_cos_deal<5>", [&]() { hipLaunchKernelGGL((k_cos_deal<5, 0>), gd, dim3(256), (size_t)CD::LDS, 0, D, B, Gd, dP, feats, cosPool, (const int32_t*)nullptr); });
    CK(hipGetLastError());
    std::vector<double> ref((size_t)B * n * n), got(ref.size());
    CK(hipMemcpy(ref.data(), cosPool, ref.size() * 8, hipMemcpyDeviceToHost));
    { std::vector<double> srt(ref.begin(), ref.begin() + (size_t)n * n); std::sort(srt.begin(), srt.end());
      printf("cosines of problem 0: min %.3f  p50 %.3f  p90 %.3f  p95 %.3f  p99 %.3f  max %.3f\n", srt[0], srt[srt.size() / 2], srt[srt.size() * 9 / 10], srt[srt.size() * 95 / 100], srt[srt.size() * 99 / 100], srt.back()); }
    std::vector<double> srtAll(ref); std::sort(srtAll.begin(), srtAll.end());
    std::vector<double> fr = {0.0, 0.02, 0.05, 0.10, 0.14, 0.25};
    if (const char* only = getenv("SEL_ONLY")) { fr.clear(); for (const char* q = only; *q; ) { fr.push_back(atof(q)); while (*q && *q != ',') ++q; if (*q) ++q; } }
    for (double f : fr) {
        const double thr = f == 0.0 ? INFINITY : srtAll[(size_t)((1.0 - f) * (srtAll.size() - 1))];
        CK(hipMemset(cosPool, 0xff, ref.size() * 8));
        char name[96]; snprintf(name, sizeof name, "k_cos_sel, thr %.4f (~%.0f %% candidates)", thr, 100 * f);
        time_it(name, [&]() { hipLaunchKernelGGL(k_cos_sel, dim3(B), dim3(1024), (size_t)CSEL_LDS, 0, D, B, dP, feats, cosPool, dense, thr); });
        CK(hipGetLastError()); CK(hipDeviceSynchronize());
        snprintf(name, sizeof name, "  + k_cos_deal<5> for the flagged problems");
        std::vector<int32_t> hd(B); CK(hipMemcpy(hd.data(), dense, B * 4, hipMemcpyDeviceToHost));
        int nd = 0; for (int b = 0; b < B; ++b) nd += hd[b];
        CK(hipMemcpy(got.data(), cosPool, ref.size() * 8, hipMemcpyDeviceToHost));
        size_t exact = 0, bad = 0, miss = 0; double worst = 0.0;
        for (int b = 0; b < B; ++b) {
            if (hd[b]) continue;
            for (size_t e = (size_t)b * n * n; e < (size_t)(b + 1) * n * n; ++e) {
                const bool same = memcmp(&ref[e], &got[e], 8) == 0;
                exact += same;
                if (!same) { const double df = fabs(ref[e] - got[e]); worst = std::max(worst, df); if (!(df <= 0.0042)) ++bad; if (!(ref[e] < thr)) ++miss; }
            }
        }
        printf("    flagged %d of %d problems; of the others' elements: %zu bit-identical (%.1f %%), |screen - exact| max %.5f, %zu beyond 0.0042, %zu not below thr yet inexact\n",
               nd, B, exact, 100.0 * exact / ((double)(B - nd) * n * n + 1e-9), worst, bad, miss);
        time_it(name, [&]() { hipLaunchKernelGGL((k_cos_deal<5, 0>), gd, dim3(256), (size_t)CD::LDS, 0, D, B, Gd, dP, feats, cosPool, (const int32_t*)dense); });
    }
    return 0;
}
