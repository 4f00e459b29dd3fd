// Shader-clock probe: one wave per XCD samples clock64() (shader cycles) against wall_clock64() (constant rate) in windows, for a
// given time, while something else (bench.py in another process, say) loads the device.  Prints, per interval of 250 ms, the
// minimum / median / maximum window average over the 8 XCDs: the clock the other kernels actually run at (nominal 2400 MHz).
//   usage: clock_probe [seconds = 10] [window_us = 500]          Build: hipcc --offload-arch=gfx950 -O3 -o clock_probe clock_probe.hip
#include <hip/hip_runtime.h>
#include <algorithm>
#include <cstdio>
#include <cstdlib>
#include <vector>
__global__ void k_probe(unsigned long long* out, int windows, unsigned long long ticksPerWindow)
{
    if (threadIdx.x != 0) return;
    for (int w = 0; w < windows; ++w) {
        const unsigned long long t0 = wall_clock64(), c0 = clock64();
        unsigned long long t1;
        do { __builtin_amdgcn_s_sleep(32); t1 = wall_clock64(); } while (t1 - t0 < ticksPerWindow);
        const unsigned long long c1 = clock64();
        out[((size_t)blockIdx.x * windows + w) * 2] = t1 - t0; out[((size_t)blockIdx.x * windows + w) * 2 + 1] = c1 - c0;
    }
}
int main(int argc, char** argv)
{
    const double seconds = argc > 1 ? atof(argv[1]) : 10.0; const int windowUs = argc > 2 ? atoi(argv[2]) : 500;
    int wallKhz = 100000; hipDeviceGetAttribute(&wallKhz, hipDeviceAttributeWallClockRate, 0);
    const int perLaunch = (int)(250000 / windowUs), launches = (int)(seconds * 4);
    unsigned long long* d; hipMalloc(&d, (size_t)8 * perLaunch * 16);
    std::vector<unsigned long long> h((size_t)8 * perLaunch * 2);
    for (int l = 0; l < launches; ++l) {
        hipLaunchKernelGGL(k_probe, dim3(8), dim3(64), 0, 0, d, perLaunch, (unsigned long long)wallKhz * windowUs / 1000);
        hipMemcpy(h.data(), d, h.size() * 8, hipMemcpyDeviceToHost);
        std::vector<double> mhz;
        for (size_t i = 0; i < h.size(); i += 2) if (h[i]) mhz.push_back((double)h[i + 1] / (double)h[i] * wallKhz / 1000.0);
        std::sort(mhz.begin(), mhz.end());
        printf("t = %5.2f s: shader MHz min %4.0f  p10 %4.0f  median %4.0f  max %4.0f\n", l * 0.25, mhz.front(), mhz[mhz.size() / 10], mhz[mhz.size() / 2], mhz.back());
        fflush(stdout);
    }
    return 0;
}
