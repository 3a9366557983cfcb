// Is s_memtime (clock64) the shader clock on gfx950?  Every workgroup runs dependent fp64 FMA chains (the march kernel's kind of
// load) and reads clock64() / wall_clock64() (s_memrealtime, 100 MHz) around them: cycles per 10 ns tick -> GHz.
// hipcc --offload-arch=gfx950 -O3 clock_probe.hip -o clock_probe && ./clock_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
__global__ __launch_bounds__(256) void probe(long long iters, double seed, long long* out, double* sink) {
    double a = seed + threadIdx.x, b = 1.0000001, c = 1e-9, d = a + 1, e = a + 2, f = a + 3;
    const long long c0 = clock64(), w0 = wall_clock64();
    for (long long i = 0; i < iters; ++i) {
        a = __builtin_fma(a, b, c); d = __builtin_fma(d, b, c); e = __builtin_fma(e, b, c); f = __builtin_fma(f, b, c);
    }
    const long long c1 = clock64(), w1 = wall_clock64();
    if (threadIdx.x == 0) { out[2 * blockIdx.x] = c1 - c0; out[2 * blockIdx.x + 1] = w1 - w0; }
    if (a + d + e + f == 12345.678) *sink = a;
}
int main() {
    int wall_khz = 0;
    hipDeviceGetAttribute(&wall_khz, hipDeviceAttributeWallClockRate, 0);
    int clk_khz = 0;
    hipDeviceGetAttribute(&clk_khz, hipDeviceAttributeClockRate, 0);
    const int G = 256 * 4;
    long long* d; double* s;
    hipMalloc(&d, G * 2 * sizeof(long long)); hipMalloc(&s, 8);
    for (long long iters : {20000LL, 200000LL, 1000000LL}) {
        hipEvent_t e0, e1; hipEventCreate(&e0); hipEventCreate(&e1);
        hipEventRecord(e0);
        hipLaunchKernelGGL(probe, dim3(G), dim3(256), 0, 0, iters, 1.0, d, s);
        hipEventRecord(e1); hipEventSynchronize(e1);
        float ms; hipEventElapsedTime(&ms, e0, e1);
        std::vector<long long> h(G * 2);
        hipMemcpy(h.data(), d, G * 2 * sizeof(long long), hipMemcpyDeviceToHost);
        double sc = 0, sw = 0; for (int i = 0; i < G; ++i) { sc += h[2 * i]; sw += h[2 * i + 1]; }
        printf("iters %lld: kernel %.3f ms, mean clock64 delta %.0f, mean wall delta %.0f (wall rate %d kHz, attr clock %d kHz) -> clock64 ticks per us %.2f, fma/clock64-tick per wave %.3f\n",
               iters, ms, sc / G, sw / G, wall_khz, clk_khz, (sc / G) / ((sw / G) / (wall_khz / 1000.0)), 4.0 * iters / (sc / G));
    }
    return 0;
}
