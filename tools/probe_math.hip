// Probe: accuracy of v_rsq_f64 / v_rcp_f64 seeds and Newton-Raphson refinements on gfx950.
// build: hipcc --offload-arch=gfx950 -O3 tools/probe_math.hip -o gpurun_out/probe_math ; run on the GPU box.
#include <hip/hip_runtime.h>
#include <cmath>
#include <cstdio>
#include <random>
#include <vector>

template <int NR> __device__ double rsq_nr(double a) {
    double y = __builtin_amdgcn_rsq(a);
    for (int i = 0; i < NR; ++i) { double t = a * y; double e = fma(-t, y, 1.0); y = fma(0.5 * y, e, y); }
    return y;
}
template <int NR> __device__ double rcp_nr(double b) {
    double r = __builtin_amdgcn_rcp(b);
    for (int i = 0; i < NR; ++i) { double e = fma(-b, r, 1.0); r = fma(r, e, r); }
    return r;
}
__global__ void k(const double* in, int n, double* out) {
    int i = blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double a = in[i];
    out[0 * n + i] = rsq_nr<0>(a); out[1 * n + i] = rsq_nr<1>(a); out[2 * n + i] = rsq_nr<2>(a);
    out[3 * n + i] = rcp_nr<0>(a); out[4 * n + i] = rcp_nr<1>(a); out[5 * n + i] = rcp_nr<2>(a);
    out[6 * n + i] = (double)__builtin_amdgcn_rsqf((float)a); out[7 * n + i] = (double)__builtin_amdgcn_rcpf((float)a);
}
int main() {
    const int n = 1 << 20;
    std::vector<double> h(n);
    std::mt19937_64 g(1);
    std::uniform_real_distribution<double> u(0.0, 1.0);
    for (int i = 0; i < n; ++i) {
        int r = i % 4;
        if (r == 0) h[i] = 0.99 + 0.0101 * u(g);               // w = 1 - es sin^2
        else if (r == 1) h[i] = 1e13 * (1.0 + 3.2 * u(g));      // p^2, n2phi
        else if (r == 2) h[i] = 1e27 * (1.0 + 9 * u(g));        // theta-stage norm^2
        else h[i] = std::exp(60 * (u(g) - 0.5));
    }
    double *di, *dout;
    hipMalloc(&di, n * 8); hipMalloc(&dout, 8 * n * 8);
    hipMemcpy(di, h.data(), n * 8, hipMemcpyHostToDevice);
    hipLaunchKernelGGL(k, dim3(n / 256), dim3(256), 0, 0, di, n, dout);
    std::vector<double> o(8 * n);
    hipMemcpy(o.data(), dout, 8 * n * 8, hipMemcpyDeviceToHost);
    const char* names[8] = {"rsq seed", "rsq nr1", "rsq nr2", "rcp seed", "rcp nr1", "rcp nr2", "rsqf(f32)", "rcpf(f32)"};
    for (int v = 0; v < 8; ++v) {
        long double worst = 0;
        for (int i = 0; i < n; ++i) {
            long double a = (v >= 6) ? (long double)(float)h[i] : (long double)h[i];
            long double ref = (v % 8 < 3 || v == 6) ? 1.0L / sqrtl(a) : 1.0L / a;
            long double e = fabsl(((long double)o[v * n + i] - ref) / ref);
            if (e > worst) worst = e;
        }
        printf("%-10s max rel err = %.3Le  (2^%.1Lf)\n", names[v], worst, log2l(worst));
    }
    return 0;
}
