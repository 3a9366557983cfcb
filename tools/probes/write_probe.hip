// Write-side probe for build_cube_kernel (config 2: 2 x 40 x 1000 x 1000 doubles = 640 MB of output, nothing else reaches HBM):
// what does the chip sustain for PURE writes, and in which access pattern?   hipcc --offload-arch=gfx950 -O3 write_probe.hip -o /tmp/wp && /tmp/wp
//   lin      : linear fill, 16 B per lane, grid covering the array once
//   node     : the kernel's pattern - one thread per node, 40 heights x 2 arrays, 8 B stores, consecutive lanes = consecutive nodes
//   node2    : two adjacent nodes per thread, 16 B stores
//   plane    : blockIdx.y = height: the resident workgroups write a few height planes at a time
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double D2 __attribute__((ext_vector_type(2)));
template <bool NT> __global__ __launch_bounds__(256) void lin(D2* o, long nvec) {
    const long i = blockIdx.x * 256L + threadIdx.x;
    if (i < nvec) { D2 v = {1.0 * i, 2.0}; if (NT) __builtin_nontemporal_store(v, o + i); else o[i] = v; }
}
template <bool NT> __global__ __launch_bounds__(256) void node(double* w, double* h, long nodes, int nz) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nodes; i += gridDim.x * 256L)
        for (int z = 0; z < nz; ++z) { const long o = z * nodes + i; const double a = 1.0 * i + z, b = 2.0 * i;
            if (NT) { __builtin_nontemporal_store(a, w + o); __builtin_nontemporal_store(b, h + o); } else { w[o] = a; h[o] = b; } }
}
template <bool NT> __global__ __launch_bounds__(256) void node2(double* w, double* h, long nodes, int nz) {
    for (long i = (blockIdx.x * 256L + threadIdx.x) * 2; i < nodes; i += gridDim.x * 512L)
        for (int z = 0; z < nz; ++z) { const long o = z * nodes + i; D2 a = {1.0 * i + z, 1.0}, b = {2.0 * i, 3.0};
            if (NT) { __builtin_nontemporal_store(a, (D2*)(w + o)); __builtin_nontemporal_store(b, (D2*)(h + o)); } else { *(D2*)(w + o) = a; *(D2*)(h + o) = b; } }
}
template <bool NT> __global__ __launch_bounds__(256) void plane(double* w, double* h, long nodes, int zchunk) {
    const int z0 = blockIdx.y * zchunk;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < nodes; i += gridDim.x * 256L)
        for (int z = z0; z < z0 + zchunk; ++z) { const long o = z * nodes + i; const double a = 1.0 * i + z, b = 2.0 * i;
            if (NT) { __builtin_nontemporal_store(a, w + o); __builtin_nontemporal_store(b, h + o); } else { w[o] = a; h[o] = b; } }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const long nodes = 1000000; const int nz = 40;
    double *w, *h; hipMalloc(&w, nodes * nz * 8); hipMalloc(&h, nodes * nz * 8);
    const double bytes = 2.0 * nodes * nz * 8;
#define P(name, ms) printf("%-28s %7.1f us  %5.2f TB/s\n", name, (ms) * 1e3, bytes / (ms) / 1e9)
    { const long nvec = nodes * nz / 2; const int g = (int)((nvec + 255) / 256);
      P("lin plain (one array x2)", 2 * timeit([&] { hipLaunchKernelGGL(lin<false>, dim3(g), dim3(256), 0, 0, (D2*)w, nvec); }));
      P("lin nt", 2 * timeit([&] { hipLaunchKernelGGL(lin<true>, dim3(g), dim3(256), 0, 0, (D2*)w, nvec); })); }
    for (int g : {2048, 3907}) {
        char nm[64];
        snprintf(nm, 64, "node plain grid %d", g); P(nm, timeit([&] { hipLaunchKernelGGL(node<false>, dim3(g), dim3(256), 0, 0, w, h, nodes, nz); }));
        snprintf(nm, 64, "node nt grid %d", g); P(nm, timeit([&] { hipLaunchKernelGGL(node<true>, dim3(g), dim3(256), 0, 0, w, h, nodes, nz); }));
        snprintf(nm, 64, "node2 plain grid %d", g / 2); P(nm, timeit([&] { hipLaunchKernelGGL(node2<false>, dim3(g / 2), dim3(256), 0, 0, w, h, nodes, nz); }));
        snprintf(nm, 64, "node2 nt grid %d", g / 2); P(nm, timeit([&] { hipLaunchKernelGGL(node2<true>, dim3(g / 2), dim3(256), 0, 0, w, h, nodes, nz); }));
    }
    for (int zc : {1, 2, 4, 8, 20}) {
        char nm[64];
        snprintf(nm, 64, "plane plain zchunk %d", zc); P(nm, timeit([&] { hipLaunchKernelGGL(plane<false>, dim3(3907, nz / zc), dim3(256), 0, 0, w, h, nodes, zc); }));
        snprintf(nm, 64, "plane nt zchunk %d", zc); P(nm, timeit([&] { hipLaunchKernelGGL(plane<true>, dim3(3907, nz / zc), dim3(256), 0, 0, w, h, nodes, zc); }));
        snprintf(nm, 64, "plane nt zchunk %d g512", zc); P(nm, timeit([&] { hipLaunchKernelGGL(plane<true>, dim3(512, nz / zc), dim3(256), 0, 0, w, h, nodes, zc); }));
    }
    return 0;
}
