// Stream-kernel probe for blend_kernel: out = w1*a + w2*b over 1e8 floats (24 B per f32 cell pair).  Variants: vectors in flight
// per lane (U), non-temporal on / off, grid size.   hipcc --offload-arch=gfx950 -O3 blend_probe.hip -o /tmp/blend_probe && /tmp/blend_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <vector>
typedef float VT __attribute__((ext_vector_type(4)));
template <int U, bool NT>
__global__ __launch_bounds__(256) void k(const VT* __restrict__ A, float w1, const VT* __restrict__ B, float w2, VT* __restrict__ O, long nvec) {
    const long stride = (long)gridDim.x * blockDim.x;
    for (long i = blockIdx.x * (long)blockDim.x + threadIdx.x; i < nvec; i += stride * U) {
        VT va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long j = i + u * stride; if (j < nvec) { va[u] = NT ? __builtin_nontemporal_load(A + j) : A[j]; vb[u] = NT ? __builtin_nontemporal_load(B + j) : B[j]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long j = i + u * stride; if (j < nvec) { VT r = w1 * va[u] + w2 * vb[u]; if (NT) __builtin_nontemporal_store(r, O + j); else O[j] = r; } }
    }
}
// block-contiguous variant: each block owns a contiguous chunk, lanes interleaved inside it
template <int U, bool NT>
__global__ __launch_bounds__(256) void kc(const VT* __restrict__ A, float w1, const VT* __restrict__ B, float w2, VT* __restrict__ O, long nvec) {
    const long per = 256L * U;
    for (long base = blockIdx.x * per; base < nvec; base += (long)gridDim.x * per) {
        VT va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) { const long j = base + u * 256 + threadIdx.x; if (j < nvec) { va[u] = NT ? __builtin_nontemporal_load(A + j) : A[j]; vb[u] = NT ? __builtin_nontemporal_load(B + j) : B[j]; } }
#pragma unroll
        for (int u = 0; u < U; ++u) { const long j = base + u * 256 + threadIdx.x; if (j < nvec) { VT r = w1 * va[u] + w2 * vb[u]; if (NT) __builtin_nontemporal_store(r, O + j); else O[j] = r; } }
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const long n = 100000000L, nvec = n / 4;
    float *a, *b, *o; hipMalloc(&a, n * 4); hipMalloc(&b, n * 4); hipMalloc(&o, n * 4);
    hipMemset(a, 0, n * 4); hipMemset(b, 0, n * 4);
    const double bytes = 3.0 * n * 4;
#define RUN(K, U, NT, G) { float ms = timeit([&] { hipLaunchKernelGGL((K<U, NT>), dim3(G), dim3(256), 0, 0, (const VT*)a, 0.25f, (const VT*)b, 0.75f, (VT*)o, nvec); }); \
    printf("%-3s U=%d nt=%d grid=%6d : %7.1f us  %6.2f TB/s\n", #K, U, (int)NT, (int)(G), ms * 1e3, bytes / ms / 1e9); }
    for (int g : {2048, 4096, 8192, 16384, 65536}) { RUN(k, 4, true, g); RUN(k, 4, false, g); RUN(k, 8, true, g); RUN(k, 2, true, g); RUN(kc, 4, true, g); RUN(kc, 8, true, g); RUN(kc, 4, false, g); }
    { const int g = (int)((nvec + 256 * 4 - 1) / (256 * 4)); RUN(kc, 4, true, g); RUN(kc, 4, false, g); RUN(k, 4, true, g); }
    { const int g = (int)((nvec + 256 * 8 - 1) / (256 * 8)); RUN(kc, 8, true, g); RUN(kc, 8, false, g); }
    { const int g = (int)((nvec + 256 * 2 - 1) / (256 * 2)); RUN(kc, 2, true, g); RUN(kc, 2, false, g); }
    { const int g = (int)((nvec + 255) / 256); RUN(kc, 1, true, g); RUN(kc, 1, false, g); }
    return 0;
}
