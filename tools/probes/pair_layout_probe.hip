// VERDICT r4 item 7 probe: one-shot station gathers at fewer bytes.  5 M random points on a 1000 x 1000 x 50 f32 (wet, hydro) cube, trilinear
// gather of both fields (the arithmetic of interp_points_kernel), from three layouts:
//   L0  (y, x, z) interleaved pairs, z fastest                       - what the cube is: 4 x 128 B lines per point
//   L1  x-columns paired:  [(iy, ix/2), z, (x parity, field)]        - 1 x the cube; even ix: both x columns of a y row in 32 contiguous B
//   L2  every (ix, ix+1) pair stored: [(iy, ix), z, (x side, field)] - 2 x the cube; always 32 contiguous B per y row
// plus the cost of MAKING L1 / L2 from L0 (a one-shot call pays it).  hipcc --offload-arch=gfx950 -O3 pair_layout_probe.hip -o pair_layout_probe
#include <hip/hip_runtime.h>
#include <cstdio>
#include <cstdlib>
#include <vector>
#include <random>
#define CK(x) do { hipError_t e_ = (x); if (e_ != hipSuccess) { printf("%s: %s\n", #x, hipGetErrorString(e_)); return 1; } } while (0)

constexpr int NY = 1000, NX = 1000, NZ = 50;

__device__ __forceinline__ void finish(const float2* v, double ty, double tx, double tz, double& w, double& h) {
    auto lerp = [](double a, double b, double t) { return fma(t, b - a, a); };
    const double w00 = lerp(v[0].x, v[1].x, tz), h00 = lerp(v[0].y, v[1].y, tz), w01 = lerp(v[2].x, v[3].x, tz), h01 = lerp(v[2].y, v[3].y, tz);
    const double w10 = lerp(v[4].x, v[5].x, tz), h10 = lerp(v[4].y, v[5].y, tz), w11 = lerp(v[6].x, v[7].x, tz), h11 = lerp(v[6].y, v[7].y, tz);
    w = lerp(lerp(w00, w01, tx), lerp(w10, w11, tx), ty); h = lerp(lerp(h00, h01, tx), lerp(h10, h11, tx), ty);
}

__device__ __forceinline__ void cell(const double* p, int& iy, int& ix, int& iz, double& ty, double& tx, double& tz) {
    const double fy = p[0], fx = p[1], fz = p[2];            // already in index space [0, n-1)
    iy = (int)fy; ix = (int)fx; iz = (int)fz; ty = fy - iy; tx = fx - ix; tz = fz - iz;
}

__global__ __launch_bounds__(256) void gather_l0(const float2* c, const double* pts, long n, double* ow, double* oh) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        int iy, ix, iz; double ty, tx, tz; cell(pts + 3 * i, iy, ix, iz, ty, tx, tz);
        const float2* p00 = c + ((long)iy * NX + ix) * NZ + iz; const float2* p01 = p00 + NZ; const float2* p10 = p00 + (long)NX * NZ; const float2* p11 = p10 + NZ;
        float2 v[8] = {p00[0], p00[1], p01[0], p01[1], p10[0], p10[1], p11[0], p11[1]};
        double w, h; finish(v, ty, tx, tz, w, h); ow[i] = w; oh[i] = h;
    }
}
// L1: float4 per (pair, z) = (even.wet, even.hyd, odd.wet, odd.hyd)
__global__ __launch_bounds__(256) void gather_l1(const float4* c, const double* pts, long n, double* ow, double* oh) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        int iy, ix, iz; double ty, tx, tz; cell(pts + 3 * i, iy, ix, iz, ty, tx, tz);
        float2 v[8];
        const int px = ix >> 1;
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float4* row = c + ((long)(iy + r) * (NX / 2) + px) * NZ + iz;
            if ((ix & 1) == 0) {
                const float4 a = row[0], b = row[1];
                v[4 * r + 0] = make_float2(a.x, a.y); v[4 * r + 1] = make_float2(b.x, b.y); v[4 * r + 2] = make_float2(a.z, a.w); v[4 * r + 3] = make_float2(b.z, b.w);
            } else {
                const float2* lo = reinterpret_cast<const float2*>(row) + 1;                 // odd element of this pair
                const float2* hi = reinterpret_cast<const float2*>(row + NZ);                // even element of the next pair
                v[4 * r + 0] = lo[0]; v[4 * r + 1] = lo[2]; v[4 * r + 2] = hi[0]; v[4 * r + 3] = hi[2];
            }
        }
        double w, h; finish(v, ty, tx, tz, w, h); ow[i] = w; oh[i] = h;
    }
}
// L2: float4 per (iy, ix in [0, NX-1), z) = (col ix, col ix+1)
__global__ __launch_bounds__(256) void gather_l2(const float4* c, const double* pts, long n, double* ow, double* oh) {
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        int iy, ix, iz; double ty, tx, tz; cell(pts + 3 * i, iy, ix, iz, ty, tx, tz);
        float2 v[8];
#pragma unroll
        for (int r = 0; r < 2; ++r) {
            const float4* row = c + ((long)(iy + r) * (NX - 1) + ix) * NZ + iz;
            const float4 a = row[0], b = row[1];
            v[4 * r + 0] = make_float2(a.x, a.y); v[4 * r + 1] = make_float2(b.x, b.y); v[4 * r + 2] = make_float2(a.z, a.w); v[4 * r + 3] = make_float2(b.z, b.w);
        }
        double w, h; finish(v, ty, tx, tz, w, h); ow[i] = w; oh[i] = h;
    }
}
__global__ __launch_bounds__(256) void build_l1(const float2* c, float4* o) {
    const long n = (long)NY * (NX / 2) * NZ;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const long iz = i % NZ, p = i / NZ, px = p % (NX / 2), iy = p / (NX / 2);
        const float2 a = c[((long)iy * NX + 2 * px) * NZ + iz], b = c[((long)iy * NX + 2 * px + 1) * NZ + iz];
        o[i] = make_float4(a.x, a.y, b.x, b.y);
    }
}
__global__ __launch_bounds__(256) void build_l2(const float2* c, float4* o) {
    const long n = (long)NY * (NX - 1) * NZ;
    for (long i = blockIdx.x * 256L + threadIdx.x; i < n; i += gridDim.x * 256L) {
        const long iz = i % NZ, p = i / NZ, ix = p % (NX - 1), iy = p / (NX - 1);
        const float2 a = c[((long)iy * NX + ix) * NZ + iz], b = c[((long)iy * NX + ix + 1) * NZ + iz];
        o[i] = make_float4(a.x, a.y, b.x, b.y);
    }
}

template <typename F> static float timeit(F f, int reps) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < reps; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / reps;
}

int main(int argc, char** argv) {
    const long n = argc > 1 ? atol(argv[1]) : 5000000;
    const size_t cells = (size_t)NY * NX * NZ;
    std::vector<float2> hc(cells);
    std::mt19937_64 g(1); std::uniform_real_distribution<float> u(0.f, 1.f);
    for (auto& v : hc) v = make_float2(u(g), 100.f + u(g));
    std::vector<double> hp(3 * n); std::uniform_real_distribution<double> ud(0.0, 1.0);
    for (long i = 0; i < n; ++i) { hp[3 * i] = 1.0 + ud(g) * (NY - 3); hp[3 * i + 1] = 1.0 + ud(g) * (NX - 3); hp[3 * i + 2] = ud(g) * (NZ - 1.001); }
    float2* c0; float4 *c1, *c2; double *pts, *ow, *oh, *rw, *rh;
    CK(hipMalloc(&c0, cells * 8)); CK(hipMalloc(&c1, cells * 8)); CK(hipMalloc(&c2, (size_t)NY * (NX - 1) * NZ * 16));
    CK(hipMalloc(&pts, n * 24)); CK(hipMalloc(&ow, n * 8)); CK(hipMalloc(&oh, n * 8)); CK(hipMalloc(&rw, n * 8)); CK(hipMalloc(&rh, n * 8));
    CK(hipMemcpy(c0, hc.data(), cells * 8, hipMemcpyHostToDevice)); CK(hipMemcpy(pts, hp.data(), n * 24, hipMemcpyHostToDevice));
    const int G = 256 * 8;
    const float b1 = timeit([&] { hipLaunchKernelGGL(build_l1, dim3(G * 4), dim3(256), 0, 0, c0, c1); }, 5);
    const float b2 = timeit([&] { hipLaunchKernelGGL(build_l2, dim3(G * 4), dim3(256), 0, 0, c0, c2); }, 5);
    const float t0 = timeit([&] { hipLaunchKernelGGL(gather_l0, dim3(G), dim3(256), 0, 0, c0, pts, n, rw, rh); }, 10);
    const float t1 = timeit([&] { hipLaunchKernelGGL(gather_l1, dim3(G), dim3(256), 0, 0, c1, pts, n, ow, oh); }, 10);
    std::vector<double> a(n), b(n);
    CK(hipMemcpy(a.data(), rw, n * 8, hipMemcpyDeviceToHost)); CK(hipMemcpy(b.data(), ow, n * 8, hipMemcpyDeviceToHost));
    long bad1 = 0; for (long i = 0; i < n; ++i) bad1 += a[i] != b[i];
    const float t2 = timeit([&] { hipLaunchKernelGGL(gather_l2, dim3(G), dim3(256), 0, 0, c2, pts, n, ow, oh); }, 10);
    CK(hipMemcpy(b.data(), ow, n * 8, hipMemcpyDeviceToHost));
    long bad2 = 0; for (long i = 0; i < n; ++i) bad2 += a[i] != b[i];
    printf("{\"points\": %ld, \"cube\": \"1000x1000x50 f32 pairs\", \"gather_L0_ms\": %.4f, \"gather_L1_even_pairs_ms\": %.4f, \"gather_L2_all_pairs_ms\": %.4f, "
           "\"build_L1_ms\": %.4f, \"build_L2_ms\": %.4f, \"L1_mismatches\": %ld, \"L2_mismatches\": %ld, "
           "\"one_shot_L1_ms\": %.4f, \"one_shot_L2_ms\": %.4f}\n", n, t0, t1, t2, b1, b2, bad1, bad2, b1 + t1, b2 + t2);
    return 0;
}
