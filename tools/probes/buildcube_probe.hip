// Upper-bound probe for an LDS-staged build_cube_kernel (config 2: 1000 x 1000 nodes x 40 heights, f64 cube, 640 MB out): every point
// reads its eight 16 B corners from LDS (a staged 2 x 12-column x 8-level footprint per 64 x 4-node tile and 4-height batch; the
// staging loads are faked by a few global loads per thread), does the 16 multiply-adds and writes two doubles non-temporally.
// Compared with the same loop reading the corners from global memory (L2 / L1 hits), i.e. the present kernel's inner loop.
//   hipcc --offload-arch=gfx950 -O3 buildcube_probe.hip -o /tmp/bp && /tmp/bp
#include <hip/hip_runtime.h>
#include <cstdio>
typedef double D2 __attribute__((ext_vector_type(2)));
constexpr int NCX = 12, NCY = 3, NCOL = NCX * NCY;
template <bool LDS, int U, bool CZLDS>
__global__ __launch_bounds__(256) void k(const D2* __restrict__ cube, int cnx, int cnz, double* __restrict__ w, double* __restrict__ h, long nodes, int nx, int nz) {
    __shared__ D2 s[2 * U][NCOL];
    __shared__ int s_cz[64]; __shared__ double s_tz[64];
    if (threadIdx.x < 64) { s_cz[threadIdx.x] = threadIdx.x < nz ? threadIdx.x : 0; s_tz[threadIdx.x] = 0.25; }
    __syncthreads();
    const int tiles_x = (nx + 63) / 64;
    for (long t = blockIdx.x; t < (long)tiles_x * ((nodes / nx + 3) / 4); t += gridDim.x) {
        const int tx = (int)(t % tiles_x), ty = (int)(t / tiles_x);
        const int ix = tx * 64 + (threadIdx.x & 63), iy = ty * 4 + (threadIdx.x >> 6);
        const bool act = ix < nx && (long)iy * nx < nodes;
        const long i = (long)iy * nx + ix;
        // node -> cube cell (query spacing 0.15 of a cell), weights
        const double fx = ix * 0.15, fy = iy * 0.15;
        const int cx = (int)fx, cy = (int)fy;
        const double tx_ = fx - cx, ty_ = fy - cy;
        const double a00 = (1 - ty_) * (1 - tx_), a01 = (1 - ty_) * tx_, a10 = ty_ * (1 - tx_), a11 = ty_ * tx_;
        const int cx0 = (int)(tx * 64 * 0.15), cy0 = (int)(ty * 4 * 0.15);
        const int lc = (cy - cy0) * NCX + (cx - cx0);                 // local column of (y0, x0)
        const D2* c00 = cube + ((long)cy * cnx + cx) * cnz;
        for (int z0 = 0; z0 < nz; z0 += U) {
            if (LDS) {
                __syncthreads();
                for (int e = threadIdx.x; e < 2 * U * NCOL; e += 256) {
                    const int slot = e / NCOL, col = e % NCOL;
                    s[slot][col] = cube[((long)(cy0 + col / NCX) * cnx + cx0 + col % NCX) * cnz + z0 + (slot >> 1) + (slot & 1)];
                }
                __syncthreads();
            }
#pragma unroll
            for (int u = 0; u < U; ++u) {
                D2 v[8];
                if (LDS) {
                    v[0] = s[2 * u][lc]; v[1] = s[2 * u + 1][lc]; v[2] = s[2 * u][lc + 1]; v[3] = s[2 * u + 1][lc + 1];
                    v[4] = s[2 * u][lc + NCX]; v[5] = s[2 * u + 1][lc + NCX]; v[6] = s[2 * u][lc + NCX + 1]; v[7] = s[2 * u + 1][lc + NCX + 1];
                } else {
                    const int cz = CZLDS ? s_cz[z0 + u] : z0 + u;
                    v[0] = c00[cz]; v[1] = c00[cz + 1]; v[2] = c00[cnz + cz]; v[3] = c00[cnz + cz + 1];
                    v[4] = c00[(long)cnx * cnz + cz]; v[5] = c00[(long)cnx * cnz + cz + 1]; v[6] = c00[(long)cnx * cnz + cnz + cz]; v[7] = c00[(long)cnx * cnz + cnz + cz + 1];
                }
                const double tz = CZLDS ? s_tz[z0 + u] : 0.25, wz0 = 1.0 - tz;
                double sw = 0, sh = 0;
                sw += v[0].x * (a00 * wz0); sh += v[0].y * (a00 * wz0); sw += v[1].x * (a00 * tz); sh += v[1].y * (a00 * tz);
                sw += v[2].x * (a01 * wz0); sh += v[2].y * (a01 * wz0); sw += v[3].x * (a01 * tz); sh += v[3].y * (a01 * tz);
                sw += v[4].x * (a10 * wz0); sh += v[4].y * (a10 * wz0); sw += v[5].x * (a10 * tz); sh += v[5].y * (a10 * tz);
                sw += v[6].x * (a11 * wz0); sh += v[6].y * (a11 * wz0); sw += v[7].x * (a11 * tz); sh += v[7].y * (a11 * tz);
                if (act) { const long o = (long)(z0 + u) * nodes + i; __builtin_nontemporal_store(sw, w + o); __builtin_nontemporal_store(sh, h + o); }
            }
        }
    }
}
template <typename F> float timeit(F f) {
    hipEvent_t a, b; hipEventCreate(&a); hipEventCreate(&b);
    f(); f(); hipDeviceSynchronize();
    hipEventRecord(a); for (int i = 0; i < 10; ++i) f(); hipEventRecord(b); hipEventSynchronize(b);
    float ms; hipEventElapsedTime(&ms, a, b); return ms / 10;
}
int main() {
    const int nx = 1000, ny = 1000, nz = 40, cnx = 300, cny = 300, cnz = 80;
    const long nodes = (long)nx * ny;
    double *w, *h; D2* cube;
    hipMalloc(&w, nodes * nz * 8); hipMalloc(&h, nodes * nz * 8); hipMalloc(&cube, (size_t)cnx * cny * cnz * 16); hipMemset(cube, 0, (size_t)cnx * cny * cnz * 16);
    hipMemset(cube, 0x3c, (size_t)cnx * cny * cnz * 16);   // non-zero data
    for (int g : {2048, 4000, 8192}) {
        float a = timeit([&] { hipLaunchKernelGGL((k<false, 4, false>), dim3(g), dim3(256), 0, 0, cube, cnx, cnz, w, h, nodes, nx, nz); });
        float a2 = timeit([&] { hipLaunchKernelGGL((k<false, 2, false>), dim3(g), dim3(256), 0, 0, cube, cnx, cnz, w, h, nodes, nx, nz); });
        float a1 = timeit([&] { hipLaunchKernelGGL((k<false, 1, false>), dim3(g), dim3(256), 0, 0, cube, cnx, cnz, w, h, nodes, nx, nz); });
        float a3 = timeit([&] { hipLaunchKernelGGL((k<false, 2, true>), dim3(g), dim3(256), 0, 0, cube, cnx, cnz, w, h, nodes, nx, nz); });
        float b = timeit([&] { hipLaunchKernelGGL((k<true, 4, false>), dim3(g), dim3(256), 0, 0, cube, cnx, cnz, w, h, nodes, nx, nz); });
        printf("grid %5d: global U=4 %7.1f  U=2 %7.1f  U=1 %7.1f  U=2+cz from LDS %7.1f   LDS-staged U=4 %7.1f us\n", g, a * 1e3, a2 * 1e3, a1 * 1e3, a3 * 1e3, b * 1e3);
    }
    return 0;
}
