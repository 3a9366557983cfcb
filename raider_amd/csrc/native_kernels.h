// Kernels behind the mirrors of the reference's two native extensions (interpolate, makePoints).
// Part of libraider_hip.so (single translation unit: included by raider_hip.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "raider_kernels.h"

using namespace rdr;

// ---- native extension kernels -------------------------------------------------------------------
// interpolate.h:23-38 bisect_left: first index with x < a[i]
__device__ __forceinline__ int upper_bound_idx(const double* a, int n, double x) {
    int left = 0, right = n;
    while (right != left) {
        const int mid = (left + right) / 2;
        if (x < a[mid]) right = mid; else left = mid + 1;
    }
    return right;
}

struct NdParams {
    int ndim;
    int64_t len[8];
    int64_t off[8];      // offset of axis d inside `axes`
    int64_t stride[8];   // C-order element strides of `values`
};

__global__ void interp_nd_kernel(NdParams P, const double* __restrict__ axes, const double* __restrict__ values,
                                 const double* __restrict__ q, int64_t n, int has_fill, double fill, double* __restrict__ out) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        int lo[8], hi[8];
        double d0[8], d1[8];
        double vol = 1.0;
        bool filled = false;
        for (int d = 0; d < P.ndim; ++d) {
            const double* g = axes + P.off[d];
            const int N = (int)P.len[d];
            const double x = q[i * P.ndim + d];
            int h = upper_bound_idx(g, N, x);
            if (has_fill) { if (h < 1 || h > N - 1) { filled = true; } }
            h = min(max(h, 1), N - 1);
            hi[d] = h; lo[d] = h - 1;
            const double x0 = g[h - 1], x1 = g[h];
            vol *= x1 - x0;
            d0[d] = x - x0; d1[d] = x1 - x;
        }
        if (filled) { out[i] = fill; continue; }
        double r;
        {
#pragma clang fp contract(off)
        if (P.ndim == 1) {
            const double* g = axes + P.off[0];
            const double x0 = g[lo[0]], x1 = g[hi[0]], y0 = values[lo[0]], y1 = values[hi[0]];
            const double slope = (y1 - y0) / (x1 - x0);                                   // interpolate.h:115-116
            r = y0 + slope * (q[i] - x0);
        } else if (P.ndim == 2) {
            const int64_t s0 = P.stride[0];
            const double z00 = values[lo[0] * s0 + lo[1]], z01 = values[lo[0] * s0 + hi[1]];
            const double z10 = values[hi[0] * s0 + lo[1]], z11 = values[hi[0] * s0 + hi[1]];
            r = (d1[0] * (z00 * d1[1] + z01 * d0[1]) + d0[0] * (z10 * d1[1] + z11 * d0[1])) / vol;   // interpolate.cpp:78-81
        } else if (P.ndim == 3) {
            const int64_t s0 = P.stride[0], s1 = P.stride[1];
            const double w000 = values[lo[0] * s0 + lo[1] * s1 + lo[2]], w001 = values[lo[0] * s0 + lo[1] * s1 + hi[2]];
            const double w010 = values[lo[0] * s0 + hi[1] * s1 + lo[2]], w011 = values[lo[0] * s0 + hi[1] * s1 + hi[2]];
            const double w100 = values[hi[0] * s0 + lo[1] * s1 + lo[2]], w101 = values[hi[0] * s0 + lo[1] * s1 + hi[2]];
            const double w110 = values[hi[0] * s0 + hi[1] * s1 + lo[2]], w111 = values[hi[0] * s0 + hi[1] * s1 + hi[2]];
            r = (d1[0] * (d1[1] * (d1[2] * w000 + d0[2] * w001) + d0[1] * (d1[2] * w010 + d0[2] * w011)) +
                 d0[0] * (d1[1] * (d1[2] * w100 + d0[2] * w101) + d0[1] * (d1[2] * w110 + d0[2] * w111))) / vol;   // interpolate.cpp:164-174
        } else {
            r = 0.0;
            for (int j = 0; j < (1 << P.ndim); ++j) {                                      // interpolate.cpp:236-252
                int64_t idx = 0;
                double term = 1.0;
                for (int d = 0; d < P.ndim; ++d) idx += (int64_t)(((j >> d) & 1) ? hi[d] : lo[d]) * P.stride[d];
                term = values[idx];
                for (int d = 0; d < P.ndim; ++d) term *= ((j >> d) & 1) ? d0[d] : d1[d];
                r += term;
            }
            r /= vol;
        }
        }
        out[i] = r;
    }
}

// interpolate_1d along the last axis of [ncol, m] (interpolate.h:78-118)
__global__ void along_axis_kernel(const double* __restrict__ xs, const double* __restrict__ ys, int64_t ncol, int64_t m,
                                  const double* __restrict__ q, int64_t mq, int has_fill, double fill, double* __restrict__ out) {
    const int64_t total = ncol * mq;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t col = t / mq;
        const double* g = xs + col * m;
        const double* v = ys + col * m;
        const double x = q[t];
        int h = upper_bound_idx(g, (int)m, x);
        if (has_fill && (h < 1 || h > m - 1)) { out[t] = fill; continue; }
        h = min(max(h, 1), (int)m - 1);
        const double x0 = g[h - 1], x1 = g[h], y0 = v[h - 1], y1 = v[h];
        {
#pragma clang fp contract(off)
            const double slope = (y1 - y0) / (x1 - x0); out[t] = y0 + slope * (x - x0);
        }
    }
}

// makePoints.pyx:35-40: ray[r,c,k] = SP[r,c] + basespace[k]*SLV[r,c], basespace = arange(0, max_len+step, step)
__global__ void make_points_kernel(const double* __restrict__ sp, const double* __restrict__ slv, int64_t nrays, int64_t npts,
                                   double step, double* __restrict__ out) {
    const int64_t total = nrays * 3 * npts;
    for (int64_t t = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; t < total; t += (int64_t)gridDim.x * blockDim.x) {
        const int64_t k = t % npts, rc = t / npts;
        {
#pragma clang fp contract(off)
            const double b = (double)k * step; const double p = b * slv[rc]; out[t] = sp[rc] + p;
        }
    }
}
