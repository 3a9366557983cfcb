// Zero-Doppler look vectors from orbit state vectors.
// Part of libraider_hip.so (single translation unit: included by raider_hip.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "raider_kernels.h"

using namespace rdr;

// ---- look vectors from orbit state vectors ----------------------------------------------------------------------------
// Replaces the per-pixel Python loop over isce3.geometry.geo2rdr + Orbit.interpolate of Raytracing.getLookVectors
// (losreader.py:219-255).  isce3 is a third-party dependency that is not under /root/reference: this restates the published
// algorithm as the call site uses it (empty Doppler LUT => zero-Doppler): Newton on azimuth time t for
// f(t) = (T - S(t)) . V(t) = 0 with f'(t) ~ -|V|^2, S/V from 4-point Hermite interpolation of the state vectors,
// threshold 1e-7 s, <= 30 iterations; los = (S(t) - T)/|S(t) - T|; failures -> NaN.  PARITY WITH isce3 IS UNPINNED.
__device__ inline void orbit_hermite(const double* __restrict__ st, const double* __restrict__ sp, const double* __restrict__ sv,
                                     int n, double t, double* pos, double* vel) {
    // 4 state vectors bracketing t (two on each side where possible)
    int lo = 0, hi = n;                      // first index with t < st[idx]
    while (lo < hi) { const int mid = (lo + hi) >> 1; if (t < st[mid]) hi = mid; else lo = mid + 1; }
    int i0 = min(max(lo - 2, 0), n - 4);
    double tt[4], h[4], hdot[4], f0[4], f1[4], g0[4], g1[4];
    for (int i = 0; i < 4; ++i) tt[i] = st[i0 + i];
    for (int i = 0; i < 4; ++i) {
        f1[i] = t - tt[i];
        double sum = 0.0;
        for (int j = 0; j < 4; ++j) if (j != i) sum += 1.0 / (tt[i] - tt[j]);
        f0[i] = 1.0 - 2.0 * (t - tt[i]) * sum;
        double prod = 1.0;
        for (int k = 0; k < 4; ++k) if (k != i) prod *= (t - tt[k]) / (tt[i] - tt[k]);
        h[i] = prod;
        double s2 = 0.0;
        for (int j = 0; j < 4; ++j) {
            if (j == i) continue;
            double p2 = 1.0;
            for (int k = 0; k < 4; ++k) if (k != i && k != j) p2 *= (t - tt[k]) / (tt[i] - tt[k]);
            s2 += p2 / (tt[i] - tt[j]);
        }
        hdot[i] = s2;
        g1[i] = h[i] + 2.0 * (t - tt[i]) * hdot[i];
        g0[i] = 2.0 * (f0[i] * hdot[i] - h[i] * sum);
    }
    for (int k = 0; k < 3; ++k) {
        double sx = 0.0, sv_ = 0.0;
        for (int i = 0; i < 4; ++i) {
            const double x = sp[3 * (i0 + i) + k], v = sv[3 * (i0 + i) + k];
            sx += (x * f0[i] + v * f1[i]) * h[i] * h[i];
            sv_ += (x * g0[i] + v * g1[i]) * h[i];
        }
        pos[k] = sx; vel[k] = sv_;
    }
}

__global__ void orbit_los_kernel(const double* __restrict__ st, const double* __restrict__ sp, const double* __restrict__ sv, int nsv,
                                 const double* __restrict__ xyz, int64_t n, double threshold, int maxiter,
                                 double* __restrict__ los, double* __restrict__ aztime, double* __restrict__ srange) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double tx = xyz[3 * i], ty = xyz[3 * i + 1], tz = xyz[3 * i + 2];
        double t = 0.5 * (st[0] + st[nsv - 1]);          // start at the orbit mid time
        double pos[3], vel[3];
        bool ok = false;
        for (int it = 0; it < maxiter; ++it) {
            orbit_hermite(st, sp, sv, nsv, t, pos, vel);
            const double dx = tx - pos[0], dy = ty - pos[1], dz = tz - pos[2];
            const double fn = dx * vel[0] + dy * vel[1] + dz * vel[2];            // zero-Doppler condition
            const double fnp = -(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
            const double step = fn / fnp;
            t -= step;
            if (fabs(step) < threshold) { ok = true; break; }
        }
        double l0 = qnan(), l1 = qnan(), l2 = qnan(), rg = qnan();
        if (ok && t >= st[0] && t <= st[nsv - 1] && tx == tx && ty == ty && tz == tz) {
            orbit_hermite(st, sp, sv, nsv, t, pos, vel);
            const double dx = pos[0] - tx, dy = pos[1] - ty, dz = pos[2] - tz;
            rg = sqrt(dx * dx + dy * dy + dz * dz);
            l0 = dx / rg; l1 = dy / rg; l2 = dz / rg;                               // losreader.py:251-252
        } else t = qnan();
        los[3 * i] = l0; los[3 * i + 1] = l1; los[3 * i + 2] = l2;
        if (aztime) aztime[i] = t;
        if (srange) srange[i] = rg;
    }
}
