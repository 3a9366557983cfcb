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

// ---- the same solver, organised for the machine -------------------------------------------------------------------------
// orbit_hermite above spends its time in 48 double-precision divisions per evaluation that depend on the node times only.
// orbit_los_fast_kernel keeps the state vectors and, per 4-node segment, the 12 reciprocals 1 / (t_i - t_j) and the 4 sums
// sum_j 1 / (t_i - t_j) in LDS (built once per workgroup); an evaluation is then ~160 multiply-adds.  A lane reloads its 44
// segment constants only when its Newton iterate moves to another segment (iteration 0 starts mid-orbit, iteration 1 is already
// in the final segment), and finds the segment by arithmetic on (nearly) uniform state-vector times.  Same iteration, same
// threshold / iteration cap / failure rules as orbit_los_kernel; results agree to rounding (1e-12 s, 1e-9 m).
constexpr int ORBIT_LDS_MAX_SV = 320;                // state vectors the LDS tables hold (an S1 orbit cut to +-600 s: 121)

struct OrbitSegRegs { double tt[4], inv[4][3], sum[4], x[4][3], v[4][3]; };

__device__ __forceinline__ void orbit_load_seg(const double* s_t, const double* s_p, const double* s_v, const double* s_seg, int i0, OrbitSegRegs& r) {
    const double* g = s_seg + 16 * i0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        r.tt[i] = s_t[i0 + i];
#pragma unroll
        for (int m = 0; m < 3; ++m) { r.inv[i][m] = g[3 * i + m]; r.x[i][m] = s_p[3 * (i0 + i) + m]; r.v[i][m] = s_v[3 * (i0 + i) + m]; }
        r.sum[i] = g[12 + i];
    }
}

__device__ __forceinline__ void orbit_eval_seg(const OrbitSegRegs& r, double t, double* pos, double* vel) {
    double d[4];
#pragma unroll
    for (int i = 0; i < 4; ++i) d[i] = t - r.tt[i];
    pos[0] = pos[1] = pos[2] = 0.0; vel[0] = vel[1] = vel[2] = 0.0;
#pragma unroll
    for (int i = 0; i < 4; ++i) {
        // the three other nodes in ascending order, as the reciprocals were stored
        const int o0 = i == 0 ? 1 : 0, o1 = i <= 1 ? 2 : 1, o2 = i <= 2 ? 3 : 2;
        const double a0 = d[o0] * r.inv[i][0], a1 = d[o1] * r.inv[i][1], a2 = d[o2] * r.inv[i][2];
        const double h = a0 * a1 * a2;
        const double hdot = fma(r.inv[i][0], a1 * a2, fma(r.inv[i][1], a0 * a2, r.inv[i][2] * (a0 * a1)));
        const double f0 = fma(-2.0 * d[i], r.sum[i], 1.0), f1 = d[i];
        const double g1 = fma(2.0 * d[i], hdot, h);
        const double g0 = 2.0 * fma(f0, hdot, -h * r.sum[i]);
        const double h2 = h * h;
#pragma unroll
        for (int k = 0; k < 3; ++k) {
            pos[k] = fma(fma(r.x[i][k], f0, r.v[i][k] * f1), h2, pos[k]);
            vel[k] = fma(fma(r.x[i][k], g0, r.v[i][k] * g1), h, vel[k]);
        }
    }
}

__global__ __launch_bounds__(256) void orbit_los_fast_kernel(const double* __restrict__ st, const double* __restrict__ sp, const double* __restrict__ sv, int nsv,
                                                             const double* __restrict__ xyz, int64_t n, double threshold, int maxiter,
                                                             double* __restrict__ los, double* __restrict__ aztime, double* __restrict__ srange) {
    extern __shared__ __attribute__((aligned(16))) unsigned char orbit_smem[];
    double* s_t = reinterpret_cast<double*>(orbit_smem);
    double* s_p = s_t + nsv;
    double* s_v = s_p + 3 * nsv;
    double* s_seg = s_v + 3 * nsv;                    // [nsv - 3][16]: inv[4][3] then sum[4]
    for (int i = threadIdx.x; i < nsv; i += blockDim.x) s_t[i] = st[i];
    for (int i = threadIdx.x; i < 3 * nsv; i += blockDim.x) { s_p[i] = sp[i]; s_v[i] = sv[i]; }
    __syncthreads();
    for (int g = threadIdx.x; g < nsv - 3; g += blockDim.x) {
        double* o = s_seg + 16 * g;
        for (int i = 0; i < 4; ++i) {
            double sum = 0.0; int m = 0;
            for (int j = 0; j < 4; ++j) {
                if (j == i) continue;
                const double iv = 1.0 / (s_t[g + i] - s_t[g + j]);
                o[3 * i + m++] = iv; sum += iv;
            }
            o[12 + i] = sum;
        }
    }
    __syncthreads();
    const double t_first = s_t[0], t_last = s_t[nsv - 1];
    const double inv_dt = (double)(nsv - 1) / (t_last - t_first);
    auto segment = [&](double t) {                    // i0 of orbit_hermite: first index with t < st[idx], minus 2, clamped
        int lo = (int)fmin(fmax((t - t_first) * inv_dt, 0.0), (double)(nsv - 1)) + 1;   // exact for uniform times, a guess otherwise
        while (lo > 0 && t < s_t[lo - 1]) --lo;
        while (lo < nsv && !(t < s_t[lo])) ++lo;
        return min(max(lo - 2, 0), nsv - 4);
    };
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double tx = xyz[3 * i], ty = xyz[3 * i + 1], tz = xyz[3 * i + 2];
        double t = 0.5 * (t_first + t_last);              // start at the orbit mid time
        double pos[3], vel[3];
        OrbitSegRegs r;
        int cur = -1;
        bool ok = false;
        for (int it = 0; it < maxiter; ++it) {
            const int i0 = segment(t);
            if (i0 != cur) { orbit_load_seg(s_t, s_p, s_v, s_seg, i0, r); cur = i0; }
            orbit_eval_seg(r, t, pos, vel);
            const double dx = tx - pos[0], dy = ty - pos[1], dz = tz - pos[2];
            const double fn = dx * vel[0] + dy * vel[1] + dz * vel[2];            // zero-Doppler condition
            const double fnp = -(vel[0] * vel[0] + vel[1] * vel[1] + vel[2] * vel[2]);
            const double step = fn / fnp;
            t -= step;
            if (fabs(step) < threshold) { ok = true; break; }
            if (!(t == t)) break;                         // NaN target: no point in iterating on
        }
        double l0 = qnan(), l1 = qnan(), l2 = qnan(), rg = qnan();
        if (ok && t >= t_first && t <= t_last && tx == tx && ty == ty && tz == tz) {
            const int i0 = segment(t);
            if (i0 != cur) { orbit_load_seg(s_t, s_p, s_v, s_seg, i0, r); cur = i0; }
            orbit_eval_seg(r, t, pos, vel);
            const double dx = pos[0] - tx, dy = pos[1] - ty, dz = pos[2] - tz;
            rg = sqrt(dx * dx + dy * dy + dz * dz);
            l0 = dx / rg; l1 = dy / rg; l2 = dz / rg;                               // losreader.py:251-252
        } else t = qnan();
        los[3 * i] = l0; los[3 * i + 1] = l1; los[3 * i + 2] = l2;
        if (aztime) aztime[i] = t;
        if (srange) srange[i] = rg;
    }
}

