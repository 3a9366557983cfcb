// Weather-model cube producer (and its ECMWF hybrid-level front end).
// Part of libraider_hip.so (single translation unit: included by raider_hip.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "raider_kernels.h"

using namespace rdr;

// ---- cube producer ---------------------------------------------------------------------------------------------------
// models/weatherModel.py:235-262 for one model-level column per wavefront (lanes = levels): _find_e (:332-353, find_svp
// :750-780), _uniform_in_z (:603-629: native interpolate_1d with NaN fill, results cast to f32), _checkForNans (:631-635 =
// interpolator.fillna3D :110-130: leading NaNs <- first valid value, interior runs linear in the index, trailing NaNs <- fill),
// refractivities (:355-361, f32 arithmetic), _adjust_grid (:371-387: extra bottom level at zmin) and _getZTD (:389-403).
// Output goes straight into the two device cubes the delay kernels read (interleaved (wet,hydro), (y,x,z)).
__device__ __forceinline__ float svp_pa(double t) {
    const double t1 = 273.15, t2 = 250.15;
    const double tref = t - t1;
    const double wgt = (t - t2) / (t1 - t2);
    const double svpw = 6.1121 * exp((17.502 * tref) / (240.97 + tref));
    const double svpi = 6.1121 * exp((22.587 * tref) / (273.86 + tref));
    double svp = svpi + (svpw - svpi) * (wgt * wgt);
    if (t > t1) svp = svpw;
    if (t < t2) svp = svpi;
    return (float)(svp * 100.0);
}

// interpolate_1d (interpolate.h:78-118) with fill NaN on an LDS-resident column
__device__ __forceinline__ double interp_col(const double* xs, const double* ys, int n, double x) {
    int left = 0, right = n;
    while (right != left) { const int mid = (left + right) / 2; if (x < xs[mid]) right = mid; else left = mid + 1; }
    if (right < 1 || right > n - 1) return qnan();
    const double x0 = xs[right - 1], x1 = xs[right], y0 = ys[right - 1], y1 = ys[right];
    double r;
    {
#pragma clang fp contract(off)
        const double slope = (y1 - y0) / (x1 - x0);
        r = y0 + slope * (x - x0);
    }
    return r;
}

// fillna3D on one column held in LDS (float col[n]); every lane fixes its own levels
__device__ __forceinline__ void fillna_col(float* col, int n, float fill, int lane) {
    int first = n, last = -1;
    for (int j = lane; j < n; j += 64) if (col[j] == col[j]) { first = min(first, j); last = max(last, j); }
    for (int off = 32; off > 0; off >>= 1) { first = min(first, __shfl_xor(first, off, 64)); last = max(last, __shfl_xor(last, off, 64)); }
    float fixed[8];                                   // nz <= 512 -> <= 8 levels per lane
    int cnt = 0;
    for (int j = lane; j < n; j += 64, ++cnt) {
        float v = col[j];
        if (!(v == v)) {
            if (last < 0 || j > last) v = fill;
            else if (j < first) v = col[first];
            else {                                    // interior run: np.interp on the index
                int i = j - 1; while (!(col[i] == col[i])) --i;
                int k = j + 1; while (!(col[k] == col[k])) ++k;
                {
#pragma clang fp contract(off)
                    const double a = (double)col[i], b = (double)col[k]; const double slope = (b - a) / (double)(k - i); v = (float)(slope * (double)(j - i) + a);
                }
            }
        }
        fixed[cnt] = v;
    }
    __builtin_amdgcn_wave_barrier();
    cnt = 0;
    for (int j = lane; j < n; j += 64, ++cnt) col[j] = fixed[cnt];
    __builtin_amdgcn_wave_barrier();
}

struct ProducerParams {
    const double* zs; const double* p; const double* t; const double* hum;   // [ncol, nlev]
    int64_t ncol; int nlev; int hum_type;                                     // 0 = q, 1 = rh
    const double* new_z; int nz; int pad;                                     // output levels (without the pad level)
    float k1, k2, k3; double zmin, R_v, R_d;
    float2* pw; double2* tot;                                                 // [ncol, nzo] interleaved (wet, hydro)
    float* t_out; float* p_out; float* e_out;                                 // optional [ncol, nzo]
};

__global__ __launch_bounds__(256) void producer_kernel(ProducerParams P) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const int wave = threadIdx.x >> 6, lane = threadIdx.x & 63;
    const int nzo = P.nz + P.pad;
    // per-wave LDS: zs,p,t,e at model levels (f64) | t,p,e,wet,hyd at output levels (f32) | output level heights (f64)
    double* wbase = reinterpret_cast<double*>(smem_raw) + (size_t)wave * (4 * P.nlev + 3 * nzo + (5 * nzo + 1) / 2 + 1);
    double* c_z = wbase; double* c_p = c_z + P.nlev; double* c_t = c_p + P.nlev; double* c_e = c_t + P.nlev;
    double* o_z = c_e + P.nlev;
    double* o_sw = o_z + nzo; double* o_sh = o_sw + nzo;       // zenith totals (suffix sums of the trapezoid terms)
    float* o_t = reinterpret_cast<float*>(o_sh + nzo); float* o_p = o_t + nzo; float* o_e = o_p + nzo; float* o_w = o_e + nzo; float* o_h = o_w + nzo;
    for (int j = lane; j < nzo; j += 64) o_z[j] = (P.pad && j == 0) ? P.zmin : P.new_z[j - P.pad];
    const int64_t wstride = (int64_t)gridDim.x * 4;
    for (int64_t col = (int64_t)blockIdx.x * 4 + wave; col < P.ncol; col += wstride) {
        __builtin_amdgcn_wave_barrier();
        for (int k = lane; k < P.nlev; k += 64) {
            const int64_t g = col * P.nlev + k;
            const double t = P.t[g], p = P.p[g], h = P.hum[g];
            const float svp = svp_pa(t);
            double e;
            {
#pragma clang fp contract(off)
                if (P.hum_type == 0) { const double w = h / (1.0 - h); e = w * P.R_v * (p - (double)svp) / P.R_d; }   // weatherModel.py:343-348
                else e = h / 100.0 * (double)svp;                                                                      // :350-353
            }
            c_z[k] = P.zs[g]; c_p[k] = p; c_t[k] = t; c_e[k] = e;
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < P.nz; j += 64) {
            const double x = P.new_z[j];
            o_t[j + P.pad] = (float)interp_col(c_z, c_t, P.nlev, x);
            o_p[j + P.pad] = (float)interp_col(c_z, c_p, P.nlev, x);
            o_e[j + P.pad] = (float)interp_col(c_z, c_e, P.nlev, x);
        }
        __builtin_amdgcn_wave_barrier();
        fillna_col(o_p + P.pad, P.nz, 0.0f, lane);
        fillna_col(o_t + P.pad, P.nz, 1e16f, lane);
        fillna_col(o_e + P.pad, P.nz, 0.0f, lane);
        for (int j = lane; j < P.nz; j += 64) {
            const float t = o_t[j + P.pad], p = o_p[j + P.pad], e = o_e[j + P.pad];
            float w, h;
            {
#pragma clang fp contract(off)
                const float a = (P.k2 * e) / t; const float b = (P.k3 * e) / (t * t); w = a + b;                      // weatherModel.py:355-357
                h = (P.k1 * p) / t;                                                                                    // :359-361
            }
            o_w[j + P.pad] = w; o_h[j + P.pad] = h;
        }
        __builtin_amdgcn_wave_barrier();
        if (P.pad && lane == 0) { o_t[0] = o_t[1]; o_p[0] = o_p[1]; o_e[0] = o_e[1]; o_w[0] = o_w[1]; o_h[0] = o_h[1]; }   // utilFcns.padLower
        __builtin_amdgcn_wave_barrier();
        // _getZTD: total[j] = 1e-6 * trapz(f[j:], zs[j:]), np.trapz = sum(d * (y[1:] + y[:-1]) / 2): the SUFFIX sums of the trapezoid
        // terms.  One term per lane and strip of 64 levels, a reverse inclusive scan inside the wave (six shuffle steps), the strips
        // from the top down with a running carry - O(n) instead of one O(n) loop per level (which was half of this kernel's
        // instructions).  (NumPy's own summation is pairwise, not sequential: either order agrees with it to a few ulp.)
        {
            double carry_w = 0.0, carry_h = 0.0;
            // (strips over the nzo OUTPUT levels, not the nzo-1 terms: with (nzo-1) % 64 == 0 - 65, 129, ... levels - the top level
            // would otherwise belong to no strip and its total stay unwritten; it has no term: the empty sum, 0)
            for (int s0 = ((nzo + 63) / 64 - 1) * 64; s0 >= 0; s0 -= 64) {
                const int k = s0 + lane;
                double tw = 0.0, th = 0.0;
                if (k < nzo - 1) {
#pragma clang fp contract(off)
                    const double d = o_z[k + 1] - o_z[k];
                    tw = d * (double)(o_w[k + 1] + o_w[k]) / 2.0;
                    th = d * (double)(o_h[k + 1] + o_h[k]) / 2.0;
                }
#pragma unroll
                for (int off = 1; off < 64; off <<= 1) {
                    const double uw = __shfl_down(tw, off, 64), uh = __shfl_down(th, off, 64);
                    if (lane + off < 64) { tw += uw; th += uh; }
                }
                tw += carry_w; th += carry_h;
                if (k < nzo) { o_sw[k] = tw; o_sh[k] = th; }
                carry_w = __shfl(tw, 0, 64); carry_h = __shfl(th, 0, 64);
            }
        }
        __builtin_amdgcn_wave_barrier();
        for (int j = lane; j < nzo; j += 64) {
            const int64_t g = col * nzo + j;
            float2 v; v.x = o_w[j]; v.y = o_h[j];
            P.pw[g] = v;
            if (P.t_out) { P.t_out[g] = o_t[j]; P.p_out[g] = o_p[j]; P.e_out[g] = o_e[j]; }
            double2 tt; tt.x = 1e-6 * o_sw[j]; tt.y = 1e-6 * o_sh[j];        // (level nzo-1: the empty sum, written as 0 by the scan)
            P.tot[g] = tt;
        }
    }
}

// ---- ECMWF hybrid model levels -> pressure and geometric height (front end of the cube producer) ----------------------------
// utilFcns.calcgeoh (:781-859): half-level pressures a + b sp, geopotential integrated upwards from the surface with the moist
// temperature, geopotential height; utilFcns.geo_to_ht (:378-410): geometric height with latitude-dependent gravity and Earth
// radius; models/ecmwf.py:92-110: (lev, y, x) top-first -> (y, x, lev) bottom-first.  One column per thread.
// FLOAT64 arithmetic on the float32 inputs.  The reference evaluates these formulas in float32 (NumPy-1 casting rules), where
// dlogP = log(P1) - log(P0) and alpha = 1 - P0/(P1-P0) dlogP lose 3-4 digits: its heights sit up to 2.4 m from the float64
// values and move by METRES with a last-bit change of logf - no other platform can reproduce that realisation of the round-off
// (the test suite's float32 NumPy restatement does, on x86: tests/test_ref_files.py), and a float32 evaluation here would only
// add a second, different one.  DESIGN.md 6.5.
__global__ __launch_bounds__(256) void ecmwf_levels_kernel(const float* __restrict__ z_surf, const float* __restrict__ lnsp,
                                                           const float* __restrict__ t, const float* __restrict__ q,
                                                           const float* __restrict__ lats, const double* __restrict__ a,
                                                           const double* __restrict__ b, int nlev, int64_t ny, int64_t nx, double R_d,
                                                           double* __restrict__ p_out, double* __restrict__ zs_out) {
    const int64_t ncol = ny * nx;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    const double g0 = 9.80665;
    for (int64_t i = (int64_t)blockIdx.x * blockDim.x + threadIdx.x; i < ncol; i += stride) {
        const double lat = (double)lats[i / nx] * DEG_TO_RAD;
        const double c2 = cos(2.0 * lat);
        const double g_ll = 9.80616 * (1.0 - 0.002637 * c2 + 0.0000059 * (c2 * c2));         // _get_g_ll
        const double cl = cos(lat), sl = sin(lat);
        const double re = sqrt(1.0 / ((cl * cl) / (6378137.0 * 6378137.0) + (sl * sl) / (6356752.0 * 6356752.0)));   // get_Re
        const double gre = g_ll / g0 * re;
        const double sp = exp((double)lnsp[i]), zs0 = (double)z_surf[i];
        double z_h = 0.0;
        for (int lev = nlev; lev >= 1; --lev) {
            const int64_t g = (int64_t)(lev - 1) * ncol + i;
            const double tl = (double)t[g] * (1.0 + 0.609133 * (double)q[g]);               // moist temperature
            const double ph = a[lev - 1] + b[lev - 1] * sp, ph1 = a[lev] + b[lev] * sp;
            double dlogp, alpha;
            if (lev == 1) { dlogp = log(ph1 / 0.1); alpha = 0.6931471805599453; }
            else { dlogp = log(ph1 / ph); alpha = 1.0 - (ph / (ph1 - ph)) * dlogp; }
            const double trd = tl * R_d;
            const double gh = (z_h + trd * alpha + zs0) / g0;
            z_h += trd * dlogp;
            const int64_t o = i * nlev + (nlev - lev);
            p_out[o] = ph;
            zs_out[o] = (gh * re) / (gre - gh);                                              // geo_to_ht
        }
    }
}
