// Device-side building blocks of libraider_hip (gfx950 / CDNA4 only).
//
//   CubeView + trilinear()     scipy RegularGridInterpolator(linear, fill=nan) on the interleaved (wet,hydro) cube
//                              [delayFcns.py:55-56, scipy _rgi.py:405-499]               (zenith / station kernels)
//   sample_issue / _finish     the same interpolation for the ray kernels, split so that several samples' gathers fly together
//   toa_newton / _t            getTopOfAtmosphere [losreader.py:706-733] (materialising API kernels / generic ray kernels)
//   crossings_kernel           pass 1: build_ray [losreader.py:772-835] -> per-level batch maximum of the ray length
//                              (delay.py:283) + NaN / z-clamp flags (delay.py:279,306-311) + the ray records for pass 2
//   march_kernel               pass 2: the trapezoid of _build_cube_ray [delay.py:285-323]
//
// Design notes (MI355X):
//   * one ray per lane, 64-lane wavefronts, 256-thread workgroups = one 16x16 pixel tile of the scene: neighbouring rays
//     walk the same few cube columns, so their gathers hit the same L1/L2 lines; everything that depends only on the slice
//     (levels, nParts, the sample schedule) is wave-uniform and lives in SGPRs / LDS, so a wave never diverges.
//   * light rays (almost all): height, latitude and longitude along the ray are degree-5 polynomials of the ray parameter
//     fitted once per ray from six full geodesy evaluations (geodesy_fast.h); level crossings and samples are then FMAs.
//     The few rays the static classification rejects (poles, grazing incidence, +-180 deg) take the generic kernels.
//   * the cube is stored (y,x,z) with z fastest and (wet,hydro) interleaved per cell: the two z neighbours of both fields
//     of one column are ONE contiguous 16 B (f32) / 32 B (f64) read, four per sample against uniform row bases.
//   * grid axes + the level table + the per-level partition live in LDS (a few KB), filled once per workgroup; nothing
//     per-level is ever materialised in HBM (the reference materialises K x N x 56 B).
//   * persistent grid (a few workgroups per CU) walking tiles with an XCD-aware mapping so that the 32 CUs sharing one L2
//     work on one contiguous band of the scene.
//   * no MFMA: this is a gather + fp64 polynomial/interpolation integrate, there is no contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>
#include <type_traits>

#include "geodesy.h"
#include "geodesy_fast.h"

namespace rdr {

constexpr int TILE = 16;         // 16x16 pixel tile per workgroup (GRID mode)
constexpr int BLOCK = TILE * TILE;
constexpr int MAX_LEVELS = 512;  // model intervals (ERA5 has 144)
constexpr int MAX_NPARTS = 65536; // integration points per model interval (guards the sample loop against diverged ray lengths)

template <typename T2>
struct CubeView {
    const T2* v;          // [(iy*nx+ix)*nz+iz] -> (wet, hydro)
    const double* axes;   // ys[ny] | xs[nx] | zs[nz]   (ascending)
    int ny, nx, nz;
    double y_lo, y_hi, x_lo, x_hi, z_lo, z_hi;   // axis end points (bounds test)
    double inv_dy, inv_dx, inv_dz;               // (n-1)/(g[n-1]-g[0]) for the uniform-axis index guess
    int uni_y, uni_x, uni_z;                     // axis is (nearly) uniform: guess is within +-1 cell
    int exact_y, exact_x;                        // axis is uniform to round-off: the cell follows from arithmetic alone
    int small;                                   // ny*nx < 2^24, nz < 2^24 and the field fits 4 GB: 32-bit index arithmetic
};

__device__ __forceinline__ double qnan() { return __longlong_as_double(0x7ff8000000000000LL); }

// index i with g[i] <= x < g[i+1], clamped to [0, n-2] (last cell closed) - scipy find_indices.
__device__ __forceinline__ int find_cell(const double* g, int n, double x, double g0, double inv_d, int uniform) {
    int i;
    if (uniform) {
        i = (int)((x - g0) * inv_d);
        i = min(max(i, 0), n - 2);
        while (i > 0 && x < g[i]) --i;
        while (i < n - 2 && x >= g[i + 1]) ++i;
    } else {
        int lo = 0, hi = n;   // first index with x < g[idx]
        while (lo < hi) {
            const int mid = (lo + hi) >> 1;
            if (x < g[mid]) hi = mid; else lo = mid + 1;
        }
        i = min(max(lo - 1, 0), n - 2);
    }
    return i;
}

__device__ __forceinline__ void ld2(const float2* p, double& a, double& b) { const float2 t = *p; a = (double)t.x; b = (double)t.y; }
__device__ __forceinline__ void ld2(const double2* p, double& a, double& b) { const double2 t = *p; a = t.x; b = t.y; }

// scipy linear RGI on both fields (zenith / station kernels).  sy/sx/sz: the axes (LDS or global).
template <typename T2>
__device__ __forceinline__ void trilinear(const CubeView<T2>& c, const double* sy, const double* sx, const double* sz,
                                          double y, double x, double z, double& wet, double& hyd) {
    // out of bounds (x < g[0] or x > g[-1]) -> fill_value nan; nan coordinate -> nan   (_rgi.py:437-442,585-592)
    const bool inside = (y >= c.y_lo) && (y <= c.y_hi) && (x >= c.x_lo) && (x <= c.x_hi) && (z >= c.z_lo) && (z <= c.z_hi);
    if (!inside) { wet = qnan(); hyd = qnan(); return; }
    const int iy = find_cell(sy, c.ny, y, c.y_lo, c.inv_dy, c.uni_y);
    const int ix = find_cell(sx, c.nx, x, c.x_lo, c.inv_dx, c.uni_x);
    const int iz = find_cell(sz, c.nz, z, c.z_lo, c.inv_dz, c.uni_z);
    const double ty = (y - sy[iy]) / (sy[iy + 1] - sy[iy]);
    const double tx = (x - sx[ix]) / (sx[ix + 1] - sx[ix]);
    const double tz = (z - sz[iz]) / (sz[iz + 1] - sz[iz]);
    const T2* p00 = c.v + ((int64_t)iy * c.nx + ix) * c.nz + iz;   // (y0,x0)
    const T2* p01 = p00 + c.nz;                                    // (y0,x1)
    const T2* p10 = p00 + (int64_t)c.nx * c.nz;                    // (y1,x0)
    const T2* p11 = p10 + c.nz;                                    // (y1,x1)
    double w[8], h[8];
    ld2(p00, w[0], h[0]); ld2(p00 + 1, w[1], h[1]);
    ld2(p01, w[2], h[2]); ld2(p01 + 1, w[3], h[3]);
    ld2(p10, w[4], h[4]); ld2(p10 + 1, w[5], h[5]);
    ld2(p11, w[6], h[6]); ld2(p11 + 1, w[7], h[7]);
    const double wy0 = 1.0 - ty, wx0 = 1.0 - tx, wz0 = 1.0 - tz;
    // weight = ((1*wy)*wx)*wz, corners in lexicographic (y,x,z) order, value = 0 + sum   (_rgi.py:490-498)
    const double a00 = wy0 * wx0, a01 = wy0 * tx, a10 = ty * wx0, a11 = ty * tx;
    const double k0 = a00 * wz0, k1 = a00 * tz, k2 = a01 * wz0, k3 = a01 * tz;
    const double k4 = a10 * wz0, k5 = a10 * tz, k6 = a11 * wz0, k7 = a11 * tz;
    double sw = 0.0, sh = 0.0;
    sw += w[0] * k0; sh += h[0] * k0;
    sw += w[1] * k1; sh += h[1] * k1;
    sw += w[2] * k2; sh += h[2] * k2;
    sw += w[3] * k3; sh += h[3] * k3;
    sw += w[4] * k4; sh += h[4] * k4;
    sw += w[5] * k5; sh += h[5] * k5;
    sw += w[6] * k6; sh += h[6] * k6;
    sw += w[7] * k7; sh += h[7] * k7;
    wet = sw; hyd = sh;
}

// LDS-resident axis tables of the ray kernels: (g[i], 1/(g[i+1]-g[i])) pairs per axis; ey / ex are null for exact axes
struct AxisTabs { const double2* ey; const double2* ex; const double2* ez; };

// ---- ray-kernel sampler ----------------------------------------------------------------------------------------
// Same scipy semantics as trilinear<> (interval g[i] <= v < g[i+1], last cell closed; outside / NaN -> NaN).
// The axis table is stored in LDS as (g[i], 1/(g[i+1]-g[i])) pairs.  x / y: arithmetic cell on exactly-uniform axes,
// guess-and-verify on nearly uniform ones; z: the segment's model interval is known, a 2- (light kernel) or 3-entry
// (generic kernel) window around it is fetched and the right entry selected in registers.  A lane whose guess fails
// (irregular axis, non-converged crossing in sub-metre levels, < 4-node axes, outside the grid) searches exactly.
__device__ __forceinline__ int bisect_cell2(const double2* e, int n, double v) {
    int lo = 0, hi = n;   // first index with v < g[idx]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v < e[mid].x) hi = mid; else lo = mid + 1;
    }
    return min(max(lo - 1, 0), n - 2);
}

// Exact cell search (scipy find_indices semantics) - the rare path of the two fast searches below.  A coordinate outside
// the axis (or NaN) gets a NaN weight, which makes the interpolated values NaN = scipy's fill_value (_rgi.py:437-442,585-592).
__device__ __forceinline__ void cell_exact(const double2* e, int n, double v, int& i, double& t) {
    if (!((v >= e[0].x) & (v <= e[n - 1].x))) { i = 0; t = qnan(); return; }
    i = bisect_cell2(e, n, v);
    t = (v - e[i].x) * e[i].y;
}

__device__ __forceinline__ void window_cell(const double2* e, int n, double v, int guess, bool trust, int& i, double& t) {
    const int i0 = min(max(guess, 1), n - 3);
    const double2 em = e[i0 - 1], e0 = e[i0], e1 = e[i0 + 1];
    const bool dn = v < e0.x, up = v >= e1.x;
    i = i0 - (int)dn + (int)up;
    const double g = dn ? em.x : (up ? e1.x : e0.x);
    const double r = dn ? em.y : (up ? e1.y : e0.y);
    t = (v - g) * r;
    if (!trust || !(t >= 0.0) || !(t <= 1.0)) cell_exact(e, n, v, i, t);      // rare
}

// The sample is known to sit in model interval `guess` or - within the Newton residual of a level crossing - just above its
// top node (every evaluated sample but a ray's very first one is interior to its segment or its TOP end): two table
// entries decide.  Anything else (t outside [0,1]) falls back to the exact search.
__device__ __forceinline__ int window2_base(int n, int guess) { return min(max(guess, 0), n - 3); }

__device__ __forceinline__ void window2_cell(const double2* e, int n, double v, int i0, bool trust, int& i, double& t) {
    const double2 e0 = e[i0], e1 = e[i0 + 1];                                 // i0 = window2_base(n, guess), slice-uniform
    // measured from the MIDDLE node g[i0+1] (where a level top sits, within the Newton residual): above it the weight is d r1,
    // below it 1 + d r0 - one subtraction, one 64-bit select (the reciprocal), one 32-bit select (the high word of 0.0 / 1.0)
    const double d = v - e1.x;
    const bool up = d >= 0.0;
    i = i0 + (int)up;
    t = fma(d, up ? e1.y : e0.y, up ? 0.0 : 1.0);
    if (!trust || !(t >= 0.0) || !(t <= 1.0)) cell_exact(e, n, v, i, t);      // rare
}

// x / y axes.  `exact` (axis uniform to round-off, the usual lat/lon grid): the cell index and the weight come from
// (v - g0) * (n-1)/(g[n-1]-g0) alone - no table access; they differ from scipy's (v - g[i])/(g[i+1]-g[i]) by the axis's own
// round-off (<= 1e-11 of a cell, checked on the host), which the continuous interpolant turns into <= 1e-12 relative.
// Otherwise (nearly uniform axis): read the guessed cell and the next node from the LDS table and verify.
// Anything else (last node, outside, NaN, irregular axis) takes the exact search.
// IDX (light rays on an exact axis): v already IS the index-space coordinate (v_real - g0) * inv_d - pass 1 folds that map
// into the ray polynomial's coefficients.
// NOCHECK (light march loop, exact axes only): the caller has PROVEN from the ray's polynomial coefficients that every sample of
// every ray of the wave lies inside [0, n-1) - the bounds test and its rare path are compiled out.
template <bool IDX = false, bool NOCHECK = false>
__device__ __forceinline__ void cell_xy(const double2* e, int n, double v, double g0, double g_last, double inv_d, bool exact, bool trust,
                                        int& i, double& t) {
    bool ok;
    const double tf = (IDX && exact) ? v : (v - g0) * inv_d;
    if (exact) {
        i = (int)tf;                                                            // = floor for the tf >= 0 this path accepts
        t = __builtin_amdgcn_fract(tf);
        if (NOCHECK) return;
        const double nm1 = (double)(n - 1);
        ok = (tf >= 0.0) & (tf < nm1);
        if (__builtin_expect(!ok, 0)) {                                         // rare: the last node, outside the axis, NaN
            asm volatile("" ::: "memory");                                      // keep this a skipped branch, not if-converted selects
            // inside the axis after all?  In axis units when v is one (scipy's own test, _rgi.py:437-442); an index-space
            // coordinate (IDX) is inside only ON the last node.  The cell follows from which end tf fell off: a coordinate a
            // rounding error below the first node belongs to cell 0, not to the last cell.
            const bool inside = IDX ? (tf == nm1) : ((v >= g0) & (v <= g_last));
            const bool low_end = tf < 1.0;
            i = (inside && !low_end) ? n - 2 : 0;
            t = inside ? (low_end ? tf : (tf - nm1) + 1.0) : qnan();
        }
        return;
    } else {
        // one table entry: the weight itself says whether the guess was the cell (t in [0, 1)); anything else - the neighbour
        // cell, the last node, outside, NaN - takes the exact search.  (A v exactly ON node i+1 may be accepted here with
        // t = 1 - 1 ulp instead of cell i+1 with t = 0: the same value of the continuous interpolant.)
        i = min(max((int)tf, 0), n - 2);
        const double2 e0 = e[i];
        t = (v - e0.x) * e0.y;
        ok = trust & (t >= 0.0) & (t < 1.0);
    }
    if (!ok) cell_exact(e, n, v, i, t);                                         // rare
}

// One trilinear sample in two halves, so that several samples' gathers can be in flight together:
//   sample_issue : cell search on the three axes, address, the four 16-byte (32-byte for f64 cubes) corner-pair loads
//   sample_finish: weights, f32->f64 conversion, the 16 FMAs                    (weights/corner order: _rgi.py:490-498)
template <typename T2>
struct PendingSample {
    T2 v[8];              // corners in (y,x,z) lexicographic order
    double ty, tx, tz;
};

// The horizontal half of a sample: cell search on x / y (cell_xy), the element offset of corner (iy, ix, iz) and the four 16-byte (32-byte
// for f64 cubes) corner-pair loads.  iz may be slice-uniform (a scalar) or per lane.
template <typename T2, bool IDX, bool NOCHECK>
__device__ __forceinline__ void gather_corners(const CubeView<T2>& c, const AxisTabs& m, double y, double x, int iz, PendingSample<T2>& s) {
    int iy, ix;
    cell_xy<IDX, NOCHECK>(m.ey, c.ny, y, c.y_lo, c.y_hi, c.inv_dy, c.exact_y, c.uni_y, iy, s.ty);
    cell_xy<IDX, NOCHECK>(m.ex, c.nx, x, c.x_lo, c.x_hi, c.inv_dx, c.exact_x, c.uni_x, ix, s.tx);
    const T2 *p00, *p01, *p10, *p11;
    if (c.small) {              // one 32-bit element offset against four uniform row bases
        // (iy nx + ix) as ONE full-rate v_mad_u32_u24: left to itself the compiler picks v_mad_u64_u32, a quarter-rate instruction
        // that holds the vector ALU for four issue slots
        unsigned col;
        asm("v_mad_u32_u24 %0, %1, %2, %3" : "=v"(col) : "v"((unsigned)iy), "s"((unsigned)c.nx), "v"((unsigned)ix));
        const unsigned off = (__umul24(col, (unsigned)c.nz) + (unsigned)iz) * (unsigned)sizeof(T2);
        const char* b = reinterpret_cast<const char*>(c.v);
        const size_t rowx = (size_t)c.nz * sizeof(T2), rowy = (size_t)c.nx * rowx;
        p00 = reinterpret_cast<const T2*>(b + off);
        p01 = reinterpret_cast<const T2*>(b + rowx + off);
        p10 = reinterpret_cast<const T2*>(b + rowy + off);
        p11 = reinterpret_cast<const T2*>(b + rowy + rowx + off);
    } else {
        p00 = c.v + ((int64_t)iy * c.nx + ix) * c.nz + iz;
        p01 = p00 + c.nz;
        p10 = p00 + (int64_t)c.nx * c.nz;
        p11 = p10 + c.nz;
    }
    s.v[0] = p00[0]; s.v[1] = p00[1];
    s.v[2] = p01[0]; s.v[3] = p01[1];
    s.v[4] = p10[0]; s.v[5] = p10[1];
    s.v[6] = p11[0]; s.v[7] = p11[1];
}

// MODE 0: generic kernels (three-entry z window).  MODE 1 (light march loop, a level's TOP sample or the ray's first sample):
// index-space x / y on exact axes, two-entry z window.  MODE 2 (light march loop, a sample strictly INSIDE model interval kz):
// its z cell is kz itself, one table entry, no select.  Any sample whose weight leaves [0,1] takes the exact search.
template <typename T2, int MODE = 0, bool NOCHECK = false>
__device__ __forceinline__ void sample_issue(const CubeView<T2>& c, const AxisTabs& m, double y, double x, double z, int kz,
                                             PendingSample<T2>& s) {
    const double2* ez = m.ez;
    int iz;
    if (MODE == 2) {
        const double2 e0 = ez[kz];                                              // kz <= nz-2: slice-uniform address
        iz = kz;
        s.tz = (z - e0.x) * e0.y;
        if (!(s.tz >= 0.0) || !(s.tz <= 1.0)) cell_exact(ez, c.nz, z, iz, s.tz);   // rare
    } else if (MODE == 1) window2_cell(ez, c.nz, z, kz, c.nz >= 4, iz, s.tz);
    else window_cell(ez, c.nz, z, kz, c.nz >= 4, iz, s.tz);
    gather_corners<T2, MODE != 0, NOCHECK>(c, m, y, x, iz, s);
}

// Generic kernels: scipy's own summation order, weight = ((1*wy)*wx)*wz over the corners in lexicographic order (_rgi.py:490-498).
template <typename T2>
__device__ __forceinline__ void sample_finish(const PendingSample<T2>& s, double& wet, double& hyd) {
    const double wy0 = 1.0 - s.ty, wx0 = 1.0 - s.tx, wz0 = 1.0 - s.tz;
    const double a00 = wy0 * wx0, a01 = wy0 * s.tx, a10 = s.ty * wx0, a11 = s.ty * s.tx;
    double sw = 0.0, sh = 0.0, k;
    k = a00 * wz0;  sw = fma((double)s.v[0].x, k, sw); sh = fma((double)s.v[0].y, k, sh);
    k = a00 * s.tz; sw = fma((double)s.v[1].x, k, sw); sh = fma((double)s.v[1].y, k, sh);
    k = a01 * wz0;  sw = fma((double)s.v[2].x, k, sw); sh = fma((double)s.v[2].y, k, sh);
    k = a01 * s.tz; sw = fma((double)s.v[3].x, k, sw); sh = fma((double)s.v[3].y, k, sh);
    k = a10 * wz0;  sw = fma((double)s.v[4].x, k, sw); sh = fma((double)s.v[4].y, k, sh);
    k = a10 * s.tz; sw = fma((double)s.v[5].x, k, sw); sh = fma((double)s.v[5].y, k, sh);
    k = a11 * wz0;  sw = fma((double)s.v[6].x, k, sw); sh = fma((double)s.v[6].y, k, sh);
    k = a11 * s.tz; sw = fma((double)s.v[7].x, k, sw); sh = fma((double)s.v[7].y, k, sh);
    wet = sw; hyd = sh;
}

// Light march loop: the same trilinear interpolant in nested-lerp form, a + t (b - a) along z, then x, then y: 7 lerps of
// (sub, fma) per field = 28 instructions instead of the 31 of the weight-product form (the f32 -> f64 conversions are exact, so
// the two forms agree to a few ulp of the interpolated value; the parity tests hold 1e-9 m on the integrated delay).
template <typename T2>
__device__ __forceinline__ void sample_finish_lerp(const PendingSample<T2>& s, double& wet, double& hyd) {
    auto lerp = [](double a, double b, double t) { return fma(t, b - a, a); };
    const double w00 = lerp((double)s.v[0].x, (double)s.v[1].x, s.tz), h00 = lerp((double)s.v[0].y, (double)s.v[1].y, s.tz);
    const double w01 = lerp((double)s.v[2].x, (double)s.v[3].x, s.tz), h01 = lerp((double)s.v[2].y, (double)s.v[3].y, s.tz);
    const double w10 = lerp((double)s.v[4].x, (double)s.v[5].x, s.tz), h10 = lerp((double)s.v[4].y, (double)s.v[5].y, s.tz);
    const double w11 = lerp((double)s.v[6].x, (double)s.v[7].x, s.tz), h11 = lerp((double)s.v[6].y, (double)s.v[7].y, s.tz);
    const double w0 = lerp(w00, w01, s.tx), h0 = lerp(h00, h01, s.tx);
    const double w1 = lerp(w10, w11, s.tx), h1 = lerp(h10, h11, s.tx);
    wet = lerp(w0, w1, s.ty); hyd = lerp(h0, h1, s.ty);
}

template <typename T2>
__device__ __forceinline__ void sample_cube(const CubeView<T2>& c, const AxisTabs& m, double y, double x, double z, int kz,
                                            double& wet, double& hyd) {
    PendingSample<T2> s;
    sample_issue(c, m, y, x, z, kz, s);
    sample_finish(s, wet, hyd);
}

// getTopOfAtmosphere (losreader.py:706-733): pos = xyz + h*los; repeat: pos += los*((h - height(pos))/factor).
// Straight restatement with the PROJ-formula height (used by the materialising API kernels rdr_top_of_atmosphere /
// rdr_build_ray, whose OUTPUT is positions: they match the reference's to ~1e-8 m).
__device__ __forceinline__ void toa_newton(double ox, double oy, double oz, double lx, double ly, double lz,
                                           double h, int iters, double factor, double& px, double& py, double& pz) {
    px = ox + h * lx; py = oy + h * ly; pz = oz + h * lz;
    for (int it = 0; it < iters; ++it) {
        const double hgt = ecef_height(px, py, pz);
        const double step = (h - hgt) / factor;
        px = px + lx * step; py = py + ly * step; pz = pz + lz * step;
    }
}

// Generic ray kernels: the same iteration carried on the scalar ray parameter t (pos = o + t*l): t0 = h,
// t += (h - height(o + t l)) / factor - identical in exact arithmetic to losreader.py:724-731, 2 live doubles per crossing
// instead of 6 - with the PROJ-formula height.
__device__ __forceinline__ double toa_newton_t(double ox, double oy, double oz, double lx, double ly, double lz,
                                               double h, int iters, double inv_factor) {
    double t = h;
#pragma unroll 1
    for (int it = 0; it < iters; ++it) {
        const double hgt = ecef_height(fma(t, lx, ox), fma(t, ly, oy), fma(t, lz, oz));
        t = fma(h - hgt, inv_factor, t);
    }
    return t;
}

// Top crossing of a model level on a light ray: getTopOfAtmosphere's 3 iterations (losreader.py:724-731,817-819) on the
// normalised ray parameter u, with the height from the ray's polynomial.  gain = su / cos_factor.  Used by BOTH passes
// (same instruction sequence -> identical crossings).
__device__ __forceinline__ double level_top_u(const double* hpoly, double hi, double su, double ou, double gain) {
    double u = fma(hi, su, ou);
#pragma unroll
    for (int it = 0; it < 3; ++it) u = fma(hi - poly5(hpoly, u), gain, u);
    return u;
}

// N independent level crossings carried side by side (same arithmetic per level as level_top_u).
template <int N>
__device__ __forceinline__ void level_top_u_n(const double* hpoly, const double* hi, double su, double ou, double gain, double* u) {
#pragma unroll
    for (int j = 0; j < N; ++j) u[j] = fma(hi[j], su, ou);
#pragma unroll
    for (int it = 0; it < 3; ++it) {
        double p[N];
#pragma unroll
        for (int j = 0; j < N; ++j) p[j] = hpoly[PN - 1];
#pragma unroll
        for (int n = PN - 2; n >= 0; --n) {
#pragma unroll
            for (int j = 0; j < N; ++j) p[j] = fma(p[j], u[j], hpoly[n]);
        }
#pragma unroll
        for (int j = 0; j < N; ++j) u[j] = fma(hi[j] - p[j], gain, u[j]);
    }
}

// ---- crossing polynomial ---------------------------------------------------------------------------------------------
// The (deliberately NOT converged) three-iteration crossing level_top_u(h) of a light ray is a composition of polynomials in
// the level height h, i.e. a smooth function of h; its degree-7 interpolant X(v) at the 8 Chebyshev nodes of the slice's
// level-top range, v = (h - hm) / hh in [-1, 1], reproduces it to < 1e-10 m along the ray up to 60 deg incidence, 7e-9 m at
// 70 deg and 3e-6 m at 85 deg under an 80 km model top (a crossing shifted by d moves a delay by ~1.5e-3 d; sweep:
// tools/ray_poly_probe.py).  Pass 1 fits it once per ray (8 x 21 FMAs + 64) and then every level crossing, in BOTH passes,
// is 7 FMAs instead of 21 - same polynomial, same level abscissae (LDS table xv), hence identical crossings in the two passes.
constexpr int PX = 8;
__device__ const double XPOLY_NODES[PX] = {0.98078528040323044913, 0.83146961230254523708, 0.55557023301960222474, 0.19509032201612826785,
                                           -0.19509032201612826785, -0.55557023301960222474, -0.83146961230254523708, -0.98078528040323044913};
__device__ const double XPOLY_VINV[PX][PX] = {   // [node j][power n]
    {-0.024864045922457250864, -0.025351161379823003334, 0.76980164952545230204, 0.78488295543032986324, -3.1779876260079822119, -3.2402480843731826008, 3.0614674589207181738, 3.1214451522580522856},
    {0.083522329739912365, 0.10045145186799834205, -2.5519026177451504679, -3.0691471822744071686, 9.6723408277623460247, 11.632825402935940496, -7.391036260090294049, -8.8891237283136355959},
    {-0.1870757203331861272, -0.33672740045197043705, 5.3803297424913405655, 9.6843376817517615307, -12.500767952508536122, -22.500787856406753567, 7.391036260090294049, 13.303513796840723793},
    {0.62841743651573101306, 3.2211615113525685678, -3.5982287742716423996, -18.443912220177554698, 6.0064147507541723095, 30.787866300500633524, -3.0614674589207181738, -15.692564486451687186},
    {0.62841743651573101306, -3.2211615113525685678, -3.5982287742716423996, 18.443912220177554698, 6.0064147507541723095, -30.787866300500633524, -3.0614674589207181738, 15.692564486451687186},
    {-0.1870757203331861272, 0.33672740045197043705, 5.3803297424913405655, -9.6843376817517615307, -12.500767952508536122, 22.500787856406753567, 7.391036260090294049, -13.303513796840723793},
    {0.083522329739912365, -0.10045145186799834205, -2.5519026177451504679, 3.0691471822744071686, 9.6723408277623460247, -11.632825402935940496, -7.391036260090294049, 8.8891237283136355959},
    {-0.024864045922457250864, 0.025351161379823003334, 0.76980164952545230204, -0.78488295543032986324, -3.1779876260079822119, 3.2402480843731826008, 3.0614674589207181738, -3.1214451522580522856}};

__device__ __forceinline__ double poly7(const double* c, double v) {
    double r = fma(c[7], v, c[6]);
    r = fma(r, v, c[5]);
    r = fma(r, v, c[4]);
    r = fma(r, v, c[3]);
    r = fma(r, v, c[2]);
    r = fma(r, v, c[1]);
    return fma(r, v, c[0]);
}

// hm / hh: centre and half width of the slice's level-top range (slice-uniform, fill_tables).  Two nodes per trip: their
// Newton chains are independent and fill each other's dependent-issue bubbles.
__device__ __forceinline__ void fit_crossing_poly(const double* hpoly, double su, double ou, double gain, double hm, double hh, double* xc) {
#pragma unroll
    for (int n = 0; n < PX; ++n) xc[n] = 0.0;
#pragma unroll 1
    for (int j = 0; j < PX; j += 2) {
        double hn[2], un[2];
        hn[0] = fma(hh, XPOLY_NODES[j], hm); hn[1] = fma(hh, XPOLY_NODES[j + 1], hm);
        level_top_u_n<2>(hpoly, hn, su, ou, gain, un);
#pragma unroll
        for (int n = 0; n < PX; ++n) xc[n] = fma(XPOLY_VINV[j + 1][n], un[1], fma(XPOLY_VINV[j][n], un[0], xc[n]));
    }
}

// Workspace record handed from pass 1 (crossings_kernel) to pass 2 (march_kernel): one column per ray SLOT
// (slot = local tile * 256 + thread), field-major so every field access is a perfectly coalesced 512 B per wave:
//   ws[f * nslots + slot],  WS_NFIELDS = 29 fields = 232 B per ray - for every ray, light or generic
//   light ray:   0..5 h(u) | 6..11 lat(u) | 12..17 lon(u) monomial coefficients (index space on exact axes) | 18..25 crossing
//                polynomial X(v) | 26 |l|/su (metres per unit of u; > 0) | 27, 28 u at the bottom / top of the first level
//   generic ray: 0..2 origin ECEF | 3..5 look vector | 6 lat0 | 7 lon0 | 8..11 sin/cos lat0, sin/cos lon0 | 12 index of the ray's
//                column in the SIDE buffer, -1: none | 26 = 0.0, which is what marks the ray as generic
// Side buffer (generic rays only, usually empty): side[k * side_cap + idx] = ray parameter t of level crossing k = 0..K.  A
// generic ray that finds the side buffer full (idx -1) has its crossings recomputed by pass 2 - slower, never wrong.
constexpr int WS_POLY_H = 0, WS_POLY_LAT = 6, WS_POLY_LON = 12, WS_XPOLY = 18, WS_SCALE = 26, WS_U0 = 27, WS_U1 = 28, WS_NFIELDS = 29;
constexpr int WS_ORIGIN = 0, WS_LOS = 3, WS_LAT0 = 6, WS_LON0 = 7, WS_S0 = 8, WS_C0 = 9, WS_SL0 = 10, WS_CL0 = 11, WS_SIDE = 12;

struct RayParams {
    // geometry
    int64_t n;
    int origin_mode, los_mode;
    int64_t nx, ny;
    const double* xpts; const double* ypts;
    const double* lat; const double* lon; const double* xyz;
    const double* los; const double* inc; const double* hd;
    double inc0, hd0;
    // slices: a batch is nslices >= 1 height slices of the SAME origins (delay.py:256: one slice per output height).  Tiles are
    // numbered slice-major (tile t belongs to slice t / tiles_per_slice); everything the reference reduces over a slice -
    // per-level maxima, flags, nParts - is kept per slice, so a batched launch gives exactly what slice-by-slice launches give.
    int nslices; int64_t tiles_per_slice;
    const double* hts;                 // [nslices] slice heights, or nullptr: one slice at `ht`
    const double* ht_ray;              // [n] per-ray origin heights (one slice only), or nullptr.  The slice's level table is then built
                                       // for `ht` <= min(ht_ray); ray i joins it at its own first contributing level (first_level)
    int64_t los_stride;                // rays between consecutive slices in los / inc / hd (0: the same arrays for every slice)
    double ht, zref, max_seg;
    // batch-global state
    unsigned long long* maxlen_bits;   // [nslices][MAX_LEVELS] per-level max ray length (bit pattern of a non-negative double); nullptr: no reduction
    int* flags;                        // [nslices] RDR_FLAG_* bits (OR-reduced)
    const int* nparts_override;        // [nslices][MAX_LEVELS] or nullptr -> ceil(maxlen/max_seg)+1
    int* nslow;                        // number of rays the static classification sent to the generic (slow) kernels
    int* tile_ctr;                     // [8] per-XCD next-tile counters of this launch (zeroed by the host)
    // pass 1 -> pass 2 workspace (this launch covers tiles [tile_begin, tile_begin + tile_count))
    double* ws; int64_t nslots;
    double* side; int64_t side_cap;    // crossings of the generic rays (compact columns), int* side_ctr = next free column
    int* side_ctr;
    int64_t tile_begin, tile_count;
    // outputs
    double* wet; double* hyd;
    // tiling of the whole batch
    int64_t ntiles; int tiles_x;
    int stage_f64;                     // light march on f64 cubes: stage every level's footprint of a wave in LDS (0: the direct gathers)
};

// The slice-uniform level table of build_ray (losreader.py:785-808), computed by one thread into LDS from the LDS z table.
__device__ inline int build_levels(const double2* ez, int nz, double ht, double zref, double* s_lo, double* s_hi, int* s_kz) {
    int K = 0;
    const double ztop = ez[nz - 1].x;
    for (int zz = 0; zz < nz - 1; ++zz) {
        double lo = ez[zz].x, hi = ez[zz + 1].x;
        if (hi == ztop) hi -= 0.01;
        if (hi < ht || lo >= zref) continue;
        if (lo < ht) lo = ht;
        if (hi > zref) hi = zref;
        if (fabs(hi - lo) < 1.0) continue;
        if (K < MAX_LEVELS) { s_lo[K] = lo; s_hi[K] = hi; s_kz[K] = zz; }
        ++K;
    }
    return min(K, MAX_LEVELS);
}

// Per-ray origin heights (RayParams::ht_ray; no reference semantics - DESIGN.md 5c): the reference's level tests
// (losreader.py:785-808) applied with the RAY's height against the slice table built for the lowest one.  Returns the first entry
// k of the table the ray contributes to (K: none) and its clipped bottom lo_i = max(z_kz, hti); every later entry contributes
// unclipped.  Both passes call it with the same inputs.
__device__ __forceinline__ int first_level(const double2* ez, int nz, const double* s_lo, const double* s_hi, const int* s_kz, int K,
                                           double hti, double& lo_i) {
    const double ztop = ez[nz - 1].x;
    // `if high_ht < ht: continue`: the interval tops (before the zref clip) ascend with k, so the entries passing this test are a
    // suffix of the table - found by bisection; (hti NaN: every comparison fails, the search ends at 0 and the ray is NaN anyway)
    int a = 0, b = K;
    while (a < b) {
        const int mid = (a + b) >> 1;
        double hr = ez[s_kz[mid] + 1].x;
        if (hr == ztop) hr -= 0.01;
        if (hr < hti) a = mid + 1; else b = mid;
    }
    for (int k = a; k < K; ++k) {                            // then the first one at least 1 m thick above the ray's own bottom
        const double lo = fmax(s_lo[k], hti);                // `if low_ht < ht: low_ht = ht`
        if (fabs(s_hi[k] - lo) < 1.0) continue;
        lo_i = lo;
        return k;
    }
    lo_i = 0.0;
    return K;
}

// Shared LDS layout of the two ray kernels.  Axis tables are (g[i], 1/(g[i+1]-g[i])) pairs; an x / y axis that is uniform to
// round-off (CubeView::exact_*) needs no table at all (cell_xy works from g0 and the spacing), so the usual lat/lon or LCC grid
// costs 16 B per z level only - a CONUS-sized HRRR grid (1059 x 1799 nodes) would otherwise take 69 KB per workgroup.
// Pass 2's per-level record (slice loop of the light marcher): everything a level needs that is the same for every ray of the slice,
// packed so that ONE LDS address register serves all the reads of a level (as separate arrays every read cost a v_mov of its own).
struct __attribute__((aligned(16))) LevelRec {
    double xv, hs;          // abscissa of the level's top in the crossing polynomial; 0.5e-6/(nParts-1)
    double step, zmid;      // 1/(nParts-1); node g[zb+1] the level's top sits on, zb = window2_base(nz, kz)
    double r0, r1;          // 1/(g[zb+1]-g[zb]), 1/(g[zb+2]-g[zb+1]): the two entries of the top sample's z window
    double gk, rk;          // g[kz], 1/(g[kz+1]-g[kz]): the model interval of the level's interior samples
    int npkz, pad[3];       // nParts | kz << 17
};
static_assert(sizeof(LevelRec) == 80, "LevelRec is read with fixed LDS offsets");

struct RaySmem {
    AxisTabs ax;            // axis tables (ey / ex are null for exact axes)
    double2* tab2;          // backing store of the tables
    double* lo; double* hi; // level table
    unsigned long long* mxcol; // [nz][MXCOLS] per-level running maxima of the workgroup, one column per lane%MXCOLS (pass 1)
    LevelRec* lev;          // [nz] pass 2 only: the same bytes as mxcol (80 <= 8 * MXCOLS per level)
    double* step;           // [nz] 1/(nParts-1) (pass 2)
    double* hs;             // [nz] 0.5e-6/(nParts-1): half the trapezoid weight per unit of ray length (pass 2)
    double* trig;           // [64] (sin, cos) of the tile's 16 row latitudes, then of its 16 column longitudes (pass 1, GRID rays);
                            // [64..112) LCC cubes: rho of the 16 rows, then (sin, cos) theta of the 16 columns
    double* xv;             // [nz] abscissa of level k's top in the crossing polynomial: (hi[k] - xmap[0]) / xmap[1]
    double* xmap;           // [2] centre and half width of the level-top range hi[1] .. hi[K-1]
    int* kz; int* np; int* K;
};
constexpr int MXCOLS = 16;
static_assert(sizeof(LevelRec) <= 8 * MXCOLS, "LevelRec aliases a level's mxcol columns");
__host__ __device__ inline int64_t table_nodes(int64_t ny, int64_t nx, int64_t nz, int exact_y, int exact_x) {
    return (exact_y ? 0 : ny) + (exact_x ? 0 : nx) + nz;
}
__device__ __forceinline__ RaySmem carve_smem(unsigned char* raw, int ny, int nx, int nz, int exact_y, int exact_x) {
    RaySmem m;
    const int nyt = exact_y ? 0 : ny, nxt = exact_x ? 0 : nx;
    m.tab2 = reinterpret_cast<double2*>(raw);                 // 16-byte aligned for ds_read_b128
    m.ax.ey = exact_y ? nullptr : m.tab2;
    m.ax.ex = exact_x ? nullptr : m.tab2 + nyt;
    m.ax.ez = m.tab2 + nyt + nxt;
    m.lo = reinterpret_cast<double*>(m.tab2 + nyt + nxt + nz);
    m.hi = m.lo + nz;
    m.mxcol = reinterpret_cast<unsigned long long*>(m.hi + nz);
    m.lev = reinterpret_cast<LevelRec*>(m.mxcol);
    m.step = reinterpret_cast<double*>(m.mxcol + (size_t)nz * MXCOLS);
    m.hs = m.step + nz;
    m.trig = m.hs + nz;
    m.xv = m.trig + 112;
    m.xmap = m.xv + nz;
    m.kz = reinterpret_cast<int*>(m.xmap + 2);
    m.np = m.kz + nz;
    m.K = m.np + nz;
    return m;
}

// bytes carve_smem lays out (the launch's dynamic LDS size) - keep the two in step
inline size_t ray_smem_bytes(int64_t ny, int64_t nx, int64_t nz, int exact_y, int exact_x) {
    return (size_t)table_nodes(ny, nx, nz, exact_y, exact_x) * 16     // axis tables
           + (size_t)nz * 8 * 2                   // lo, hi
           + (size_t)nz * 8 * MXCOLS              // mxcol
           + (size_t)nz * 8 * 2 + 112 * 8         // step, hs, trig
           + (size_t)nz * 8 + 16                  // xv, xmap
           + (size_t)nz * 4 * 2 + 16;             // kz, np, K
}

// Axis tables: once per workgroup.
template <typename T2>
__device__ __forceinline__ void fill_axes(const CubeView<T2>& c, const RaySmem& m) {
    const int tid = threadIdx.x;
    auto fill = [&](double2* dst, const double* g, int n) {
        for (int i = tid; i < n; i += BLOCK) {
            double2 e; e.x = g[i]; e.y = (i == n - 1) ? 0.0 : 1.0 / (g[i + 1] - g[i]);
            dst[i] = e;
        }
    };
    if (m.ax.ey) fill(const_cast<double2*>(m.ax.ey), c.axes, c.ny);
    if (m.ax.ex) fill(const_cast<double2*>(m.ax.ex), c.axes + c.ny, c.nx);
    fill(const_cast<double2*>(m.ax.ez), c.axes + c.ny + c.nx, c.nz);
    __syncthreads();
}

// Level table of one slice (height ht): whenever a workgroup moves on to a tile of another slice.  Called by every thread.
__device__ __forceinline__ int fill_levels(int nz, const RaySmem& m, double ht, double zref) {
    int tid = threadIdx.x;
    asm volatile("" : "+v"(tid));
    __syncthreads();                                   // the previous slice's readers are done with the tables
    if (tid == 0) {
        const int K = build_levels(m.ax.ez, nz, ht, zref, m.lo, m.hi, m.kz);
        *m.K = K;
        // range of the level tops the crossing polynomial has to cover (levels 1 .. K-1; level 0 has its own iteration)
        const double a = K > 1 ? m.hi[1] : (K > 0 ? m.hi[0] : 0.0), b = K > 1 ? m.hi[K - 1] : a;
        m.xmap[0] = 0.5 * (a + b);
        m.xmap[1] = fmax(0.5 * (b - a), 1.0);
    }
    __syncthreads();
    const int K = *m.K;
    const double hm = m.xmap[0], ihh = 1.0 / m.xmap[1];
    for (int k = tid; k < K; k += BLOCK) m.xv[k] = (m.hi[k] - hm) * ihh;
    __syncthreads();
    return K;
}

// XCD-aware tile walk: workgroup b runs on XCD b%8 -> each XCD sweeps one contiguous band of tiles (its cube slab stays in
// that XCD's L2).  Within a band the workgroups take tiles from a device counter (one atomic per tile), so no workgroup is
// left holding a fixed share while others have finished.
struct TileWalk {
    int64_t chunk; int xcd; int* ctr; int* slot;
    __device__ __forceinline__ TileWalk(int64_t count, int* counters, int* lds_slot) {
        xcd = blockIdx.x & 7; chunk = (count + 7) / 8; ctr = counters + xcd; slot = lds_slot;
    }
    __device__ __forceinline__ bool next(int64_t count, int64_t& local) {
        __syncthreads();
        if (threadIdx.x == 0) *slot = atomicAdd(ctr, 1);
        __syncthreads();
        const int64_t tt = *slot;
        local = (int64_t)xcd * chunk + tt;
        return tt < chunk && local < count;
    }
};

// Sines / cosines (and, on a conic cube, the origin projection terms) of one of a tile's 16 row latitudes (lane < 16) or 16
// column longitudes, into the tile's LDS table.  (Inlined on purpose: as a real call - tried to keep the libm constants from being
// hoisted and spilled - the kernel returned corrupted slice flags on LCC cubes.)
template <bool LCC>
__device__ __forceinline__ void tile_trig(double v, int lane, const LccParams& proj, double* trig) {
    double sv, cv;
    sincos(v * DEG_TO_RAD, &sv, &cv);
    trig[2 * lane] = sv; trig[2 * lane + 1] = cv;
    if (LCC) {                                             // the origin's projection: per row / per column as well
        if (lane < 16) trig[64 + lane] = lcc_rho(proj, sv, cv);
        else { double st, ct; lcc_theta(proj, v, st, ct); trig[80 + 2 * (lane - 16)] = st; trig[81 + 2 * (lane - 16)] = ct; }
    }
}

// ---- pass 1: per-ray set-up + level crossings (build_ray, losreader.py:772-835) ------------------------------------
// Optional outputs (both may be on): the per-level batch maximum of the ray length + flags (what delay.py:283,306-311
// reduce over the slice) and the workspace record for pass 2.
// SLOW = false: the light-fp64 path, skips (but counts) the rays the static classification rejects;
// SLOW = true : generic geodesy, processes ONLY those rays, exits at once when there are none.
// LCC (light kernel only): the cube is on a Lambert-conformal-conic grid; a separate instantiation so that the projection's
// pow/tan/sincos code does not weigh on the register allocation of the lon/lat one.
// OM (light kernel only): how the batch is given, fixed at compile time so that the hot instantiations carry none of the libm
// code of the other input forms (atan / atan2 of XYZ origins, four sincos of inc/heading look vectors) - with everything in one
// body the compiler hoisted so many of their invariants that the per-level loop spilled (220 B of scratch per lane in round 1):
//   1: GRID origins + per-pixel look vectors;  2: GRID origins + incidence / heading (arrays or scalars) or zenith;  0: any.
// PR (light kernel only): per-ray origin heights (P.ht_ray) - a separate instantiation so that the slice kernels carry none of it; the
// generic kernel looks at P.ht_ray at run time.
template <typename T2, bool SLOW, bool LCC = false, int OM = 0, bool PR = false>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(SLOW ? 1 : 4, SLOW ? 8 : 4))) void crossings_kernel(CubeView<T2> c, RayParams P, LccParams proj) {
    static_assert(!SLOW || (OM == 0 && !PR), "the generic kernel takes every input form");
    const bool per_ray = PR || (SLOW && P.ht_ray != nullptr);
    if (SLOW && *P.nslow == 0) return;
    const int origin_mode = OM != 0 ? 0 : P.origin_mode;
    const int los_mode = OM == 1 ? 0 : P.los_mode;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const RaySmem m = carve_smem(smem_raw, c.ny, c.nx, c.nz, c.exact_y, c.exact_x);
    fill_axes(c, m);
    const int tid = threadIdx.x;
    const bool reduce = P.maxlen_bits != nullptr;
    // per-level maximum of the ray length over a slice (delay.py:283): every lane folds its length into column lane%16 of
    // the workgroup's LDS table with one ds_max_u64 (non-negative doubles order like their bit patterns); the columns are
    // combined when the workgroup leaves the slice (flush).  NaN lengths are left out here and reported through the flags,
    // which is how ndarray.max's NaN poisoning is reproduced on the host side.
    int my_flags = 0;
    int K = 0, slice = -1;
    double ht = P.ht;
    auto flush = [&]() {                                   // called by every thread of the workgroup
        if (!reduce || slice < 0) return;
        __syncthreads();
        int tf = threadIdx.x;
        asm volatile("" : "+v"(tf));                       // (opaque: no address derived from it is kept live across the tile loop)
        for (int k = tf; k < K; k += BLOCK) {
            unsigned long long v = 0ULL;
#pragma unroll
            for (int cc = 0; cc < MXCOLS; ++cc) v = max(v, m.mxcol[k * MXCOLS + cc]);
            atomicMax(&P.maxlen_bits[(int64_t)slice * MAX_LEVELS + k], v);
        }
        int f = my_flags;
        for (int off = 32; off > 0; off >>= 1) f |= __shfl_xor(f, off, 64);
        if ((tf & 63) == 0 && f) atomicOr(P.flags + slice, f);
        my_flags = 0;
    };
    TileWalk walk(P.tile_count, P.tile_ctr, m.K + 2);
    int64_t lt;
    while (walk.next(P.tile_count, lt)) {
        const int64_t tg = P.tile_begin + lt;              // tile of the batch; t: tile within its slice
        const int sl = (int)(tg / P.tiles_per_slice);
        const int64_t t = tg - (int64_t)sl * P.tiles_per_slice;
        if (sl != slice) {                                 // (workgroup-uniform) first tile, or the walk has reached the next slice
            flush();
            slice = sl;
            ht = P.hts ? P.hts[sl] : P.ht;
            K = fill_levels(c.nz, m, ht, P.zref);
            int tz = threadIdx.x;
            asm volatile("" : "+v"(tz));
            if (reduce) for (int k = tz; k < K * MXCOLS; k += BLOCK) m.mxcol[k] = 0ULL;
            __syncthreads();
        }
        // The thread index as the tile body sees it: an opaque per-tile copy, so that nothing derived from it (LDS addresses, record
        // pointers, row / column offsets) is hoisted out of the tile loop and kept live - or spilled - across the polynomial fit.
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));
        const int64_t lbase = (int64_t)sl * P.los_stride;  // this slice's block of the look-vector / incidence arrays
        int64_t i, row = 0, col = 0; bool active;
        if (origin_mode == 0) {
            const int64_t ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
            row = ty * TILE + (tl >> 4); col = tx * TILE + (tl & 15);
            active = row < P.ny && col < P.nx;
            i = row * P.nx + col;
        } else {
            i = t * BLOCK + tl;
            active = i < P.n;
        }
        // ---- origin: llh -> ECEF (delay.py:262-267)
        double lat = 0, lon = 0, ox = qnan(), oy = qnan(), oz = qnan();
        double hti = ht;                                   // the ray's origin height: the slice's, or its own
        if (PR || SLOW) { if (per_ray && active) hti = P.ht_ray[i]; }
        RayBase base;
        if (!SLOW && origin_mode == 0) {
            // a tile has 16 distinct latitudes and 16 distinct longitudes: 32 lanes take the sines / cosines for everybody
            // (the previous tile's readers are past the barriers of walk.next())
            if (tl < 32) {
                const int64_t ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
                const int64_t r = ty * TILE + (tl & 15), cc = tx * TILE + (tl & 15);
                const double v = tl < 16 ? (r < P.ny ? P.ypts[r] : 0.0) : (cc < P.nx ? P.xpts[cc] : 0.0);
                tile_trig<LCC>(v, tl, proj, m.trig);
            }
            __syncthreads();
            if (active) { lat = P.ypts[row]; lon = P.xpts[col]; }
            base.lat0 = lat; base.lon0 = lon;
            base.s0 = active ? m.trig[2 * (tl >> 4)] : 0.0; base.c0 = active ? m.trig[2 * (tl >> 4) + 1] : 1.0;
            base.sl0 = active ? m.trig[32 + 2 * (tl & 15)] : 0.0; base.cl0 = active ? m.trig[33 + 2 * (tl & 15)] : 1.0;
            if (active) {                                          // lla2ecef (geodesy.h) with the shared sines / cosines
                const double N = WGS84_A / sqrt(1.0 - WGS84_ES * base.s0 * base.s0);
                ox = (N + hti) * base.c0 * base.cl0;
                oy = (N + hti) * base.c0 * base.sl0;
                oz = (N * (1.0 - WGS84_ES) + hti) * base.s0;
            }
        } else {
            if (active) {
                if (origin_mode == 0) { lat = P.ypts[row]; lon = P.xpts[col]; }
                else if (P.lat) { lat = P.lat[i]; lon = P.lon[i]; }
                if (origin_mode == 2) {
                    ox = P.xyz[3 * i]; oy = P.xyz[3 * i + 1]; oz = P.xyz[3 * i + 2];
                    if (!P.lat) { double h0_; ecef2lla(ox, oy, oz, lon, lat, h0_); }   // frame for the delta lat/lon formulas
                } else lla2ecef(lat, lon, hti, ox, oy, oz);
            }
            base = make_base(lat, lon);
        }
        // ---- look vector (delay.py:270)
        double lx = qnan(), ly = qnan(), lz = qnan();
        if (active) {
            // (inc / heading: the origin's own sines / cosines are the ones inc_hd_to_ecef would compute again)
            if (los_mode == 0) { const double* lp = P.los + 3 * (lbase + i); lx = lp[0]; ly = lp[1]; lz = lp[2]; }
            else if (los_mode == 1) inc_hd_to_ecef_sc(P.inc[lbase + i], P.hd ? P.hd[lbase + i] : P.hd0, base.s0, base.c0, base.sl0, base.cl0, lx, ly, lz);
            else if (los_mode == 2) inc_hd_to_ecef_sc(P.inc0, P.hd0, base.s0, base.c0, base.sl0, base.cl0, lx, ly, lz);
            else { lx = base.c0 * base.cl0; ly = base.c0 * base.sl0; lz = base.s0; }   // zenith (losreader.py:302-316)
        }
        // |l| (1 for unit look vectors): ray length between two crossings = (t_hi - t_lo) * |l|   (losreader.py:821)
        const double nl2 = fma(lx, lx, fma(ly, ly, lz * lz));
        const double nl = nl2 * rsq_nr<2>(nl2);
        // Static classification: may the ray polynomials be used along the WHOLE ray?  gam bounds the angular travel
        // (t_max <= (zref-ht)/cos(inc) because the local zenith angle of a straight ray decreases with height); the ray must
        // stay clear of the poles (cos(lat) > gam + 0.02), turn by less than 0.2 rad in longitude (gam / (cos(lat) - gam)) and
        // be shorter than 0.035 rad - inside that region the degree-5 interpolants are good to 1e-7 m in height and 2e-5 m
        // on the ground (tools/ray_poly_probe.py), which moves delays by < 1e-10 m - and must not reach the +-180 meridian
        // (the light path does not wrap longitudes).
        const double cosi = (lx * base.c0 * base.cl0 + ly * base.c0 * base.sl0 + lz * base.s0) / nl;
        const double gam = (P.zref - hti) / (cosi * 6.3e6);
        // LCC cubes: spherical cones only (HRRR), and the node projections use short series in log(t/t_origin) <= gam / cos(lat)
        // (geodesy_fast.h); an ellipsoidal cone goes to the generic kernels (lcc_forward) ray by ray.
        // The +-180 deg meridian: the light path carries longitude UNWRAPPED (origin + small delta), the reference wraps every
        // sample into (-180, 180] (atan2).  The two agree whenever no sample crosses the meridian (origin more than 5 deg away from
        // it - a ray travels < 0.2 rad = 11.5 deg only at high latitude, where the other clauses bite first... kept as before), and
        // ALSO when a crossing sample is outside the cube either way: a lon/lat cube whose x axis ends at or before +180 and starts
        // more than 12 deg after -180 (wrap_pos) gives NaN for an unwrapped 180.x (beyond the axis) and for the reference's
        // wrapped -179.x (before it) alike; mirrored for origins near -180 (wrap_neg).  Dateline scenes (Fiji, Chukotka, the
        // Aleutians) then stay on the light path.  On a conic / polar-stereographic cube only the cone's own cut meridian
        // lam0 +- 180 matters (theta = n (lam - lam0) is wrapped there); geographic +-180 is an ordinary meridian.
        bool lon_ok;
        if (proj.kind == 1) {
            double dl = lon - proj.lam0 * RAD_TO_DEG;
            dl -= 360.0 * rint(dl * (1.0 / 360.0));
            lon_ok = fabs(dl) + 12.0 < 180.0;
        } else {
            const bool wrap_pos = (c.x_hi <= 180.0) & (c.x_lo > -168.0), wrap_neg = (c.x_lo >= -180.0) & (c.x_hi < 168.0);
            lon_ok = (fabs(lon) + 5.0 < 180.0) || ((fabs(lon) <= 180.0) && (lon > 0.0 ? wrap_pos : wrap_neg));
        }
        const bool fast_ok = !active || ((cosi > 0.05) && (base.c0 > gam + 0.02) && (gam < 0.2 * (base.c0 - gam)) && (gam < 0.035) &&
                                         lon_ok && (proj.kind != 1 || (proj.e == 0.0 && gam < 0.09 * base.c0)));   // (run-time test: the generic kernels must classify identically)
        const int64_t slot = lt * BLOCK + tl;
        if (!SLOW) {
            const unsigned long long slow_mask = __ballot(!fast_ok);
            if (slow_mask && (tl & 63) == 0) atomicAdd(P.nslow, (int)__popcll(slow_mask));
        }
        const bool mine = SLOW ? !fast_ok : fast_ok;      // lanes this instantiation is responsible for
        if (SLOW && !__any(mine)) continue;               // (wave-uniform) nothing to mop up in this wave
        // per-ray heights: the ray's first level of the slice table and that level's clipped bottom (k0 == K: no level at all -
        // the ray counts for nothing and its delays are 0, as a slice without levels).  A height below the table's is a caller
        // error (flag 32: the slice's outputs are NaN).
        int k0 = 0; double lo_first = K > 0 ? m.lo[0] : 0.0;
        if (PR || SLOW) {
            if (per_ray) {
                k0 = first_level(m.ax.ez, c.nz, m.lo, m.hi, m.kz, K, hti, lo_first);
                if (active && mine && hti < ht) my_flags |= 32;
            }
        }
        const bool cnt = active && mine && k0 < K;
        // this lane's column of the per-level maxima; derived afresh per tile (an address kept live across the polynomial fit,
        // the register-hungriest stretch of the kernel, is the one value the allocator had to spill)
        int mxcol_idx = tl & (MXCOLS - 1);
        asm volatile("" : "+v"(mxcol_idx));
        unsigned long long* const mxc = m.mxcol + mxcol_idx;
        double* const w = P.ws ? P.ws + slot : nullptr;
        const int64_t ns = P.nslots;
        if constexpr (SLOW) {
            // a column of the side buffer for this ray's K+1 crossings: one device atomic per wave, columns handed out in lane
            // order; -1 when the buffer is full (pass 2 then recomputes the crossings)
            int64_t sidx = -1;
            if (w) {
                const unsigned long long mm = __ballot(mine);
                const int lane = tl & 63;
                int base = 0;
                if (lane == 0) base = atomicAdd(P.side_ctr, (int)__popcll(mm));
                base = __shfl(base, 0, 64);
                const int64_t cand = (int64_t)base + __popcll(mm & ((1ULL << lane) - 1ULL));
                if (mine && cand < P.side_cap) sidx = cand;
            }
            double* const sd = (sidx >= 0) ? P.side + sidx : nullptr;
            if (w && mine) {
                w[(int64_t)WS_SCALE * ns] = 0.0;
                w[(int64_t)(WS_ORIGIN + 0) * ns] = ox; w[(int64_t)(WS_ORIGIN + 1) * ns] = oy; w[(int64_t)(WS_ORIGIN + 2) * ns] = oz;
                w[(int64_t)(WS_LOS + 0) * ns] = lx; w[(int64_t)(WS_LOS + 1) * ns] = ly; w[(int64_t)(WS_LOS + 2) * ns] = lz;
                w[(int64_t)WS_LAT0 * ns] = lat; w[(int64_t)WS_LON0 * ns] = lon;
                w[(int64_t)WS_S0 * ns] = base.s0; w[(int64_t)WS_C0 * ns] = base.c0;
                w[(int64_t)WS_SL0 * ns] = base.sl0; w[(int64_t)WS_CL0 * ns] = base.cl0;
                w[(int64_t)WS_SIDE * ns] = (double)sidx;
            }
            double t_hi = 0.0, inv_cosf = 1.0;
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                if (k < k0) continue;                // (per-ray heights: the ray joins the slice's level table at k0)
                const double lo = k == k0 ? lo_first : m.lo[k], hi = m.hi[k];
                // first interval: cos_factor is None -> 10 iterations with factor 1 for both ends (losreader.py:812-825);
                // later intervals reuse the previous top as their bottom (losreader.py:811-812)
                double t_lo = t_hi;
                if (k == k0) t_lo = toa_newton_t(ox, oy, oz, lx, ly, lz, lo, 10, 1.0);
                t_hi = toa_newton_t(ox, oy, oz, lx, ly, lz, hi, k == k0 ? 10 : 3, inv_cosf);
                const double L = fabs(t_hi - t_lo) * nl;                                   // np.linalg.norm(high - low) (losreader.py:821): positive also for the
                                                                                            // reversed segment of an origin ABOVE zref inside zref's model interval
                if (k == k0) inv_cosf = L / (hi - lo);                                      // 1/cos_factor, losreader.py:824-825
                if (sd) {
                    if (k == k0) sd[0] = t_lo;
                    sd[(int64_t)(k + 1) * P.side_cap] = t_hi;
                }
                if (reduce) {
                    if (cnt) my_flags |= (L != L) ? 1 : 2;
                    atomicMax(&mxc[k * MXCOLS], (unsigned long long)__double_as_longlong((cnt && L > 0.0) ? L : 0.0));
                    if (k == k0 && cnt) {            // first sample of the ray (fraction 0)
                        const double h0 = ecef_height(fma(t_lo, lx, ox), fma(t_lo, ly, oy), fma(t_lo, lz, oz));
                        if (!(h0 < c.z_lo)) my_flags |= 4;
                    }
                    if (k == K - 1 && cnt) {         // last sample of the ray (fraction 1)
                        const double h1 = ecef_height(fma(t_hi, lx, ox), fma(t_hi, ly, oy), fma(t_hi, lz, oz));
                        if (!(h1 > c.z_hi)) my_flags |= 8;
                    }
                }
            }
        } else {
            // ---- light rays: fit h(u), lat(u), lon(u) once, then everything is polynomial arithmetic.
            // Range of the ray parameter any Newton iterate / sample can take: iterates start at t0 = level height
            // (>= ht) and move monotonically to the crossing, which lies in [0, (zref - ht)/cos(inc)].
            // (an origin ABOVE zref inside zref's model interval - the reference's one reversed segment, low_ht = ht > high_ht = zref - starts its
            // iterates at t0 = ht and ends them at a NEGATIVE ray parameter: both belong to the range.  Unchanged for ht <= zref.)
            const double t_end = (P.zref - hti) / (cosi * nl);
            const double t_a = fmin(fmin(0.0, hti), t_end) - 1.0;
            const double t_b = fmax(fmax(P.zref, hti), t_end) + 1.0;
            const double half = 0.5 * (t_b - t_a), mid = 0.5 * (t_b + t_a);
            const double su = 1.0 / half, ou = -mid * su;
            RayPoly q;
            LccOrigin lorg = {0.0, 0.0, 0.0};
            const bool shared_lcc = LCC && !SLOW && origin_mode == 0;
            if (shared_lcc) { lorg.rho = m.trig[64 + (tl >> 4)]; lorg.st = m.trig[80 + 2 * (tl & 15)]; lorg.ct = m.trig[81 + 2 * (tl & 15)]; }
            fit_ray_poly<LCC>(base, ox, oy, oz, lx, ly, lz, mid, half, proj, shared_lcc ? &lorg : nullptr, q);
            const double scale = nl * half;                           // ray length per unit of u
            if (w && mine) {
                // exact-uniform axes: hand pass 2 the INDEX-space coordinate (v - g0) * (n-1)/(g[n-1]-g0) directly (cell_xy<true>)
                if (c.exact_y) {
                    q.lat[0] = (q.lat[0] - c.y_lo) * c.inv_dy;
#pragma unroll
                    for (int n = 1; n < PN; ++n) q.lat[n] *= c.inv_dy;
                }
                if (c.exact_x) {
                    q.lon[0] = (q.lon[0] - c.x_lo) * c.inv_dx;
#pragma unroll
                    for (int n = 1; n < PN; ++n) q.lon[n] *= c.inv_dx;
                }
#pragma unroll
                for (int n = 0; n < PN; ++n) {
                    w[(int64_t)(WS_POLY_H + n) * ns] = q.h[n];
                    w[(int64_t)(WS_POLY_LAT + n) * ns] = q.lat[n];
                    w[(int64_t)(WS_POLY_LON + n) * ns] = q.lon[n];
                }
                w[(int64_t)WS_SCALE * ns] = scale;
            }
            // getTopOfAtmosphere carried on u: u0 = u(h); u += (h - H(u)) * su / factor   (losreader.py:724-731)
            // Level 0 (ten plain Newton steps per end, losreader.py:770-777) sets the gain of every later level.
            double u_hi = 0.0, gain = su;
            double last_len = 0.0;      // a light ray's lengths are NaN for every level or for none (they all stem from one polynomial)
            if (PR ? (k0 < K) : (K > 0)) {
                const double lo = PR ? lo_first : m.lo[0], hi = PR ? m.hi[k0] : m.hi[0];
                double u_lo = fma(lo, su, ou);
                u_hi = fma(hi, su, ou);
#pragma unroll 1
                for (int it = 0; it < 10; ++it) {
                    u_lo = fma(lo - poly5(q.h, u_lo), su, u_lo);
                    u_hi = fma(hi - poly5(q.h, u_hi), su, u_hi);
                }
                const double L = fabs(u_hi - u_lo) * scale;                                    // a norm (losreader.py:821): an origin above zref in zref's own model
                                                                                               // interval gives ONE reversed segment (low_ht = ht > high_ht = zref) of positive length
                last_len = L;
                gain = su * (L / (hi - lo));                                                   // su / cos_factor, losreader.py:824-825
                if (w && mine) { w[(int64_t)WS_U0 * ns] = u_lo; w[(int64_t)WS_U1 * ns] = u_hi; }
                if (reduce) {
                    atomicMax(&mxc[PR ? k0 * MXCOLS : 0], (unsigned long long)__double_as_longlong((cnt && L > 0.0) ? L : 0.0));
                    if (cnt && !(poly5(q.h, u_lo) < c.z_lo)) my_flags |= 4;                    // first sample of the ray
                }
            }
            // Later levels: the three-iteration crossing as a polynomial of the level height (fit_crossing_poly)
            double xc[PX];
            fit_crossing_poly(q.h, su, ou, gain, m.xmap[0], m.xmap[1], xc);
            if (w && mine) {
#pragma unroll
                for (int n = 0; n < PX; ++n) w[(int64_t)(WS_XPOLY + n) * ns] = xc[n];
            }
            if (reduce) {
                // The level abscissae are read from LDS ahead of the ds_max of the previous pair: LDS operations complete in
                // order, so a read issued after an atomic would wait for it.
                // A lane that does not count (tile padding, a ray left to the generic kernel) contributes length 0 to every
                // level: its scale is zeroed once, here, instead of a select per level; v_max_f64(L, 0) then also maps a NaN
                // length (0 x NaN of such a lane, or a genuinely NaN ray - reported through the flags) to 0, so that the u64
                // maximum never sees a negative or non-finite bit pattern.
                // The metres-per-unit-u scale is folded into the polynomial once (8 multiplications), so a level costs 7 FMAs, one
                // subtraction and the v_max.
                const double mscale = cnt ? scale : 0.0;
                const double u_end = (PR ? (k0 + 1 < K) : (K > 1)) ? poly7(xc, m.xv[K - 1]) : u_hi;   // the ray's last crossing (flags below)
                double xm[PX];
#pragma unroll
                for (int n = 0; n < PX; ++n) xm[n] = xc[n] * mscale;
                double s_hi = u_hi * mscale;                                                   // scaled crossing of the previous level
                int k = 1;
                if constexpr (PR) {
                    // per-ray heights: levels up to the ray's own first one (k <= k0) count 0; level k0 + 1 starts from the
                    // ten-iteration crossing u_hi of level k0, as level 1 of a slice does
                    const double s_first = s_hi;
                    for (; k < K; ++k) {
                        const double st = poly7(xm, m.xv[k]);
                        const double L = st - (k == k0 + 1 ? s_first : s_hi);
                        s_hi = st;
                        atomicMax(&mxc[k * MXCOLS], (unsigned long long)__double_as_longlong(k > k0 ? fmax(fabs(L), 0.0) : 0.0));
                    }
                }
                for (; k + 2 <= K; k += 2) {
                    const double v0 = m.xv[k], v1 = m.xv[k + 1];
                    const double sa = poly7(xm, v0), sb = poly7(xm, v1);
                    const double La = sa - s_hi, Lb = sb - sa;
                    s_hi = sb;
                    atomicMax(&mxc[k * MXCOLS], (unsigned long long)__double_as_longlong(fmax(fabs(La), 0.0)));
                    atomicMax(&mxc[(k + 1) * MXCOLS], (unsigned long long)__double_as_longlong(fmax(fabs(Lb), 0.0)));
                }
                for (; k < K; ++k) {
                    const double st = poly7(xm, m.xv[k]);
                    const double L = st - s_hi;
                    s_hi = st;
                    atomicMax(&mxc[k * MXCOLS], (unsigned long long)__double_as_longlong(fmax(fabs(L), 0.0)));
                }
                if (PR ? (k0 + 1 < K) : (K > 1)) last_len = u_end * scale;                    // (only its NaN-ness is used)
                if (K > 0 && cnt && !(poly5(q.h, u_end) > c.z_hi)) my_flags |= 8;              // last sample of the ray
            }
            if (reduce && cnt && K > 0) my_flags |= (last_len != last_len) ? 1 : 2;
        }
    }
    flush();
}

// ---- pass 2: trapezoid integration of both fields along every ray (delay.py:285-323) -------------------------------
// SLOW as in crossings_kernel: <false> integrates the classified-fast rays with the light geodesy, <true> the rest
// with the generic one (and returns immediately when there are none).
// GRID (light kernel only), known at compile time so that the per-sample code carries no trace of the other variants:
//   1 (REGULAR): both horizontal axes are exactly uniform and the cube allows 32-bit offsets - the usual lat/lon or LCC model grid;
//   2 (TABLES): both axes only NEARLY uniform (e.g. 0.1-degree nodes stored as float32) - guess-and-verify against the LDS tables -
//               and 32-bit offsets;   0: whatever the run-time flags say.
// PR (light kernel only): per-ray origin heights (P.ht_ray) - its own instantiation, the slice kernels carry none of it.
template <typename T2, bool SLOW, int GRID = 0, bool PR = false>
// (f64 cubes - azimuth-time-grid blends - hold 8 x 16 B of corners per sample: three waves per SIMD without scratch beat four with 80 B of
// it, 6.8 against 11.4 ms per 16 M rays; the per-ray-height instantiation, 24 B of scratch at four waves, is better off as it is: 5.7 against 6.4 ms)
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(SLOW ? 1 : (sizeof(T2) == 16 ? 3 : 4), SLOW ? 8 : (sizeof(T2) == 16 ? 3 : 4)))) void march_kernel(CubeView<T2> c_in, RayParams P, LccParams proj) {
    static_assert(!SLOW || !PR, "the generic kernel looks at P.ht_ray at run time");
    if (SLOW && *P.nslow == 0) return;
    const bool per_ray = PR || (SLOW && P.ht_ray != nullptr);
    constexpr bool REGULAR = GRID == 1;
    CubeView<T2> c = c_in;
    if (REGULAR) { c.exact_y = 1; c.exact_x = 1; c.small = 1; }
    if (GRID == 2) { c.exact_y = 0; c.exact_x = 0; c.uni_y = 1; c.uni_x = 1; c.small = 1; }
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const RaySmem m = carve_smem(smem_raw, c.ny, c.nx, c.nz, c.exact_y, c.exact_x);
    // f64 cubes on exactly uniform axes (round 4): a wave's footprint of one model level - 4 x 4 cube columns x 3 z entries of 16 B -
    // staged in LDS, the eight corners of every sample of the level read from there (STAGED below)
    constexpr bool STAGED = !SLOW && !PR && REGULAR && sizeof(T2) == 16;
    constexpr int STAGE_N = 48;                        // [z 0..2][y 0..3][x 0..3]
    __shared__ T2 s_stage[STAGED ? (BLOCK / 64) * STAGE_N : 1];
    fill_axes(c, m);
    const int tid = threadIdx.x;
    int K = 0, slice = -1;
    double poison = 0.0;
    bool clamp_lo = false, clamp_hi = false, clamp_any = false;
    TileWalk walk(P.tile_count, P.tile_ctr, m.K + 2);
    int64_t lt;
    while (walk.next(P.tile_count, lt)) {
        const int64_t tg = P.tile_begin + lt;              // tile of the batch; t: tile within its slice
        const int sl = (int)(tg / P.tiles_per_slice);
        const int64_t t = tg - (int64_t)sl * P.tiles_per_slice;
        if (sl != slice) {                                 // (workgroup-uniform) the slice's level table and integration partition
            slice = sl;
            K = fill_levels(c.nz, m, P.hts ? P.hts[sl] : P.ht, P.zref);
            int tz = threadIdx.x;
            asm volatile("" : "+v"(tz));                   // (opaque: nothing derived from it is hoisted out of the tile loop and spilled)
            if (tz == 0) m.K[1] = 0;
            __syncthreads();
            for (int k = tz; k < K; k += BLOCK) {
                int np;
                if (P.nparts_override) np = P.nparts_override[(int64_t)sl * MAX_LEVELS + k];
                else {
                    const double parts = ceil(__longlong_as_double((long long)P.maxlen_bits[(int64_t)sl * MAX_LEVELS + k]) / P.max_seg) + 1.0;   // delay.py:283
                    np = (parts >= 1.0 && parts <= (double)MAX_NPARTS) ? (int)parts : -1;
                }
                if (np < 2 || np > MAX_NPARTS) {      // diverged lengths (e.g. look vectors far from unit length): refuse to loop over them
                    np = 2;
                    atomicOr(P.flags + sl, 16);       // RDR_FLAG_DIVERGED
                    m.K[1] = 1;
                }
                m.np[k] = np;
                m.step[k] = 1.0 / ((double)np - 1.0);                    // np.linspace(0,1,np) (delay.py:287)
                m.hs[k] = 0.5e-6 * m.step[k];                            // delay.py:314-315: end points get half of L*1e-6/(np-1)
                if constexpr (!SLOW && !PR) {                            // the slice loop's packed record of the level
                    const int kzk = m.kz[k], last = c.nz - 1;
                    const int zb = max(window2_base(c.nz, kzk), 0);      // (nz = 2: the window is never trusted, only in-range reads matter)
                    LevelRec r;
                    r.xv = m.xv[k]; r.hs = m.hs[k]; r.step = m.step[k];
                    r.zmid = m.ax.ez[min(zb + 1, last)].x; r.r0 = m.ax.ez[zb].y; r.r1 = m.ax.ez[min(zb + 1, last)].y;
                    r.gk = m.ax.ez[kzk].x; r.rk = m.ax.ez[kzk].y;
                    r.npkz = np | (kzk << 17); r.pad[0] = r.pad[1] = r.pad[2] = 0;
                    m.lev[k] = r;
                    // one record past the last level (K <= nz-1): the "next level" of the last one repeats its top abscissa with weight 0,
                    // so the loop computes du1 = X(v) - X(v) = 0 and adds 0 * 0 to the top weight instead of branching on `more`
                    if (k == K - 1) { r.hs = 0.0; r.npkz = 2 | (kzk << 17); m.lev[K] = r; }
                }
            }
            __syncthreads();
            const int flags_in = P.flags[sl];
            // Every output of the slice is NaN when its partition is undefined: diverged lengths, or a NaN ray length anywhere in
            // the slice - ndarray.max poisons nParts and the reference raises (delay.py:283).  The synchronous entry points raise
            // the same error; a caller of the asynchronous ones (device arrays, no host round trip) gets NaN, never a finite
            // delay computed with a partition the reference does not define.
            poison = (m.K[1] || (flags_in & (1 | 32))) ? qnan() : 0.0;      // (32: a per-ray height below the slice table's)
            clamp_lo = !(flags_in & 4);               // ALL first samples below zmin  (delay.py:306-307)
            clamp_hi = !(flags_in & 8);               // ALL last samples above zmax   (delay.py:310-311)
            clamp_any = clamp_lo | clamp_hi;
        }
        int tl = threadIdx.x;
        asm volatile("" : "+v"(tl));                       // per-tile opaque copy of the thread index (see crossings_kernel)
        int64_t i; bool active;
        if (P.origin_mode == 0) {
            const int64_t ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
            const int64_t row = ty * TILE + (tl >> 4), col = tx * TILE + (tl & 15);
            active = row < P.ny && col < P.nx;
            i = row * P.nx + col;
        } else {
            i = t * BLOCK + tl;
            active = i < P.n;
        }
        const double* w = P.ws + (lt * BLOCK + tl);
        const int64_t ns = P.nslots;
        const double scale_rec = w[(int64_t)WS_SCALE * ns];         // light ray: ray length per unit of u (> 0 or NaN); generic ray: 0
        const bool fast_ok = !active || scale_rec != 0.0;
        const bool mine = SLOW ? !fast_ok : fast_ok;
        if (SLOW && !__any(mine)) continue;
        double acc_w = 0.0, acc_h = 0.0;
        // per-ray heights: the ray's first level of the slice table, found exactly as pass 1 found it (idle lanes: none)
        int k0 = 0; double lo_first = K > 0 ? m.lo[0] : 0.0;
        if (PR || SLOW) {
            if (per_ray) k0 = (active && mine) ? first_level(m.ax.ez, c.nz, m.lo, m.hi, m.kz, K, P.ht_ray[i], lo_first) : K;
        }
        if constexpr (!SLOW) {
            // ---- light rays: every distinct sample point of the ray once, level by level --------------------------------
            // The sample schedule (level k, fraction j/(np-1)) is the same for every ray of the slice, so the loop counters
            // live in scalar registers.  A sample shared by two segments (top of k = bottom of k+1, losreader.py:811-812) is
            // evaluated once and carries both trapezoid end weights; a level is its interior samples (usually none or one)
            // plus its top sample.  Latency is hidden by the four waves per SIMD, not by batching samples inside a lane.
            RayPoly q;
            double xc[PX];
#pragma unroll
            for (int n = 0; n < PN; ++n) {
                q.h[n] = w[(int64_t)(WS_POLY_H + n) * ns];
                q.lat[n] = w[(int64_t)(WS_POLY_LAT + n) * ns];
                q.lon[n] = w[(int64_t)(WS_POLY_LON + n) * ns];
            }
            // The level crossings come from the ray's crossing polynomial as the loop reaches them (7 FMAs, no global loads
            // inside the loop besides the gathers, so the only memory waits are on a sample's own corners).
#pragma unroll
            for (int n = 0; n < PX; ++n) xc[n] = w[(int64_t)(WS_XPOLY + n) * ns];
            const double scale = scale_rec;
            // MODE 1: a level's top sample / the ray's first sample; MODE 2: a sample strictly inside its model interval
            // lanes whose samples count: a lane of tile padding or a generic ray (its record is not a polynomial) computes garbage that is
            // never stored, and must not decide a wave-uniform branch
            const unsigned long long live = __builtin_amdgcn_ballot_w64(active && mine);
            auto run = [&](auto nochk) {
            constexpr bool NC = decltype(nochk)::value;
            auto issue_top = [&](double us, int zbase, bool floor_it, bool ceil_it, PendingSample<T2>& s) {
                double ph = poly5(q.h, us);
                const double plat = poly5(q.lat, us), plon = poly5(q.lon, us);       // delay.py:295 through the ray polynomials
                // all-pixels z-clamp of the very first / very last sample (delay.py:306-311): when it applies every pixel is
                // below (above) the cube, so "set to zmin" == max(ph, zmin).  Slice-uniform conditions: real (scalar) branches,
                // so the two samples of a ray they can apply to are the only ones that pay for them.
                if (floor_it) { asm volatile("" ::: "memory"); ph = fmax(ph, c.z_lo); }
                if (ceil_it) { asm volatile("" ::: "memory"); ph = fmin(ph, c.z_hi); }
                sample_issue<T2, 1, NC>(c, m.ax, plat, plon, ph, zbase, s);              // delay.py:298,319
            };
            auto issue_mid = [&](double us, int kz_, PendingSample<T2>& s) {
                const double ph = poly5(q.h, us);
                const double plat = poly5(q.lat, us), plon = poly5(q.lon, us);
                sample_issue<T2, 2, NC>(c, m.ax, plat, plon, ph, kz_, s);
            };
            auto finish = [&](const PendingSample<T2>& s, double wv) {
                double vw, vh;
                sample_finish_lerp(s, vw, vh);
                acc_w = fma(wv, vw, acc_w); acc_h = fma(wv, vh, acc_h);              // delay.py:323
            };
            if constexpr (PR) {
                // Per-ray heights: the level schedule (k, j) stays slice-uniform - scalar loop counters, the partition of the slice - and a
                // lane joins it at its own first level k0 with its own first sample; until then it idles (execution mask).  The loop
                // starts at the wave's lowest k0.  Same arithmetic per sample as the slice loop below, so that equal heights give the
                // slice kernel's delays bit for bit.
                int kmin = k0;
#pragma unroll
                for (int off = 32; off > 0; off >>= 1) kmin = min(kmin, __shfl_xor(kmin, off, 64));
                kmin = __builtin_amdgcn_readfirstlane(kmin);
                // every lane's FIRST sample (the bottom of its own first level) in one evaluation for the whole wave - per-lane level
                // constants from LDS - instead of once per distinct k0 inside the loop (a DEM tile has a dozen of those)
                const bool has = k0 < K;
                const int kf = has ? k0 : 0;
                double u_k = has ? w[(int64_t)WS_U0 * ns] : 0.0, u_last = has ? w[(int64_t)WS_U1 * ns] : 0.0;
                double du = u_last - u_k;
                if (K > 0) {
                    const int kzf = m.kz[kf];
                    PendingSample<T2> s;
                    issue_top(fma(0.0 * m.step[kf], du, u_k), window2_base(c.nz, kzf - ((lo_first <= m.ax.ez[kzf].x) ? 1 : 0)), clamp_lo, false, s);
                    if (has) finish(s, m.hs[kf] * fabs(du));
                }
#pragma unroll 1
                for (int k = kmin; k < K; ++k) {
                    const int np = __builtin_amdgcn_readfirstlane(m.np[k]);
                    const int kz = __builtin_amdgcn_readfirstlane(m.kz[k]);
                    const double step = m.step[k], hs = m.hs[k];
                    const bool more = k + 1 < K;
                    if (k >= k0) {
                        const int zbase = window2_base(c.nz, kz);
                        const double w_mid = (2.0 * hs) * fabs(du);
#pragma unroll 1
                        for (int j = 1; j < np - 1; ++j) {
                            PendingSample<T2> s;
                            issue_mid(fma((double)j * step, du, u_k), kz, s);
                            finish(s, w_mid);
                        }
                        PendingSample<T2> top;
                        issue_top(u_k + du, zbase, false, clamp_hi && !more, top);
                        double du1 = 0.0;
                        double w_top = hs * fabs(du);
                        if (more) {
                            const double t2 = poly7(xc, m.xv[k + 1]);
                            du1 = t2 - u_last; u_last = t2;
                            w_top = fma(m.hs[k + 1], fabs(du1), w_top);
                        }
                        finish(top, w_top);
                        u_k += du; du = du1;
                    }
                }
            } else {
            // Everything slice-uniform about a level comes from its packed LDS record through ONE address register (la + fixed offsets).
            typedef __attribute__((address_space(3))) const LevelRec LdsRec;
            int la = (int)(size_t)m.lev;                                             // LDS byte address of record 0 (a 32-bit LDS pointer)
            asm volatile("" : "+v"(la));                                             // (a VGPR once, not a v_mov per read)
            const LdsRec* rec = (const LdsRec*)(size_t)(unsigned)la;
            int npkz = __builtin_amdgcn_readfirstlane(rec->npkz);
            int np = npkz & 0x1ffff, kz = npkz >> 17;                                // np >= 2 (fill above)
            double hs = rec->hs;
            double u_k = w[(int64_t)WS_U0 * ns];
            double u_last = w[(int64_t)WS_U1 * ns];
            double du = u_last - u_k;
            // the ray's very first sample is the BOTTOM of its segment: when that is a model node (origin at or below it), the
            // two-entry z window must start one interval lower
            if (K > 0) {
                PendingSample<T2> s;
                issue_top(fma(0.0 * rec->step, du, u_k), window2_base(c.nz, kz - ((m.lo[0] <= m.ax.ez[kz].x) ? 1 : 0)), clamp_lo, false, s);
                finish(s, hs * fabs(du));
            }
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                const int zbase = window2_base(c.nz, kz);                            // first table entry of the two-entry z window
                const bool more = k + 1 < K;
                // trapezoid weights per unit of u (delay.py:314-315): interior samples, and the top one with both its segments.  The weight is the
                // segment's LENGTH (a norm, losreader.py:821): |du| - an origin above zref in zref's model interval walks its one segment downwards
                const double w_mid = (2.0 * hs) * fabs(du);
                if (np > 2) {
                    const double step = rec->step, gk = rec->gk, rk = rec->rk;
#pragma unroll 1
                    for (int j = 1; j < np - 1; ++j) {                               // low + frac * (high - low), delay.py:292
                        PendingSample<T2> s;
                        const double us = fma((double)j * step, du, u_k);
                        const double ph = poly5(q.h, us), plat = poly5(q.lat, us), plon = poly5(q.lon, us);
                        int iz = kz;                                                 // strictly inside model interval kz: one table entry, no select
                        s.tz = (ph - gk) * rk;
                        if (!(s.tz >= 0.0) || !(s.tz <= 1.0)) cell_exact(m.ax.ez, c.nz, ph, iz, s.tz);   // rare
                        gather_corners<T2, true, NC>(c, m.ax, plat, plon, iz, s);
                        finish(s, w_mid);
                    }
                }
                // top sample: its gathers are issued first, the next level's crossing (7 FMAs that need no memory) is computed
                // while they are in flight, then the sample is finished with the weight of both its segments
                PendingSample<T2> top;
                {
                    const double us = u_k + du;                                      // fraction exactly 1.0, as np.linspace returns it
                    double ph = poly5(q.h, us);
                    const double plat = poly5(q.lat, us), plon = poly5(q.lon, us);   // delay.py:295 through the ray polynomials
                    if (clamp_hi && !more) { asm volatile("" ::: "memory"); ph = fmin(ph, c.z_hi); }      // delay.py:310-311 (slice-uniform branch)
                    // z cell: the top sits on node g[zbase+1] within the residual of the three-iteration crossing - ABOVE it for every ray
                    // whose zenith angle falls with height (profiles/r06_nodetop_ab.txt).  When no live lane of the wave is below the node
                    // (one compare + a scalar test) the cell index is the scalar zbase+1 and the weight one multiplication; otherwise the
                    // per-lane two-entry window.  Same values either way.
                    const double d = ph - rec->zmid;
                    int iz; bool ok;
                    if (__builtin_expect((__builtin_amdgcn_ballot_w64(!(d >= 0.0)) & live) == 0ULL, 1)) {
                        iz = zbase + 1;
                        top.tz = d * rec->r1;
                        ok = top.tz <= 1.0;                                                                // (idle lanes: NaN)
                    } else {
                        double ds = d;
                        asm volatile("" : "+v"(ds));                                                       // (nothing of this block is hoisted into the fast path)
                        const bool up = ds >= 0.0;
                        iz = zbase + (int)up;
                        top.tz = fma(ds, up ? rec->r1 : rec->r0, up ? 0.0 : 1.0);
                        ok = (top.tz >= 0.0) & (top.tz <= 1.0);
                    }
                    if (!(ok & (c.nz >= 4))) cell_exact(m.ax.ez, c.nz, ph, iz, top.tz);                  // rare
                    gather_corners<T2, true, NC>(c, m.ax, plat, plon, iz, top);                          // delay.py:298,319
                }
                // (the record after the last level: same abscissa, weight 0 - see the fill above)
                const double t2 = poly7(xc, rec[1].xv);
                const double du1 = t2 - u_last, hs1 = rec[1].hs;
                u_last = t2;
                const double w_top = fma(hs1, fabs(du1), hs * fabs(du));
                npkz = __builtin_amdgcn_readfirstlane(rec[1].npkz);
                finish(top, w_top);
                u_k += du; du = du1; hs = hs1;
                np = npkz & 0x1ffff; kz = npkz >> 17;
                ++rec;
            }
            }   // (slice loop)
            };
            // Bounds of the horizontal cell search, decided ONCE per wave from the polynomial coefficients.  Every sample's ray
            // parameter lies between the ray's crossings: the two of the first level (u0, u1 of the record) and X(v_k), |v_k| <= 1,
            // whose magnitude is at most sum |x_k|.  So when |u0|, |u1| and sum |x_k| are all <= 1.001 (the crossings sit inside
            // the fit range [-1, 1] by construction; diverged or non-finite rays fail this), every sample has |u| <= 1.001 and
            // p(u) lies within c0 +- 1.006 sum_{k>=1} |c_k| (1.001^5 < 1.006).  When that interval is inside [0, n-1) for the
            // index-space lat AND lon polynomials of every ray of the wave, no gather can leave the cube and the per-sample bounds
            // tests (4 compares + the rare-path plumbing) are compiled out of the loop.  Lanes of tile padding get a harmless
            // in-range polynomial; a wave holding a generic ray (whose record is not a polynomial), a diverged ray or a ray near
            // the cube's edge keeps the checked loop.
            bool lane_safe = false;
            if (REGULAR) {
                double u0r = w[(int64_t)WS_U0 * ns], u1r = w[(int64_t)WS_U1 * ns];
                if (!active) {
#pragma unroll
                    for (int n = 0; n < PN; ++n) { q.lat[n] = 0.0; q.lon[n] = 0.0; }
                    q.lat[0] = 0.5; q.lon[0] = 0.5;
#pragma unroll
                    for (int n = 0; n < PX; ++n) xc[n] = 0.0;
                    u0r = 0.0; u1r = 0.0;
                }
                auto inside = [&](const double* cf, int n) {
                    const double r = 1.006 * (fabs(cf[1]) + fabs(cf[2]) + fabs(cf[3]) + fabs(cf[4]) + fabs(cf[5]));
                    return (cf[0] - r >= 0.0) & (cf[0] + r < (double)(n - 1));
                };
                double xs = 0.0;
#pragma unroll
                for (int n = 0; n < PX; ++n) xs += fabs(xc[n]);
                lane_safe = (mine || !active) && (fabs(u0r) <= 1.001) && (fabs(u1r) <= 1.001) && (xs <= 1.001) && inside(q.lat, c.ny) && inside(q.lon, c.nx);
            }
            // ---- f64 cubes: the level's footprint through LDS --------------------------------------------------------------------
            // An f64 cube doubles the bytes of every gather (4 x 32 B per lane and sample): the instantiation is bound by the vector L1's
            // 64 B/clk return path, not by instruction issue (6.8 against 5.0 ms per 16 M rays, round 3).  But the 64 rays of a wave
            // (4 x 16 neighbouring pixels) sit in one or two cube cells at any given level: per level the wave loads the 4 x 4 columns
            // around lane 0's cell, z entries zb .. zb+2 (48 lanes, ONE 16 B load each), into LDS and every sample of the level reads
            // its eight corners from there (128 B/clk, broadcast) - same operands, same arithmetic: the same bits.  A sample whose
            // cell leaves the staged block for ANY lane of the wave (coarse scenes, large look-angle gradients) takes the direct
            // gathers, wave-uniformly.  Only for waves with the no-check proof above (cells inside the cube by construction).
            auto run_staged = [&]() {
                T2* const st = s_stage + (tl >> 6) * STAGE_N;
                const int lane = tl & 63;
                unsigned lane_off = ((unsigned)((lane >> 2) & 3) * (unsigned)c.nx + (unsigned)(lane & 3)) * (unsigned)c.nz + (unsigned)(lane >> 4);
                asm volatile("" : "+v"(lane_off));        // (opaque: the block's base below stays ONE scalar product, not two vector multiplies per level)
                const bool loader = lane < STAGE_N;
                bool stage_on = true;                     // wave-uniform: off for the rest of the tile once a level's cells leave the block
                                                          // (the footprint is set by scene spacing / cube spacing: a coarse scene never fits)
                const char* const vb = reinterpret_cast<const char*>(c.v);
                const size_t rowx = (size_t)c.nz * sizeof(T2), rowy = (size_t)c.nx * rowx;
                int fy0 = 0, fx0 = 0, zb = 0;
                // the eight corners of the sample in cell (iy, ix, iz): from the staged block when every lane's cell is in it
                auto fetch = [&](int iy, int ix, int iz, PendingSample<T2>& sm) {
                    const unsigned ry = (unsigned)(iy - fy0), rx = (unsigned)(ix - fx0), rz = (unsigned)(iz - zb);
                    bool staged_here = false;
                    if (stage_on) {
                        const bool out = active & (max(max(ry, rx), rz + 1u) > 2u);
                        staged_here = __builtin_amdgcn_ballot_w64(out) == 0;
                        stage_on = staged_here;
                    }
                    if (staged_here) {
                        // (an LDS-address-space pointer, and a fence: left as a generic pointer the two branches' loads are merged into
                        // ONE set of flat loads behind a pointer select - LDS data through the flat path, slower than the gathers)
                        typedef decltype(T2().x) S1;
                        typedef S1 S2 __attribute__((ext_vector_type(2)));
                        typedef __attribute__((address_space(3))) S2 LdsS2;
                        const LdsS2* p = (const LdsS2*)st + (active ? (rz * 16u + ry * 4u + rx) : 0u);
                        auto get = [&](int i_, int o_) { const S2 t_ = p[o_]; sm.v[i_].x = t_.x; sm.v[i_].y = t_.y; };
                        get(0, 0); get(1, 16); get(2, 1); get(3, 17); get(4, 4); get(5, 20); get(6, 5); get(7, 21);
                        asm volatile("" ::: "memory");
                    } else {
                        const unsigned off = (__umul24(__umul24((unsigned)iy, (unsigned)c.nx) + (unsigned)ix, (unsigned)c.nz) + (unsigned)iz) * (unsigned)sizeof(T2);
                        const T2* p00 = reinterpret_cast<const T2*>(vb + off);
                        const T2* p01 = reinterpret_cast<const T2*>(vb + rowx + off);
                        const T2* p10 = reinterpret_cast<const T2*>(vb + rowy + off);
                        const T2* p11 = reinterpret_cast<const T2*>(vb + rowy + rowx + off);
                        sm.v[0] = p00[0]; sm.v[1] = p00[1]; sm.v[2] = p01[0]; sm.v[3] = p01[1];
                        sm.v[4] = p10[0]; sm.v[5] = p10[1]; sm.v[6] = p11[0]; sm.v[7] = p11[1];
                    }
                };
                auto finish = [&](const PendingSample<T2>& sm, double wv) {
                    double vw, vh;
                    sample_finish_lerp(sm, vw, vh);
                    acc_w = fma(wv, vw, acc_w); acc_h = fma(wv, vh, acc_h);
                };
                int np = __builtin_amdgcn_readfirstlane(m.np[0]);
                int kz = __builtin_amdgcn_readfirstlane(m.kz[0]);
                double step = m.step[0], hs = m.hs[0];
                double u_k = w[(int64_t)WS_U0 * ns];
                double u_last = w[(int64_t)WS_U1 * ns];
                double du = u_last - u_k;
                if (K > 0) {                              // the ray's very first sample: direct gathers (its z window may start one interval lower)
                    PendingSample<T2> s0;
                    double ph = poly5(q.h, fma(0.0 * step, du, u_k));
                    const double plat = poly5(q.lat, fma(0.0 * step, du, u_k)), plon = poly5(q.lon, fma(0.0 * step, du, u_k));
                    if (clamp_lo) { asm volatile("" ::: "memory"); ph = fmax(ph, c.z_lo); }
                    sample_issue<T2, 1, true>(c, m.ax, plat, plon, ph, window2_base(c.nz, kz - ((m.lo[0] <= m.ax.ez[kz].x) ? 1 : 0)), s0);
                    finish(s0, hs * fabs(du));
                }
#pragma unroll 1
                for (int k = 0; k < K; ++k) {
                    zb = window2_base(c.nz, kz);
                    const bool more = k + 1 < K;
                    const double w_mid = (2.0 * hs) * fabs(du);
                    // the level's TOP sample first: its cell places the staged block
                    PendingSample<T2> top;
                    double ph = poly5(q.h, u_k + du);
                    const double plat = poly5(q.lat, u_k + du), plon = poly5(q.lon, u_k + du);
                    if (clamp_hi && !more) { asm volatile("" ::: "memory"); ph = fmin(ph, c.z_hi); }
                    int iy, ix, iz;
                    cell_xy<true, true>(m.ax.ey, c.ny, plat, c.y_lo, c.y_hi, c.inv_dy, true, true, iy, top.ty);
                    cell_xy<true, true>(m.ax.ex, c.nx, plon, c.x_lo, c.x_hi, c.inv_dx, true, true, ix, top.tx);
                    window2_cell(m.ax.ez, c.nz, ph, zb, c.nz >= 4, iz, top.tz);
                    if (stage_on) {
                        fy0 = min(max(__builtin_amdgcn_readfirstlane(iy) - 1, 0), c.ny - 4);
                        fx0 = min(max(__builtin_amdgcn_readfirstlane(ix) - 1, 0), c.nx - 4);
                        const unsigned base = (unsigned)__builtin_amdgcn_readfirstlane((fy0 * c.nx + fx0) * c.nz + zb);
                        __builtin_amdgcn_wave_barrier();          // (the previous level's readers are done: same wave, program order)
                        if (loader) st[lane] = c.v[base + lane_off];
                        __builtin_amdgcn_wave_barrier();
                    }
#pragma unroll 1
                    for (int j = 1; j < np - 1; ++j) {
                        PendingSample<T2> sm;
                        const double us = fma((double)j * step, du, u_k);
                        const double mh = poly5(q.h, us), mlat = poly5(q.lat, us), mlon = poly5(q.lon, us);
                        int my, mx, mz = kz;
                        cell_xy<true, true>(m.ax.ey, c.ny, mlat, c.y_lo, c.y_hi, c.inv_dy, true, true, my, sm.ty);
                        cell_xy<true, true>(m.ax.ex, c.nx, mlon, c.x_lo, c.x_hi, c.inv_dx, true, true, mx, sm.tx);
                        const double2 e0 = m.ax.ez[kz];
                        sm.tz = (mh - e0.x) * e0.y;
                        if (!(sm.tz >= 0.0) || !(sm.tz <= 1.0)) cell_exact(m.ax.ez, c.nz, mh, mz, sm.tz);   // rare
                        fetch(my, mx, mz, sm);
                        finish(sm, w_mid);
                    }
                    fetch(iy, ix, iz, top);
                    double du1 = 0.0, hs1 = 0.0;
                    double w_top = hs * fabs(du);
                    if (more) {
                        const double t2 = poly7(xc, m.xv[k + 1]);
                        du1 = t2 - u_last; u_last = t2; hs1 = m.hs[k + 1];
                        w_top = fma(hs1, fabs(du1), w_top);
                    }
                    finish(top, w_top);
                    u_k += du; du = du1; hs = hs1;
                    if (more) {
                        np = __builtin_amdgcn_readfirstlane(m.np[k + 1]);
                        kz = __builtin_amdgcn_readfirstlane(m.kz[k + 1]);
                        step = m.step[k + 1];
                    }
                }
            };
            if (REGULAR && __all(lane_safe)) {
                if constexpr (STAGED) {
                    if (P.stage_f64 && c.ny >= 4 && c.nx >= 4 && c.nz >= 3) run_staged();      // (the staged block holds z entries zb .. zb+2, zb >= 0)
                    else run(std::integral_constant<bool, true>{});
                } else run(std::integral_constant<bool, true>{});
            }
            else run(std::integral_constant<bool, false>{});
            acc_w *= scale; acc_h *= scale;
        } else {
            double vw_top = 0.0, vh_top = 0.0;    // sample values at the top of the previous segment (= bottom of this one)
            // generic rays: origin / look vector / origin frame; light rays: the three polynomials
            const double ox = w[(int64_t)(WS_ORIGIN + 0) * ns], oy = w[(int64_t)(WS_ORIGIN + 1) * ns], oz = w[(int64_t)(WS_ORIGIN + 2) * ns];
            const double lx = w[(int64_t)(WS_LOS + 0) * ns], ly = w[(int64_t)(WS_LOS + 1) * ns], lz = w[(int64_t)(WS_LOS + 2) * ns];
            const double nl2 = fma(lx, lx, fma(ly, ly, lz * lz));
            const double scale = nl2 * rsq_nr<2>(nl2);                  // |l|: ray length per unit of t
            // level crossings: streamed from the ray's side-buffer column one level ahead of their use, or - when pass 1 found the
            // side buffer full - recomputed with the very same iteration (same inputs, same instruction sequence)
            const int64_t sidx = mine ? (int64_t)w[(int64_t)WS_SIDE * ns] : -1;
            const double* const sd = (sidx >= 0 && P.side) ? P.side + sidx : nullptr;
            double t_hi = 0.0, t_next = 0.0, inv_cosf = 1.0;
            if (sd) { t_hi = sd[0]; if (k0 < K) t_next = sd[(int64_t)(k0 + 1) * P.side_cap]; }
            else if (mine && k0 < K) t_hi = toa_newton_t(ox, oy, oz, lx, ly, lz, lo_first, 10, 1.0);
    #pragma unroll 1
            for (int k = 0; k < K; ++k) {
                if (k < k0) continue;                // (per-ray heights: the ray joins the slice's level table at k0; else k0 = 0)
                const double t_lo = t_hi;
                if (sd) {
                    t_hi = t_next;
                    if (k + 2 <= K) t_next = sd[(int64_t)(k + 2) * P.side_cap];
                } else if (mine) {
                    t_hi = toa_newton_t(ox, oy, oz, lx, ly, lz, m.hi[k], k == k0 ? 10 : 3, inv_cosf);
                    if (k == k0) inv_cosf = (fabs(t_hi - t_lo) * scale) / (m.hi[k0] - lo_first);   // losreader.py:824-825, as in pass 1
                }
                const double dt = t_hi - t_lo;
                const int np = m.np[k];
                const double step = m.step[k];
                const double segw = (fabs(dt) * scale * 1.0e-6) * step;    // delay.py:315: L*1e-6/(np-1), L = |high-low| (losreader.py:821)
                const double dts = dt * step;                        // sample spacing: low + (j*step)*(high-low), delay.py:292
                const int kz = m.kz[k];
                // j = 0 of this segment is the SAME point as j = np-1 of the previous one (low_xyz is high_xyz,
                // losreader.py:811-812): its interpolated value is reused instead of recomputed (the reference evaluates
                // it twice and gets the same number both times).  The order of accumulation is unchanged.
                if (k > k0) { acc_w = fma(0.5 * segw, vw_top, acc_w); acc_h = fma(0.5 * segw, vh_top, acc_h); }
    #pragma unroll 1
                for (int j = (k == k0 ? 0 : 1); j < np; ++j) {
                    const double ts = fma((double)j, dts, t_lo);
                    double plon, plat, ph;
                    ecef2lla(fma(ts, lx, ox), fma(ts, ly, oy), fma(ts, lz, oz), plon, plat, ph);                     // delay.py:295
                    if (proj.kind == 1) { double px_, py_; lcc_forward(proj, plat, plon, px_, py_); plon = px_; plat = py_; }   // ecef_to_model, delay.py:253
                    // all-pixels z-clamp of the very first / very last sample (delay.py:306-311): when it applies every
                    // pixel is below (above) the cube, so "set to zmin" == max(ph, zmin); the bounds are wave-uniform
                    if (clamp_any) {
                        const double zfloor = (clamp_lo && k == k0 && j == 0) ? c.z_lo : -__builtin_huge_val();
                        const double zceil = (clamp_hi && k == K - 1 && j == np - 1) ? c.z_hi : __builtin_huge_val();
                        ph = fmin(fmax(ph, zfloor), zceil);
                    }
                    double vw, vh;
                    sample_cube(c, m.ax, plat, plon, ph, kz, vw, vh);                   // delay.py:298,319
                    const double wt = ((j == 0) | (j == np - 1)) ? 0.5 * segw : segw;     // delay.py:314-315
                    acc_w = fma(wt, vw, acc_w); acc_h = fma(wt, vh, acc_h);               // delay.py:323
                    vw_top = vw; vh_top = vh;                                             // after the loop: value at j = np-1
                }
            }
        }
        if (active && mine) { const int64_t o = (int64_t)sl * P.n + i; P.wet[o] = acc_w + poison; P.hyd[o] = acc_h + poison; }
    }
}

}  // namespace rdr
