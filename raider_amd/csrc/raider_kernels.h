// Device-side building blocks of libraider_hip (gfx950 / CDNA4 only).
//
//   CubeView + trilinear()  : scipy RegularGridInterpolator(linear, fill=nan) on the interleaved
//                             (wet,hydro) cube          [delayFcns.py:55-56, scipy _rgi.py:405-499]
//   toa_newton()            : getTopOfAtmosphere        [losreader.py:706-733]
//   ray_kernel<MODE>        : build_ray fused with the per-level trapezoid of _build_cube_ray
//                             [losreader.py:772-835, delay.py:283-323]; MODE 0 = pass 1 (per-level
//                             batch max of ray length + clamp/NaN flags), MODE 1 = pass 2 (integrate)
//
// Design notes (MI355X):
//   * one ray per lane, 64-lane wavefronts, 256-thread workgroups = one 16x16 pixel tile of the
//     scene: neighbouring rays walk the same few cube columns, so their gathers hit the same
//     L1/L2 lines; the per-level loop bounds are batch-uniform, so a wave never diverges.
//   * the cube is stored (y,x,z) with z fastest and (wet,hydro) interleaved per cell: the two z
//     neighbours of both fields of one column are ONE contiguous 16 B (f32) / 32 B (f64) read.
//   * grid axes + the level table + the per-level partition live in LDS (a few KB), filled once
//     per workgroup; nothing per-level is ever materialised in HBM (the reference materialises
//     K x N x 56 B).
//   * persistent grid (a few workgroups per CU) walking tiles with an XCD-aware mapping so that the 32
//     CUs sharing one L2 work on one contiguous band of the scene.
//   * no MFMA: this is a gather + transcendental-heavy fp64 integrate, there is no contraction.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "geodesy.h"
#include "geodesy_fast.h"

namespace rdr {

constexpr int TILE = 16;         // 16x16 pixel tile per workgroup (GRID mode)
constexpr int BLOCK = TILE * TILE;
constexpr int MAX_LEVELS = 512;  // model intervals (ERA5 has 144)

template <typename T2>
struct CubeView {
    const T2* v;          // [(iy*nx+ix)*nz+iz] -> (wet, hydro)
    const double* axes;   // ys[ny] | xs[nx] | zs[nz]   (ascending)
    int ny, nx, nz;
    double y_lo, y_hi, x_lo, x_hi, z_lo, z_hi;   // axis end points (bounds test)
    double inv_dy, inv_dx, inv_dz;               // (n-1)/(g[n-1]-g[0]) for the uniform-axis index guess
    int uni_y, uni_x, uni_z;                     // axis is (nearly) uniform: guess is within +-1 cell
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

__device__ __forceinline__ int find_cell_hint(const double* g, int n, double x, int hint) {
    int i = min(max(hint, 0), n - 2);
    while (i > 0 && x < g[i]) --i;
    while (i < n - 2 && x >= g[i + 1]) ++i;
    return i;
}

__device__ __forceinline__ void ld2(const float2* p, double& a, double& b) { const float2 t = *p; a = (double)t.x; b = (double)t.y; }
__device__ __forceinline__ void ld2(const double2* p, double& a, double& b) { const double2 t = *p; a = t.x; b = t.y; }

// scipy linear RGI on both fields.  sy/sx/sz: the axes (LDS or global).  zhint >= 0: start the z search
// at that interval (ray marcher knows the model interval), else use the uniform guess / bisection.
// RECIP: cell-width reciprocals follow the axes in the table (sy[ny+nx+nz + i]); the ray kernel uses them to turn
// the three divisions per sample into multiplications (1 ulp difference in the weights).
template <typename T2, bool RECIP = false>
__device__ __forceinline__ void trilinear(const CubeView<T2>& c, const double* sy, const double* sx, const double* sz,
                                          double y, double x, double z, int zhint, double& wet, double& hyd) {
    // out of bounds (x < g[0] or x > g[-1]) -> fill_value nan; nan coordinate -> nan   (_rgi.py:437-442,585-592)
    const bool inside = (y >= c.y_lo) && (y <= c.y_hi) && (x >= c.x_lo) && (x <= c.x_hi) && (z >= c.z_lo) && (z <= c.z_hi);
    if (!inside) { wet = qnan(); hyd = qnan(); return; }
    const int iy = find_cell(sy, c.ny, y, c.y_lo, c.inv_dy, c.uni_y);
    const int ix = find_cell(sx, c.nx, x, c.x_lo, c.inv_dx, c.uni_x);
    const int iz = zhint >= 0 ? find_cell_hint(sz, c.nz, z, zhint) : find_cell(sz, c.nz, z, c.z_lo, c.inv_dz, c.uni_z);
    double ty, tx, tz;
    if (RECIP) {
        const int na = c.ny + c.nx + c.nz;
        ty = (y - sy[iy]) * sy[na + iy];
        tx = (x - sx[ix]) * sx[na + ix];
        tz = (z - sz[iz]) * sz[na + iz];
    } else {
        ty = (y - sy[iy]) / (sy[iy + 1] - sy[iy]);
        tx = (x - sx[ix]) / (sx[ix + 1] - sx[ix]);
        tz = (z - sz[iz]) / (sz[iz + 1] - sz[iz]);
    }
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

// ---- ray-kernel sampler ----------------------------------------------------------------------------------------
// Same scipy semantics as trilinear<> (interval g[i] <= v < g[i+1], last cell closed; outside / NaN -> NaN).
// Cell search = ONE LDS round trip per axis: the axis table is stored as (g[i], 1/(g[i+1]-g[i])) pairs, a guess i0
// is made (linear guess on (nearly) uniform axes, the segment's model interval along z), the three entries
// i0-1, i0, i0+1 are fetched together and the right one is selected in registers.  A lane whose guess is more
// than one cell off (non-uniform axis, non-converged crossing in sub-metre levels, < 4-node axes) bisects instead.
__device__ __forceinline__ int bisect_cell2(const double2* e, int n, double v) {
    int lo = 0, hi = n;   // first index with v < g[idx]
    while (lo < hi) {
        const int mid = (lo + hi) >> 1;
        if (v < e[mid].x) hi = mid; else lo = mid + 1;
    }
    return min(max(lo - 1, 0), n - 2);
}

__device__ __forceinline__ void window_cell(const double2* e, int n, double v, int guess, bool trust, int& i, double& t) {
    const int i0 = min(max(guess, 1), n - 3);
    const double2 em = e[i0 - 1], e0 = e[i0], e1 = e[i0 + 1];
    const bool dn = v < e0.x, up = v >= e1.x;
    i = i0 - (int)dn + (int)up;
    const double g = dn ? em.x : (up ? e1.x : e0.x);
    const double r = dn ? em.y : (up ? e1.y : e0.y);
    t = (v - g) * r;
    if (!trust || !(t >= 0.0) || !(t <= 1.0)) {      // rare: exact search
        i = bisect_cell2(e, n, v);
        t = (v - e[i].x) * e[i].y;
    }
}

// (nearly) uniform x / y axes: the linear guess is right unless the point sits within round-off of a node (or the
// axis deviates from uniform): read the guessed cell and the next node, verify, else search exactly.
__device__ __forceinline__ void guess_cell(const double2* e, int n, double v, int guess, bool trust, int& i, double& t) {
    i = min(max(guess, 0), n - 2);
    const double2 e0 = e[i];
    const double g1 = e[i + 1].x;
    t = (v - e0.x) * e0.y;
    const bool ok = trust & (v >= e0.x) & ((v < g1) | (i == n - 2));
    if (!ok) {                                        // rare: exact search
        i = bisect_cell2(e, n, v);
        t = (v - e[i].x) * e[i].y;
    }
}

// tab2 = (g, 1/dg) pairs of [ys | xs | zs] in LDS.
template <typename T2>
__device__ __forceinline__ void sample_cube(const CubeView<T2>& c, const double2* tab2, double y, double x, double z, int kz,
                                            double& wet, double& hyd) {
    const bool inside = (y >= c.y_lo) && (y <= c.y_hi) && (x >= c.x_lo) && (x <= c.x_hi) && (z >= c.z_lo) && (z <= c.z_hi);
    if (!inside) { wet = qnan(); hyd = qnan(); return; }
    const double2* ey = tab2; const double2* ex = tab2 + c.ny; const double2* ez = ex + c.nx;
    int iy, ix, iz; double ty, tx, tz;
    guess_cell(ey, c.ny, y, (int)((y - c.y_lo) * c.inv_dy), c.uni_y, iy, ty);
    guess_cell(ex, c.nx, x, (int)((x - c.x_lo) * c.inv_dx), c.uni_x, ix, tx);
    window_cell(ez, c.nz, z, kz, c.nz >= 4, iz, tz);
    const T2* p00 = c.v + ((int64_t)iy * c.nx + ix) * c.nz + iz;
    const T2* p01 = p00 + c.nz;
    const T2* p10 = p00 + (int64_t)c.nx * c.nz;
    const T2* p11 = p10 + c.nz;
    double w[8], h[8];
    ld2(p00, w[0], h[0]); ld2(p00 + 1, w[1], h[1]);
    ld2(p01, w[2], h[2]); ld2(p01 + 1, w[3], h[3]);
    ld2(p10, w[4], h[4]); ld2(p10 + 1, w[5], h[5]);
    ld2(p11, w[6], h[6]); ld2(p11 + 1, w[7], h[7]);
    const double wy0 = 1.0 - ty, wx0 = 1.0 - tx, wz0 = 1.0 - tz;
    const double a00 = wy0 * wx0, a01 = wy0 * tx, a10 = ty * wx0, a11 = ty * tx;
    double sw = 0.0, sh = 0.0, k;      // weight = (wy*wx)*wz, corners in (y,x,z) lexicographic order (_rgi.py:490-498)
    k = a00 * wz0; sw = fma(w[0], k, sw); sh = fma(h[0], k, sh);
    k = a00 * tz;  sw = fma(w[1], k, sw); sh = fma(h[1], k, sh);
    k = a01 * wz0; sw = fma(w[2], k, sw); sh = fma(h[2], k, sh);
    k = a01 * tz;  sw = fma(w[3], k, sw); sh = fma(h[3], k, sh);
    k = a10 * wz0; sw = fma(w[4], k, sw); sh = fma(h[4], k, sh);
    k = a10 * tz;  sw = fma(w[5], k, sw); sh = fma(h[5], k, sh);
    k = a11 * wz0; sw = fma(w[6], k, sw); sh = fma(h[6], k, sh);
    k = a11 * tz;  sw = fma(w[7], k, sw); sh = fma(h[7], k, sh);
    wet = sw; hyd = sh;
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

// Ray kernels: the same iteration carried on the scalar ray parameter t (pos = o + t*l): t0 = h,
// t += (h - height(o + t l)) / factor - identical in exact arithmetic to losreader.py:724-731, 2 live doubles per
// crossing instead of 6 - with the light-fp64 TRUE height (geodesy_fast.h): crossings land within 2e-5 m (at 40 km)
// of the reference's, which moves the delays by < 1e-10 m.  SLOW = the generic PROJ-formula height, used by the
// *_kernel<T2, true> instantiations that mop up the rare rays the static classification rejects.
template <bool SLOW>
__device__ __forceinline__ double height_sel(double x, double y, double z) {
    return SLOW ? ecef_height(x, y, z) : height_fast_nocheck(x, y, z);
}

template <bool SLOW>
__device__ __forceinline__ double toa_newton_t(double ox, double oy, double oz, double lx, double ly, double lz,
                                               double h, int iters, double inv_factor) {
    double t = h;
    // light path: the early iterates only steer the last ones (the iteration contracts strongly), so they use the
    // 22-instruction height_cheap; the last iterate of a 3-step crossing / the last 4 of a 10-step one are accurate
    const int ncheap = SLOW ? 0 : (iters <= 3 ? iters - 1 : iters - 4);
#pragma unroll 1
    for (int it = 0; it < ncheap; ++it) {
        const double hgt = height_cheap(fma(t, lx, ox), fma(t, ly, oy), fma(t, lz, oz));
        t = fma(h - hgt, inv_factor, t);
    }
#pragma unroll 1
    for (int it = ncheap; it < iters; ++it) {
        const double hgt = height_sel<SLOW>(fma(t, lx, ox), fma(t, ly, oy), fma(t, lz, oz));
        t = fma(h - hgt, inv_factor, t);
    }
    return t;
}

// Workspace record handed from pass 1 (crossings_kernel) to pass 2 (march_kernel): one column per ray SLOT
// (slot = local tile * 256 + thread), field-major so every field access is a perfectly coalesced 512 B per wave:
//   ws[f * nslots + slot]
//   f = 0        1.0: light ray (polynomial geodesy), 0.0: generic ray
//   light ray:   1..6 h(u) | 7..12 lat(u) [deg] | 13..18 lon(u) [deg] monomial coefficients | 19  |l| / su  (metres per unit of u)
//   generic ray: 1..3 origin ECEF | 4..6 look vector | 7 lat0 | 8 lon0 | 9..12 sin/cos lat0, sin/cos lon0
//   f = 20 .. 20+K  ray parameter of the K+1 level crossings (u for a light ray, t for a generic one)
constexpr int WS_FAST = 0, WS_POLY_H = 1, WS_POLY_LAT = 7, WS_POLY_LON = 13, WS_SCALE = 19;
constexpr int WS_ORIGIN = 1, WS_LOS = 4, WS_LAT0 = 7, WS_LON0 = 8, WS_S0 = 9, WS_C0 = 10, WS_SL0 = 11, WS_CL0 = 12;
constexpr int WS_T = 20;
constexpr int WS_FIELDS_FIXED = WS_T + 1;     // + K

struct RayParams {
    // geometry
    int64_t n;
    int origin_mode, los_mode;
    int64_t nx, ny;
    const double* xpts; const double* ypts;
    const double* lat; const double* lon; const double* xyz;
    const double* los; const double* inc; const double* hd;
    double inc0, hd0;
    // slice
    double ht, zref, max_seg;
    // batch-global state
    unsigned long long* maxlen_bits;   // [MAX_LEVELS] per-level max ray length (bit pattern of a non-negative double); nullptr: no reduction
    int* flags;                        // RDR_FLAG_* bits (OR-reduced)
    const int* nparts_override;        // [K] or nullptr -> ceil(maxlen/max_seg)+1
    int* nslow;                        // number of rays the static classification sent to the generic (slow) kernels
    int projected;                     // the cube is on a projected grid: every ray is integrated by the generic kernel
    // pass 1 -> pass 2 workspace (this launch covers tiles [tile_begin, tile_begin + tile_count))
    double* ws; int64_t nslots;
    int64_t tile_begin, tile_count;
    // outputs
    double* wet; double* hyd;
    // tiling of the whole batch
    int64_t ntiles; int tiles_x;
};

// The slice-uniform level table of build_ray (losreader.py:785-808), computed by one thread into LDS.
__device__ inline int build_levels(const double* zs, int nz, double ht, double zref, double* s_lo, double* s_hi, int* s_kz) {
    int K = 0;
    const double ztop = zs[nz - 1];
    for (int zz = 0; zz < nz - 1; ++zz) {
        double lo = zs[zz], hi = zs[zz + 1];
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

// max over the 64 lanes of a wave, DPP only (no LDS round trips): row (16 lanes) butterflies, then the two
// row-broadcast steps of gfx9 wave64; the result is valid in lane 63.
template <int CTRL, int ROW_MASK>
__device__ __forceinline__ double dpp_max_step(double v) {
    const long long b = __double_as_longlong(v);
    int lo = (int)(b & 0xffffffffLL), hi = (int)(b >> 32);
    // unselected / out-of-range lanes keep their own value (old = src, bound_ctrl off)
    const int lo2 = __builtin_amdgcn_update_dpp(lo, lo, CTRL, ROW_MASK, 0xf, false);
    const int hi2 = __builtin_amdgcn_update_dpp(hi, hi, CTRL, ROW_MASK, 0xf, false);
    const double o = __longlong_as_double(((long long)hi2 << 32) | (unsigned int)lo2);
    return fmax(v, o);
}
__device__ __forceinline__ double wave_max_lane63(double v) {
    v = dpp_max_step<0x111, 0xf>(v);   // row_shr:1
    v = dpp_max_step<0x112, 0xf>(v);   // row_shr:2
    v = dpp_max_step<0x114, 0xf>(v);   // row_shr:4
    v = dpp_max_step<0x118, 0xf>(v);   // row_shr:8   -> lane 15 of every row holds the row max
    v = dpp_max_step<0x142, 0xa>(v);   // row_bcast:15 into rows 1 and 3
    v = dpp_max_step<0x143, 0xc>(v);   // row_bcast:31 into rows 2 and 3 -> lane 63 holds the wave max
    return v;
}

// Shared LDS layout of the two ray kernels.
struct RaySmem {
    double* tab;            // [ys | xs | zs]
    double2* tab2;          // the same axes as (g[i], 1/(g[i+1]-g[i])) pairs (pass 2 cell search)
    double* lo; double* hi; // level table
    unsigned long long* mx; // per-level block max (pass 1)
    int* kz; int* np; int* K;
};
__device__ __forceinline__ RaySmem carve_smem(unsigned char* raw, int ny, int nx, int nz) {
    RaySmem m;
    const int na = ny + nx + nz;
    m.tab2 = reinterpret_cast<double2*>(raw);                 // 16-byte aligned for ds_read_b128
    m.tab = reinterpret_cast<double*>(m.tab2 + na);
    m.lo = m.tab + na;
    m.hi = m.lo + nz;
    m.mx = reinterpret_cast<unsigned long long*>(m.hi + nz);
    m.kz = reinterpret_cast<int*>(m.mx + nz);
    m.np = m.kz + nz;
    m.K = m.np + nz;
    return m;
}

template <typename T2>
__device__ __forceinline__ int fill_tables(const CubeView<T2>& c, const RaySmem& m, double ht, double zref) {
    const int na = c.ny + c.nx + c.nz;
    const int tid = threadIdx.x;
    for (int i = tid; i < na; i += BLOCK) m.tab[i] = c.axes[i];
    __syncthreads();
    for (int i = tid; i < na; i += BLOCK) {
        const bool last = (i == c.ny - 1) || (i == c.ny + c.nx - 1) || (i == na - 1);
        double2 e; e.x = m.tab[i]; e.y = last ? 0.0 : 1.0 / (m.tab[i + 1] - m.tab[i]);
        m.tab2[i] = e;
    }
    if (tid == 0) *m.K = build_levels(m.tab + c.ny + c.nx, c.nz, ht, zref, m.lo, m.hi, m.kz);
    __syncthreads();
    return *m.K;
}

// XCD-aware persistent tile walk: workgroup b runs on XCD b%8 -> each XCD sweeps one contiguous band of tiles.
struct TileWalk {
    int64_t chunk, tt; int nslot, xcd;
    __device__ __forceinline__ TileWalk(int64_t count) {
        xcd = blockIdx.x & 7; nslot = gridDim.x >> 3; chunk = (count + 7) / 8; tt = blockIdx.x >> 3;
    }
    __device__ __forceinline__ bool next(int64_t count, int64_t& local) {
        if (tt >= chunk) return false;
        local = (int64_t)xcd * chunk + tt;
        tt += nslot;
        return local < count;
    }
};

// ---- pass 1: per-ray set-up + level crossings (build_ray, losreader.py:772-835) ------------------------------------
// Optional outputs (both may be on): the per-level batch maximum of the ray length + flags (what delay.py:283,306-311
// reduce over the slice) and the workspace record for pass 2.
// SLOW = false: the light-fp64 path, skips (but counts) the rays the static classification rejects;
// SLOW = true : generic geodesy, processes ONLY those rays, exits at once when there are none.
template <typename T2, bool SLOW>
__global__ __launch_bounds__(BLOCK) __attribute__((amdgpu_waves_per_eu(4, 4))) void crossings_kernel(CubeView<T2> c, RayParams P) {
    if (SLOW && *P.nslow == 0) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const RaySmem m = carve_smem(smem_raw, c.ny, c.nx, c.nz);
    const int K = fill_tables(c, m, P.ht, P.zref);
    const int tid = threadIdx.x;
    const bool reduce = P.maxlen_bits != nullptr;
    if (reduce) for (int k = tid; k < K; k += BLOCK) m.mx[k] = 0ULL;
    __syncthreads();
    int my_flags = 0;
    TileWalk walk(P.tile_count);
    int64_t lt;
    while (walk.next(P.tile_count, lt)) {
        const int64_t t = P.tile_begin + lt;
        int64_t i, row = 0, col = 0; bool active;
        if (P.origin_mode == 0) {
            const int64_t ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
            row = ty * TILE + (tid >> 4); col = tx * TILE + (tid & 15);
            active = row < P.ny && col < P.nx;
            i = row * P.nx + col;
        } else {
            i = t * BLOCK + tid;
            active = i < P.n;
        }
        // ---- origin: llh -> ECEF (delay.py:262-267)
        double lat = 0, lon = 0, ox = qnan(), oy = qnan(), oz = qnan();
        if (active) {
            if (P.origin_mode == 0) { lat = P.ypts[row]; lon = P.xpts[col]; }
            else if (P.lat) { lat = P.lat[i]; lon = P.lon[i]; }
            if (P.origin_mode == 2) {
                ox = P.xyz[3 * i]; oy = P.xyz[3 * i + 1]; oz = P.xyz[3 * i + 2];
                if (!P.lat) { double h0_; ecef2lla(ox, oy, oz, lon, lat, h0_); }   // frame for the delta lat/lon formulas
            } else lla2ecef(lat, lon, P.ht, ox, oy, oz);
        }
        const RayBase base = make_base(lat, lon);
        // ---- look vector (delay.py:270)
        double lx = qnan(), ly = qnan(), lz = qnan();
        if (active) {
            if (P.los_mode == 0) { lx = P.los[3 * i]; ly = P.los[3 * i + 1]; lz = P.los[3 * i + 2]; }
            else if (P.los_mode == 1) inc_hd_to_ecef(P.inc[i], P.hd[i], lat, lon, lx, ly, lz);
            else if (P.los_mode == 2) inc_hd_to_ecef(P.inc0, P.hd0, lat, lon, lx, ly, lz);
            else { lx = base.c0 * base.cl0; ly = base.c0 * base.sl0; lz = base.s0; }   // zenith (losreader.py:302-316)
        }
        // |l| (1 for unit look vectors): ray length between two crossings = (t_hi - t_lo) * |l|   (losreader.py:821)
        const double nl2 = fma(lx, lx, fma(ly, ly, lz * lz));
        const double nl = nl2 * rsq_nr<2>(nl2);
        // Static classification: may the light geodesy be used along the WHOLE ray?  (cos(lat) stays > 0.01 and the ray
        // stays within 0.03 rad of its origin in latitude and longitude.)  t_max <= (zref-ht)/cos(inc) because the local zenith angle of a
        // straight ray decreases with height.
        const double cosi = (lx * base.c0 * base.cl0 + ly * base.c0 * base.sl0 + lz * base.s0) / nl;
        const double gam = (P.zref - P.ht) / (cosi * 6.3e6);                       // bound on the angular travel
        // ... and never crosses the +-180 meridian (the light path does not wrap longitudes): |lon0| + travel < 180 deg
        const bool fast_ok = !active || ((cosi > 0.05) && (base.c0 > gam + 0.02) && (gam < 0.03 * (base.c0 - gam)) &&
                                         (fabs(lon) + 2.0 < 180.0) && !P.projected);
        const int64_t slot = lt * BLOCK + tid;
        if (!SLOW) {
            const unsigned long long slow_mask = __ballot(!fast_ok);
            if (slow_mask && (tid & 63) == 0) atomicAdd(P.nslow, (int)__popcll(slow_mask));
        }
        const bool mine = SLOW ? !fast_ok : fast_ok;      // lanes this instantiation is responsible for
        if (SLOW && !__any(mine)) continue;               // (wave-uniform) nothing to mop up in this wave
        const bool cnt = active && mine;
        double* const w = P.ws ? P.ws + slot : nullptr;
        const int64_t ns = P.nslots;
        if constexpr (SLOW) {
            if (w && mine) {
                w[(int64_t)WS_FAST * ns] = 0.0;
                w[(int64_t)(WS_ORIGIN + 0) * ns] = ox; w[(int64_t)(WS_ORIGIN + 1) * ns] = oy; w[(int64_t)(WS_ORIGIN + 2) * ns] = oz;
                w[(int64_t)(WS_LOS + 0) * ns] = lx; w[(int64_t)(WS_LOS + 1) * ns] = ly; w[(int64_t)(WS_LOS + 2) * ns] = lz;
                w[(int64_t)WS_LAT0 * ns] = lat; w[(int64_t)WS_LON0 * ns] = lon;
                w[(int64_t)WS_S0 * ns] = base.s0; w[(int64_t)WS_C0 * ns] = base.c0;
                w[(int64_t)WS_SL0 * ns] = base.sl0; w[(int64_t)WS_CL0 * ns] = base.cl0;
            }
            double t_hi = 0.0, inv_cosf = 1.0;
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                const double lo = m.lo[k], hi = m.hi[k];
                // first interval: cos_factor is None -> 10 iterations with factor 1 for both ends (losreader.py:812-825);
                // later intervals reuse the previous top as their bottom (losreader.py:811-812)
                double t_lo = t_hi;
                if (k == 0) t_lo = toa_newton_t<true>(ox, oy, oz, lx, ly, lz, lo, 10, 1.0);
                t_hi = toa_newton_t<true>(ox, oy, oz, lx, ly, lz, hi, k == 0 ? 10 : 3, inv_cosf);
                const double L = (t_hi - t_lo) * nl;
                if (k == 0) inv_cosf = L / (hi - lo);                                       // 1/cos_factor, losreader.py:824-825
                if (w && mine) {
                    if (k == 0) w[(int64_t)WS_T * ns] = t_lo;
                    w[(int64_t)(WS_T + k + 1) * ns] = t_hi;
                }
                if (reduce) {
                    // NaN poisons the max exactly as ndarray.max does (delay.py:283): tracked via flags
                    if (cnt) my_flags |= (L != L) ? 1 : 2;
                    const double mx = wave_max_lane63((cnt && L == L) ? L : 0.0);
                    if ((tid & 63) == 63) atomicMax(&m.mx[k], (unsigned long long)__double_as_longlong(mx));
                    if (k == 0 && cnt) {             // first sample of the ray (fraction 0)
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
            const double t_a = fmin(0.0, P.ht) - 1.0;
            const double t_b = fmax(P.zref, (P.zref - P.ht) / (cosi * nl)) + 1.0;
            const double half = 0.5 * (t_b - t_a), mid = 0.5 * (t_b + t_a);
            const double su = 1.0 / half, ou = -mid * su;
            RayPoly q;
            fit_ray_poly(base, ox, oy, oz, lx, ly, lz, mid, half, q);
            const double scale = nl * half;                           // ray length per unit of u
            if (w && mine) {
                w[(int64_t)WS_FAST * ns] = 1.0;
#pragma unroll
                for (int n = 0; n < PN; ++n) {
                    w[(int64_t)(WS_POLY_H + n) * ns] = q.h[n];
                    w[(int64_t)(WS_POLY_LAT + n) * ns] = q.lat[n];
                    w[(int64_t)(WS_POLY_LON + n) * ns] = q.lon[n];
                }
                w[(int64_t)WS_SCALE * ns] = scale;
            }
            // getTopOfAtmosphere carried on u: u0 = u(h); u += (h - H(u)) * su / factor   (losreader.py:724-731)
            double u_hi = 0.0, gain = su, inv_cosf = 1.0;
#pragma unroll 1
            for (int k = 0; k < K; ++k) {
                const double lo = m.lo[k], hi = m.hi[k];
                double u_lo = u_hi;
                if (k == 0) {
                    u_lo = fma(lo, su, ou);
#pragma unroll 1
                    for (int it = 0; it < 10; ++it) u_lo = fma(lo - poly5(q.h, u_lo), su, u_lo);
                }
                u_hi = fma(hi, su, ou);
                if (k == 0) {
#pragma unroll 1
                    for (int it = 0; it < 10; ++it) u_hi = fma(hi - poly5(q.h, u_hi), su, u_hi);
                } else {
#pragma unroll
                    for (int it = 0; it < 3; ++it) u_hi = fma(hi - poly5(q.h, u_hi), gain, u_hi);
                }
                const double L = (u_hi - u_lo) * scale;
                if (k == 0) { inv_cosf = L / (hi - lo); gain = su * inv_cosf; }               // 1/cos_factor, losreader.py:824-825
                if (w && mine) {
                    if (k == 0) w[(int64_t)WS_T * ns] = u_lo;
                    w[(int64_t)(WS_T + k + 1) * ns] = u_hi;
                }
                if (reduce) {
                    if (cnt) my_flags |= (L != L) ? 1 : 2;
                    const double Lv = (cnt && L == L) ? L : 0.0;
                    // most waves do not raise the workgroup's running maximum: one broadcast LDS read decides
                    const double cur = __longlong_as_double((long long)m.mx[k]);
                    if (__any(Lv > cur)) {
                        const double mx = wave_max_lane63(Lv);
                        if ((tid & 63) == 63) atomicMax(&m.mx[k], (unsigned long long)__double_as_longlong(mx));
                    }
                    if (k == 0 && cnt && !(poly5(q.h, u_lo) < c.z_lo)) my_flags |= 4;          // first sample of the ray
                    if (k == K - 1 && cnt && !(poly5(q.h, u_hi) > c.z_hi)) my_flags |= 8;      // last sample of the ray
                }
            }
        }
    }
    if (reduce) {
        __syncthreads();
        for (int k = tid; k < K; k += BLOCK) atomicMax(&P.maxlen_bits[k], m.mx[k]);
        int f = my_flags;
        for (int off = 32; off > 0; off >>= 1) f |= __shfl_xor(f, off, 64);
        if ((tid & 63) == 0 && f) atomicOr(P.flags, f);
    }
}

// ---- pass 2: trapezoid integration of both fields along every ray (delay.py:285-323) -------------------------------
// SLOW as in crossings_kernel: <false> integrates the classified-fast rays with the light geodesy, <true> the rest
// with the generic one (and returns immediately when there are none).
template <typename T2, bool SLOW>
__global__ __launch_bounds__(BLOCK) void march_kernel(CubeView<T2> c, RayParams P, LccParams proj) {
    if (SLOW && *P.nslow == 0 && !P.projected) return;
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const RaySmem m = carve_smem(smem_raw, c.ny, c.nx, c.nz);
    const int K = fill_tables(c, m, P.ht, P.zref);
    const int tid = threadIdx.x;
    for (int k = tid; k < K; k += BLOCK) {
        int np;
        if (P.nparts_override) np = P.nparts_override[k];
        else np = (int)ceil(__longlong_as_double((long long)P.maxlen_bits[k]) / P.max_seg) + 1;   // delay.py:283
        m.np[k] = np;
    }
    __syncthreads();
    const int flags_in = *P.flags;
    const bool clamp_lo = !(flags_in & 4);   // ALL first samples below zmin  (delay.py:306-307)
    const bool clamp_hi = !(flags_in & 8);   // ALL last samples above zmax   (delay.py:310-311)
    TileWalk walk(P.tile_count);
    int64_t lt;
    while (walk.next(P.tile_count, lt)) {
        const int64_t t = P.tile_begin + lt;
        int64_t i; bool active;
        if (P.origin_mode == 0) {
            const int64_t ty = t / P.tiles_x, tx = t - ty * P.tiles_x;
            const int64_t row = ty * TILE + (tid >> 4), col = tx * TILE + (tid & 15);
            active = row < P.ny && col < P.nx;
            i = row * P.nx + col;
        } else {
            i = t * BLOCK + tid;
            active = i < P.n;
        }
        const double* w = P.ws + (lt * BLOCK + tid);
        const int64_t ns = P.nslots;
        const bool fast_ok = !active || w[(int64_t)WS_FAST * ns] != 0.0;
        const bool mine = SLOW ? !fast_ok : fast_ok;
        if (SLOW && !__any(mine)) continue;
        double acc_w = 0.0, acc_h = 0.0;
        double vw_top = 0.0, vh_top = 0.0;    // sample values at the top of the previous segment (= bottom of this one)
        // generic rays: origin / look vector / origin frame; light rays: the three polynomials
        double ox = 0, oy = 0, oz = 0, lx = 0, ly = 0, lz = 0, scale;
        RayPoly q;
        if constexpr (SLOW) {
            ox = w[(int64_t)(WS_ORIGIN + 0) * ns]; oy = w[(int64_t)(WS_ORIGIN + 1) * ns]; oz = w[(int64_t)(WS_ORIGIN + 2) * ns];
            lx = w[(int64_t)(WS_LOS + 0) * ns]; ly = w[(int64_t)(WS_LOS + 1) * ns]; lz = w[(int64_t)(WS_LOS + 2) * ns];
            const double nl2 = fma(lx, lx, fma(ly, ly, lz * lz));
            scale = nl2 * rsq_nr<2>(nl2);                          // |l|: ray length per unit of t
        } else {
#pragma unroll
            for (int n = 0; n < PN; ++n) {
                q.h[n] = w[(int64_t)(WS_POLY_H + n) * ns];
                q.lat[n] = w[(int64_t)(WS_POLY_LAT + n) * ns];
                q.lon[n] = w[(int64_t)(WS_POLY_LON + n) * ns];
            }
            scale = w[(int64_t)WS_SCALE * ns];                      // ray length per unit of u
        }
        double t_hi = w[(int64_t)WS_T * ns];
        double t_next = w[(int64_t)(WS_T + 1) * ns];             // crossings are streamed one level ahead of their use
#pragma unroll 1
        for (int k = 0; k < K; ++k) {
            const double t_lo = t_hi;
            t_hi = t_next;
            if (k + 2 <= K) t_next = w[(int64_t)(WS_T + k + 2) * ns];
            const double dt = t_hi - t_lo;
            const int np = m.np[k];
            const double step = 1.0 / ((double)np - 1.0);        // np.linspace(0,1,np) (delay.py:287)
            const double segw = (dt * scale * 1.0e-6) * step;    // delay.py:315: L*1e-6/(np-1), L = |high-low| (losreader.py:821)
            const double dts = dt * step;                        // sample spacing: low + (j*step)*(high-low), delay.py:292
            const int kz = m.kz[k];
            // j = 0 of this segment is the SAME point as j = np-1 of the previous one (low_xyz is high_xyz,
            // losreader.py:811-812): its interpolated value is reused instead of recomputed (the reference evaluates
            // it twice and gets the same number both times).  The order of accumulation is unchanged.
            if (k > 0) { acc_w = fma(0.5 * segw, vw_top, acc_w); acc_h = fma(0.5 * segw, vh_top, acc_h); }
#pragma unroll 1
            for (int j = (k == 0 ? 0 : 1); j < np; ++j) {
                const double ts = fma((double)j, dts, t_lo);
                double plon, plat, ph;
                if constexpr (SLOW) {
                    ecef2lla(fma(ts, lx, ox), fma(ts, ly, oy), fma(ts, lz, oz), plon, plat, ph);
                    if (proj.kind == 1) { double px_, py_; lcc_forward(proj, plat, plon, px_, py_); plon = px_; plat = py_; }   // ecef_to_model, delay.py:253,295
                } else {                                                              // delay.py:295 through the ray polynomials
                    ph = poly5(q.h, ts); plat = poly5(q.lat, ts); plon = poly5(q.lon, ts);
                }
                // all-pixels z-clamp of the very first / very last sample (delay.py:306-311): when it applies every
                // pixel is below (above) the cube, so "set to zmin" == max(ph, zmin); the bounds are wave-uniform
                const double zfloor = (clamp_lo && k == 0 && j == 0) ? c.z_lo : -__builtin_huge_val();
                const double zceil = (clamp_hi && k == K - 1 && j == np - 1) ? c.z_hi : __builtin_huge_val();
                ph = fmin(fmax(ph, zfloor), zceil);
                double vw, vh;
                sample_cube(c, m.tab2, plat, plon, ph, kz, vw, vh);                   // delay.py:298,319
                const double wt = ((j == 0) | (j == np - 1)) ? 0.5 * segw : segw;     // delay.py:314-315
                acc_w = fma(wt, vw, acc_w); acc_h = fma(wt, vh, acc_h);               // delay.py:323
                vw_top = vw; vh_top = vh;                                             // after the loop: value at j = np-1
            }
        }
        if (active && mine) { P.wet[i] = acc_w; P.hyd[i] = acc_h; }
    }
}

}  // namespace rdr
