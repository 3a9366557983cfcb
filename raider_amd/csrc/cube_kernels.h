// Cube packing / blending, look vectors, zenith gathers, partition exchange, azimuth-time weights.
// Part of libraider_hip.so (single translation unit: included by raider_hip.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>

#include <cstdint>

#include "raider_kernels.h"

using namespace rdr;

// ------------------------------------------------------------------------------------------------
// small kernels
// ------------------------------------------------------------------------------------------------
// (y,x,z) interleaved device cube from two strided source fields.  SWAP: the source is in the OTHER byte order (a NetCDF-3 file is
// big-endian): the bytes are swapped here, on the way through, so that a file mapping can be uploaded as it is - no host pass over
// the data.  nan_flag: set to 1 when a value is NaN (delayFcns.py:50-52 scans the fields for NaNs on the host).
template <typename T, typename T2, bool SWAP>
__global__ void pack_cube_kernel(const T* __restrict__ wet, const T* __restrict__ hyd, T2* __restrict__ dst,
                                 int64_t ny, int64_t nx, int64_t nz, int64_t sy, int64_t sx, int64_t sz,
                                 int fy, int fx, int fz, int* __restrict__ nan_flag) {
    const int64_t total = ny * nx * nz;
    bool bad = false;
    auto fix = [](T v) -> T {
        if constexpr (!SWAP) return v;
        else if constexpr (sizeof(T) == 4) return __uint_as_float(__builtin_bswap32(__float_as_uint(v)));
        else return __longlong_as_double((long long)__builtin_bswap64((unsigned long long)__double_as_longlong(v)));
    };
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t iz = o % nz, r = o / nz, ix = r % nx, iy = r / nx;
        const int64_t jy = fy ? ny - 1 - iy : iy, jx = fx ? nx - 1 - ix : ix, jz = fz ? nz - 1 - iz : iz;
        const int64_t s = jy * sy + jx * sx + jz * sz;
        T2 v; v.x = fix(wet[s]); v.y = fix(hyd[s]);
        bad |= (v.x != v.x) | (v.y != v.y);
        dst[o] = v;
    }
    if (nan_flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
}

// The same for sources whose x axis is contiguous (sx == 1: file order (z, y, x), weatherModel.py:685-693 - and the planar results of
// _build_cube that become an intermediate delay cube).  The generic kernel above walks the OUTPUT (z fastest), i.e. reads such a source at a
// stride of ny * nx elements: every 8 B from its own line (measured on the 80 x 151 x 201 intermediate cube of BASELINE configs[1]: 272 MB of
// HBM reads for a 39 MB cube, 50 us - the longest kernel of that step).  Here a workgroup moves a 32 (x) x 32 (z) tile of one y row through
// LDS: reads coalesced along x (256 B per half-wave), writes coalesced along z (32 x sizeof(T2) contiguous bytes per half-wave).
template <typename T, typename T2, bool SWAP>
__global__ __launch_bounds__(256) void pack_cube_xfast_kernel(const T* __restrict__ wet, const T* __restrict__ hyd, T2* __restrict__ dst,
                                                              int64_t ny, int64_t nx, int64_t nz, int64_t sy, int64_t sz,
                                                              int fy, int fx, int fz, int* __restrict__ nan_flag) {
    __shared__ T2 tile[32][33];                      // [z][x], padded: the transposed read walks a column
    auto fix = [](T v) -> T {
        if constexpr (!SWAP) return v;
        else if constexpr (sizeof(T) == 4) return __uint_as_float(__builtin_bswap32(__float_as_uint(v)));
        else return __longlong_as_double((long long)__builtin_bswap64((unsigned long long)__double_as_longlong(v)));
    };
    const int64_t tx_n = (nx + 31) / 32, tz_n = (nz + 31) / 32, ntiles = ny * tx_n * tz_n;
    const int lx = threadIdx.x & 31, lr = threadIdx.x >> 5;          // 32 lanes along the fast axis, 8 rows per pass
    bool bad = false;
    for (int64_t t = blockIdx.x; t < ntiles; t += gridDim.x) {
        const int64_t tz = t % tz_n, r = t / tz_n, tx = r % tx_n, iy = r / tx_n;
        const int64_t jy = fy ? ny - 1 - iy : iy;
        const int64_t x0 = tx * 32, z0 = tz * 32;
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                // read: lanes along x, rows = z
            const int64_t ix = x0 + lx, iz = z0 + lr + 8 * k;
            if (ix < nx && iz < nz) {
                const int64_t jx = fx ? nx - 1 - ix : ix, jz = fz ? nz - 1 - iz : iz;
                const int64_t src = jy * sy + jx + jz * sz;
                T2 v; v.x = fix(wet[src]); v.y = fix(hyd[src]);
                bad |= (v.x != v.x) | (v.y != v.y);
                tile[lr + 8 * k][lx] = v;
            }
        }
        __syncthreads();
#pragma unroll
        for (int k = 0; k < 4; ++k) {                                // write: lanes along z, rows = x
            const int64_t iz = z0 + lx, ix = x0 + lr + 8 * k;
            if (ix < nx && iz < nz) dst[(iy * nx + ix) * nz + iz] = tile[lx][lr + 8 * k];
        }
        __syncthreads();
    }
    if (nan_flag && __any(bad) && (threadIdx.x & 63) == 0) atomicOr(nan_flag, 1);
}

template <typename T, typename T2>
__global__ void unpack_cube_kernel(const T2* __restrict__ src, T* __restrict__ wet, T* __restrict__ hyd, int64_t total) {
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const T2 v = src[o]; wet[o] = v.x; hyd[o] = v.y;
    }
}

// cli/raider.py:817-819: sum([w*ds[var]]) = 0 + w1*a + w2*b in the array dtype (numpy<2 value-based casting).
// A pure stream (24 B per f32 cell: two reads, one write): both fields take the same weights, so the interleaved cube is one flat
// array of scalars; 16 bytes per lane and access, four accesses of each input in flight per lane before the first use,
// non-temporal (the epochs are read once, the blend is read by other kernels later).
template <typename T>
__global__ __launch_bounds__(256) void blend_kernel(const T* __restrict__ a, T w1, const T* __restrict__ b, T w2, T* __restrict__ out, int64_t nscalars) {
    constexpr int V = 16 / sizeof(T);
    typedef T VT __attribute__((ext_vector_type(V)));
    constexpr int U = 4;
    const VT* __restrict__ A = reinterpret_cast<const VT*>(a);
    const VT* __restrict__ B = reinterpret_cast<const VT*>(b);
    VT* __restrict__ Oo = reinterpret_cast<VT*>(out);
    const int64_t nvec = nscalars / V;
    const int64_t stride = (int64_t)gridDim.x * blockDim.x;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nvec; i += stride * U) {
        VT va[U], vb[U];
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * stride;
            if (j < nvec) { va[u] = __builtin_nontemporal_load(A + j); vb[u] = __builtin_nontemporal_load(B + j); }
        }
#pragma unroll
        for (int u = 0; u < U; ++u) {
            const int64_t j = i + u * stride;
            if (j < nvec) {
                VT r;
#pragma unroll
                for (int e = 0; e < V; ++e) {
#pragma clang fp contract(off)
                    const T p = w1 * va[u][e], q = w2 * vb[u][e];
                    r[e] = p + q;
                }
                __builtin_nontemporal_store(r, Oo + j);
            }
        }
    }
    if (blockIdx.x == 0 && threadIdx.x == 0) {
        for (int64_t o = nvec * V; o < nscalars; ++o) {
#pragma clang fp contract(off)
            const T p = w1 * a[o], q = w2 * b[o];
            out[o] = p + q;
        }
    }
}

// ---- device-resident exchange of the pass-1 partition (multi-GPU: the all-reduce runs on this buffer, no host round trip) ----
// partition[0..K) = per-level maxima of the ray length (doubles), partition[K..K+4) = the four RDR_FLAG_* bits as 0.0 / 1.0,
// so that ONE element-wise MAX all-reduce combines the shards (MAX of non-negative doubles = MAX of their bit patterns).
__global__ void pack_partition_kernel(const unsigned long long* __restrict__ bits, const int* __restrict__ flags, int K, double* __restrict__ out) {
    const int f = *flags;
    for (int k = threadIdx.x; k < K + 4; k += blockDim.x)
        out[k] = k < K ? __longlong_as_double((long long)bits[k]) : (((f >> (k - K)) & 1) ? 1.0 : 0.0);
}

// np.isnan(out).any() per slice (delay.py:187) on the device, before the outputs leave it: a[0 .. n) and b[0 .. n) are consecutive
// slices of `per` values each; slice s gets RDR_FLAG_NAN_OUTPUT (64) or-ed into flags[s] when either field holds a NaN.
__global__ __launch_bounds__(256) void nan_scan_kernel(const double* __restrict__ a, const double* __restrict__ b, int64_t n, int64_t per,
                                                       int* __restrict__ flags) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double x = a[i], y = b[i];
        if ((x != x) | (y != y)) {
            int* f = flags + i / per;
            if (!(__atomic_load_n(f, __ATOMIC_RELAXED) & 64)) atomicOr(f, 64);
        }
    }
}

__global__ void unpack_partition_kernel(const double* __restrict__ in, int K, unsigned long long* __restrict__ bits, int* __restrict__ flags) {
    for (int k = threadIdx.x; k < K; k += blockDim.x) bits[k] = (unsigned long long)__double_as_longlong(in[k]);
    if (threadIdx.x == 0) {
        int f = 0;
        for (int b = 0; b < 4; ++b) if (in[K + b] > 0.0) f |= 1 << b;
        *flags = f | (*flags & (32 | 64));       // (a bad per-ray height seen by THIS rank's pass 1 keeps poisoning its outputs)
    }
}

// ---- azimuth-time-grid temporal weighting -----------------------------------------------------------------------------
// get_inverse_weights_for_dates (s1_azimuth_timing.py:326-399): w_d = m_d / (|t - date_d| + reg) / sum_d(...), m_d = 1 when
// |t - date_d| <= window.  A voxel with no date inside the window divides 0 by 0 -> NaN, as in the reference.
struct DateSet { int nd; double date[8]; double window, reg; };

__global__ void time_weights_kernel(DateSet D, const double* __restrict__ az, int64_t n, double* __restrict__ w, int* __restrict__ any_in_window) {
    int seen = 0;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double t = az[i];
        double m[8], total = 0.0;
        for (int d = 0; d < D.nd; ++d) {
            const double diff = fabs(t - D.date[d]);
            const bool in = diff <= D.window;
            seen |= in;
            m[d] = (1.0 / (diff + D.reg)) * (in ? 1.0 : 0.0);
            total += m[d];
        }
        for (int d = 0; d < D.nd; ++d) w[(int64_t)d * n + i] = m[d] / total;
    }
    if (seen) atomicOr(any_in_window, 1);
}

// cli/raider.py:817-819 with per-voxel weights: sum([wgt * ds[var] ...]) = ((0 + w0 a0) + w1 a1) + ... in f64 (a weight ARRAY
// is float64, so numpy promotes the f32 fields).  Weights are in file order (z, y, x), cubes in device order (y, x, z).
struct CubeSet { int nd; const void* v[8]; };

template <typename T2>
__global__ void blend_weighted_kernel(CubeSet S, const double* __restrict__ w, int64_t ny, int64_t nx, int64_t nz, double2* __restrict__ out) {
    const int64_t total = ny * nx * nz;
    for (int64_t o = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; o < total; o += (int64_t)gridDim.x * blockDim.x) {
        const int64_t iz = o % nz, ix = (o / nz) % nx, iy = o / (nz * nx);
        const int64_t wi = (iz * ny + iy) * nx + ix;
        double aw = 0.0, ah = 0.0;
        for (int d = 0; d < S.nd; ++d) {
            const T2 v = reinterpret_cast<const T2*>(S.v[d])[o];
            const double wd = w[(int64_t)d * total + wi];
            {
#pragma clang fp contract(off)
                const double pw = wd * (double)v.x, ph = wd * (double)v.y;
                aw = aw + pw; ah = ah + ph;
            }
        }
        double2 r; r.x = aw; r.y = ah;
        out[o] = r;
    }
}

// A point query = scipy RGI __call__ on both fields (delay.py:120-121,214) and, optionally, the second stage of tropo_delay's point
// branch fused behind it (delay.py:110-128): the points as the caller's THREE arrays (x != NULL; no packed (n,3) copy on the host) or
// packed (y = pts[n,3], x = z = NULL), and - for a projected line of sight - the division of Conventional.__call__
// (losreader.py:130-133) before the two values leave the device.  Either output may be NULL.
//   pmode 0: none; 1: proj[i] = incidence (deg), delay / cosd(inc) as inc_hd_to_enu(...)[..., -1] gives it; 2: one incidence inc0;
//   3: proj[i] = the divisor itself (cos of the look angle from state_to_los, losreader.py:122-128)
__device__ __forceinline__ double project_divisor(int pmode, const double* __restrict__ proj, double inc0, int64_t i) {
    if (pmode == 1) return cos(proj[i] * DEG_TO_RAD);
    if (pmode == 2) return cos(inc0 * DEG_TO_RAD);
    return proj[i];
}

struct PointQuery {
    const double *y, *x, *z;      // x == NULL: y is the packed (n,3) array
    int pmode; const double* proj; double inc0;
    __device__ __forceinline__ void point(int64_t i, double& py, double& px, double& pz) const {
        if (x) { py = y[i]; px = x[i]; pz = z[i]; }
        else { py = y[3 * i]; px = y[3 * i + 1]; pz = y[3 * i + 2]; }
    }
    __device__ __forceinline__ void store(int64_t i, double w, double h, double* __restrict__ wet, double* __restrict__ hyd) const {
        if (pmode) { const double up = project_divisor(pmode, proj, inc0, i); w = w / up; h = h / up; }
        if (wet) wet[i] = w;
        if (hyd) hyd[i] = h;
    }
};

// A4/A5: scipy RGI at packed points (n,3) = (y,x,z)
template <typename T2>
__global__ __launch_bounds__(256) void interp_points_kernel(CubeView<T2> c, PointQuery Q, int64_t n,
                                                            double* __restrict__ wet, double* __restrict__ hyd, int axes_in_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const double* s_y = c.axes;                       // very long axes stay in global memory (L1 / L2 hits)
    if (axes_in_lds) {
        double* t = reinterpret_cast<double*>(smem_raw);
        for (int i = threadIdx.x; i < c.ny + c.nx + c.nz; i += blockDim.x) t[i] = c.axes[i];
        __syncthreads();
        s_y = t;
    }
    const double* s_x = s_y + c.ny;
    const double* s_z = s_x + c.nx;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double y, x, z;
        Q.point(i, y, x, z);
        double w, h;
        trilinear(c, s_y, s_x, s_z, y, x, z, w, h);
        Q.store(i, w, h, wet, hyd);
    }
}

// ---- the two-epoch blend made FOR the point gather (round 6) -----------------------------------------------------------------------------
// When the blended cube's only reader is a gather at random points (BASELINE configs[4] on one rank: 5 M stations on a 50 M-cell cube), the
// blend may as well write the layout the gather wants: neighbouring x columns PAIRED,
//     P[(iy * npx + px) * nz + iz] = (wet, hydro of column 2 px | wet, hydro of column 2 px + 1),   npx = ceil(nx / 2)
// so that a point whose cell starts on an even column finds both x columns of a y row in 32 contiguous bytes (z, z+1): 2 cache lines per
// point instead of 4; an odd cell still reads 4 - 3 on average (measured 470 B per point against 572, tools/probes/pair_layout_probe.hip).
// Same arithmetic as blend_kernel (the two products rounded separately, then added, in the cube's dtype): the same bits, other addresses.
// One thread per (row, pair, ZV consecutive z): ZV x 8-byte runs of four source columns in, ZV x 16 contiguous bytes out (f32 cubes).
template <typename T2, int ZV>
__global__ __launch_bounds__(256) void blend_pair_kernel(const T2* __restrict__ a, decltype(T2().x) w1, const T2* __restrict__ b, decltype(T2().x) w2,
                                                         T2* __restrict__ out, int ny, int nx, int nz) {
    typedef decltype(T2().x) T;
    const int npx = (nx + 1) >> 1, nzv = nz / ZV;
    const int64_t total = (int64_t)ny * npx * nzv;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < total; i += (int64_t)gridDim.x * blockDim.x) {
        const int izv = (int)(i % nzv);
        const int64_t p = i / nzv;
        const int px = (int)(p % npx), iy = (int)(p / npx);
        const int64_t se = ((int64_t)iy * nx + 2 * px) * nz + (int64_t)izv * ZV;      // even column of the pair (source element)
        const bool has_odd = 2 * px + 1 < nx;
        // (non-temporal: the epochs are read once; ZV = 2: one 16-byte access per column and epoch, 32 contiguous bytes out)
        typedef T VZ __attribute__((ext_vector_type(2 * ZV)));
        typedef T V4 __attribute__((ext_vector_type(4)));
        const VZ zero = {};
        const VZ ae = __builtin_nontemporal_load(reinterpret_cast<const VZ*>(a + se)), be = __builtin_nontemporal_load(reinterpret_cast<const VZ*>(b + se));
        const VZ ao = has_odd ? __builtin_nontemporal_load(reinterpret_cast<const VZ*>(a + se + nz)) : zero;
        const VZ bo = has_odd ? __builtin_nontemporal_load(reinterpret_cast<const VZ*>(b + se + nz)) : zero;
        V4* o = reinterpret_cast<V4*>(out + 2 * (p * nz + (int64_t)izv * ZV));
#pragma unroll
        for (int k = 0; k < ZV; ++k) {
#pragma clang fp contract(off)
            V4 r;
            { const T p1 = w1 * ae[2 * k], q1 = w2 * be[2 * k]; r[0] = p1 + q1; }
            { const T p1 = w1 * ae[2 * k + 1], q1 = w2 * be[2 * k + 1]; r[1] = p1 + q1; }
            { const T p1 = w1 * ao[2 * k], q1 = w2 * bo[2 * k]; r[2] = p1 + q1; }
            { const T p1 = w1 * ao[2 * k + 1], q1 = w2 * bo[2 * k + 1]; r[3] = p1 + q1; }
            o[k] = r;
        }
    }
}

// scipy RGI (trilinear<> of raider_kernels.h, the same weights, corner order and sum) on a PAIRED cube
template <typename T2>
__global__ __launch_bounds__(256) void interp_points_pair_kernel(CubeView<T2> c, const T2* __restrict__ P, PointQuery Q, int64_t n,
                                                                 double* __restrict__ wet, double* __restrict__ hyd, int axes_in_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const double* s_y = c.axes;
    if (axes_in_lds) {
        double* t = reinterpret_cast<double*>(smem_raw);
        for (int i = threadIdx.x; i < c.ny + c.nx + c.nz; i += blockDim.x) t[i] = c.axes[i];
        __syncthreads();
        s_y = t;
    }
    const double* s_x = s_y + c.ny;
    const double* s_z = s_x + c.nx;
    const int npx = (c.nx + 1) >> 1;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double y, x, z;
        Q.point(i, y, x, z);
        double sw = qnan(), sh = qnan();
        const bool inside = (y >= c.y_lo) && (y <= c.y_hi) && (x >= c.x_lo) && (x <= c.x_hi) && (z >= c.z_lo) && (z <= c.z_hi);
        if (inside) {
            const int iy = find_cell(s_y, c.ny, y, c.y_lo, c.inv_dy, c.uni_y);
            const int ix = find_cell(s_x, c.nx, x, c.x_lo, c.inv_dx, c.uni_x);
            const int iz = find_cell(s_z, c.nz, z, c.z_lo, c.inv_dz, c.uni_z);
            const double ty = (y - s_y[iy]) / (s_y[iy + 1] - s_y[iy]);
            const double tx = (x - s_x[ix]) / (s_x[ix + 1] - s_x[ix]);
            const double tz = (z - s_z[iz]) / (s_z[iz + 1] - s_z[iz]);
            // Whole 16-byte (f32) pair entries, addressed without a data-dependent pointer (left to per-corner selects of ADDRESSES the compiler
            // issues sixteen 4-byte loads): entries e, e+1 = levels iz, iz+1 of pair px - both x columns of an even cell; an odd cell takes
            // their odd halves and the even halves of the next pair's entries (two more loads, for those lanes only).
            typedef decltype(T2().x) T;
            typedef T V4 __attribute__((ext_vector_type(4)));
            const V4* P4 = reinterpret_cast<const V4*>(P);
            const int px = ix >> 1;
            const bool even = (ix & 1) == 0;
            T2 v[8];                                   // corners in (y, x, z) lexicographic order
#pragma unroll
            for (int r = 0; r < 2; ++r) {
                const int64_t e = ((int64_t)(iy + r) * npx + px) * c.nz + iz;
                const V4 E0 = P4[e], E1 = P4[e + 1];
                V4 F0 = E0, F1 = E1;
                if (!even) { F0 = P4[e + c.nz]; F1 = P4[e + c.nz + 1]; }
                v[4 * r + 0].x = even ? E0[0] : E0[2]; v[4 * r + 0].y = even ? E0[1] : E0[3];       // (x0, z0)
                v[4 * r + 1].x = even ? E1[0] : E1[2]; v[4 * r + 1].y = even ? E1[1] : E1[3];       // (x0, z1)
                v[4 * r + 2].x = even ? E0[2] : F0[0]; v[4 * r + 2].y = even ? E0[3] : F0[1];       // (x1, z0)
                v[4 * r + 3].x = even ? E1[2] : F1[0]; v[4 * r + 3].y = even ? E1[3] : F1[1];       // (x1, z1)
            }
            const double wy0 = 1.0 - ty, wx0 = 1.0 - tx, wz0 = 1.0 - tz;
            const double a00 = wy0 * wx0, a01 = wy0 * tx, a10 = ty * wx0, a11 = ty * tx;
            const double k[8] = {a00 * wz0, a00 * tz, a01 * wz0, a01 * tz, a10 * wz0, a10 * tz, a11 * wz0, a11 * tz};
            sw = 0.0; sh = 0.0;
#pragma unroll
            for (int j = 0; j < 8; ++j) { sw += (double)v[j].x * k[j]; sh += (double)v[j].y * k[j]; }
        }
        Q.store(i, sw, sh, wet, hyd);
    }
}

// The two-epoch temporal interpolation (cli/raider.py:817-819) applied ON THE FLY at the eight corners of every query point instead of
// to the whole cube first: corner = w1 * a + w2 * b in the cube's own dtype, the two products rounded separately - exactly blend_kernel's
// arithmetic - then the gather of trilinear<>: the same bits as blend-then-gather.  Reads eight lines per point instead of four but
// never touches the 24 B per cell of a blend: it wins when a rank queries fewer points than ~5 % of the cube's cells - the station block of
// one rank of an 8-GPU job (BASELINE configs[4]: 625 k of 5 M stations on a 50 M-cell cube), where the replicated blend is what stops
// the job from scaling.
template <typename T2>
__global__ __launch_bounds__(256) void interp_points_blend_kernel(CubeView<T2> c, const T2* __restrict__ vb, double w1d, double w2d, PointQuery Q, int64_t n,
                                                                  double* __restrict__ wet, double* __restrict__ hyd, int axes_in_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    typedef decltype(T2().x) T;
    const T w1 = (T)w1d, w2 = (T)w2d;
    const double* s_y = c.axes;
    if (axes_in_lds) {
        double* t = reinterpret_cast<double*>(smem_raw);
        for (int i = threadIdx.x; i < c.ny + c.nx + c.nz; i += blockDim.x) t[i] = c.axes[i];
        __syncthreads();
        s_y = t;
    }
    const double* s_x = s_y + c.ny;
    const double* s_z = s_x + c.nx;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double y, x, z;
        Q.point(i, y, x, z);
        double sw = qnan(), sh = qnan();
        const bool inside = (y >= c.y_lo) && (y <= c.y_hi) && (x >= c.x_lo) && (x <= c.x_hi) && (z >= c.z_lo) && (z <= c.z_hi);
        if (inside) {
            const int iy = find_cell(s_y, c.ny, y, c.y_lo, c.inv_dy, c.uni_y);
            const int ix = find_cell(s_x, c.nx, x, c.x_lo, c.inv_dx, c.uni_x);
            const int iz = find_cell(s_z, c.nz, z, c.z_lo, c.inv_dz, c.uni_z);
            const double ty = (y - s_y[iy]) / (s_y[iy + 1] - s_y[iy]);
            const double tx = (x - s_x[ix]) / (s_x[ix + 1] - s_x[ix]);
            const double tz = (z - s_z[iz]) / (s_z[iz + 1] - s_z[iz]);
            const int64_t o00 = ((int64_t)iy * c.nx + ix) * c.nz + iz;
            const int64_t off[4] = {o00, o00 + c.nz, o00 + (int64_t)c.nx * c.nz, o00 + (int64_t)c.nx * c.nz + c.nz};
            T2 a[8], b[8];
#pragma unroll
            for (int k = 0; k < 4; ++k) { a[2 * k] = c.v[off[k]]; a[2 * k + 1] = c.v[off[k] + 1]; b[2 * k] = vb[off[k]]; b[2 * k + 1] = vb[off[k] + 1]; }
            double w[8], h[8];
#pragma unroll
            for (int k = 0; k < 8; ++k) {
#pragma clang fp contract(off)
                const T pw = w1 * a[k].x, qw = w2 * b[k].x, ph = w1 * a[k].y, qh = w2 * b[k].y;
                w[k] = (double)(T)(pw + qw); h[k] = (double)(T)(ph + qh);
            }
            const double wy0 = 1.0 - ty, wx0 = 1.0 - tx, wz0 = 1.0 - tz;
            const double a00 = wy0 * wx0, a01 = wy0 * tx, a10 = ty * wx0, a11 = ty * tx;
            const double k0 = a00 * wz0, k1 = a00 * tz, k2 = a01 * wz0, k3 = a01 * tz;
            const double k4 = a10 * wz0, k5 = a10 * tz, k6 = a11 * wz0, k7 = a11 * tz;
            sw = 0.0; sh = 0.0;
            sw += w[0] * k0; sh += h[0] * k0;
            sw += w[1] * k1; sh += h[1] * k1;
            sw += w[2] * k2; sh += h[2] * k2;
            sw += w[3] * k3; sh += h[3] * k3;
            sw += w[4] * k4; sh += h[4] * k4;
            sw += w[5] * k5; sh += h[5] * k5;
            sw += w[6] * k6; sh += h[6] * k6;
            sw += w[7] * k7; sh += h[7] * k7;
        }
        Q.store(i, sw, sh, wet, hyd);
    }
}

// ---- station queries on a cube that fits no cache: the corner-quad copy ----------------------------------------------------
// A random point's 2 x 2 x 2 corners sit in FOUR columns of the (y,x,z) cube, i.e. four 128 B lines for 16 B each: 5 M stations on
// a 1000 x 1000 x 50 f32 cube move 2.78 GB (556 B per point) for 0.52 GB of algorithmic bytes - and the kernel runs at the HBM
// rate of those lines (profiles/r02_secondary.json).  The quad copy stores the cube CELL-COLUMN-major instead: for every cell
// column (iy, ix) the four corner columns interleaved level by level, in 128 B blocks of LPB levels that overlap by one level
//     block j of cell column (iy, ix):  levels CPB j .. CPB j + LPB - 1,  each level = v(y0,x0) v(y0,x1) v(y1,x0) v(y1,x1)
// so that the two levels of ANY cell iz lie in ONE block (j = iz / CPB): one line per point, 168 B measured instead of 556.
// f32 cubes: LPB = 4 levels, CPB = 3 cells per block (5.3 x the cube's bytes); f64 cubes: LPB = 2, CPB = 1 (8 x).
// Built once per cube, on demand (rdr_cube_point_index / the second large rdr_interp3 call on a cube beyond 32 MB).
template <typename T2> struct Quad {
    static constexpr int LPB = 128 / (4 * (int)sizeof(T2));
    static constexpr int CPB = LPB - 1;
};

// one thread per 16-byte part of a block: a wave writes 8 consecutive blocks = 1 KB
template <typename T2>
__global__ __launch_bounds__(256) void quad_build_kernel(const T2* __restrict__ v, int ny, int nx, int nz, int nblk, uint4* __restrict__ q) {
    const int64_t total = (int64_t)(ny - 1) * (nx - 1) * nblk * 8;
    // every workgroup a CONTIGUOUS range of parts, the ranges of the workgroups of one XCD (blockIdx % 8) adjacent: an XCD then builds one
    // band of cell rows and reads each source column from ITS L2 for both rows and both columns that share it (interleaved over the whole
    // cube every XCD read everything: 1.4 GB fetched for the 400 MB cube, 0.72 ms; now write-bound)
    const int64_t nb = gridDim.x, per_x = (nb + 7) / 8;
    const int64_t bid = (int64_t)(blockIdx.x & 7) * per_x + (blockIdx.x >> 3);
    const int64_t per = ((total + per_x * 8 - 1) / (per_x * 8) + 255) / 256 * 256;
    const int64_t t_end = min(total, (bid + 1) * per);
    // UQ parts per thread and trip, their loads issued together (one part per trip left every thread with a single 8 / 16 B load in
    // flight: latency-bound at half the write rate)
    constexpr int UQ = 4;
    typedef unsigned U4 __attribute__((ext_vector_type(4)));
    for (int64_t t0 = bid * per + threadIdx.x; t0 < t_end; t0 += (int64_t)blockDim.x * UQ) {
        T2 va[UQ], vb[UQ];
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int64_t t = min(t0 + (int64_t)u * blockDim.x, t_end - 1);
            const int p = (int)(t & 7);
            const int64_t b = t >> 3;
            int j, ix, iy;
            if (total < (1LL << 34)) {                // (32-bit divisions: a quarter of the instructions of the 64-bit ones)
                const unsigned bu = (unsigned)b, colu = bu / (unsigned)nblk;
                j = (int)(bu - colu * (unsigned)nblk);
                iy = (int)(colu / (unsigned)(nx - 1)); ix = (int)(colu - (unsigned)iy * (unsigned)(nx - 1));
            } else {
                j = (int)(b % nblk);
                const int64_t col = b / nblk;
                ix = (int)(col % (nx - 1)); iy = (int)(col / (nx - 1));
            }
            if constexpr (sizeof(T2) == 8) {        // float2: part p = level p/2, row y0 / y1 = p%2, both x corners
                const int lev = min(j * Quad<T2>::CPB + (p >> 1), nz - 1);
                const T2* a = v + ((int64_t)(iy + (p & 1)) * nx + ix) * nz + lev;
                va[u] = a[0]; vb[u] = a[nz];
            } else {                                  // double2: part p = level p/4, corner p%4
                const int lev = min(j * Quad<T2>::CPB + (p >> 2), nz - 1);
                va[u] = v[((int64_t)(iy + ((p >> 1) & 1)) * nx + ix + (p & 1)) * nz + lev];
            }
        }
        asm volatile("" ::: "memory");              // (all loads of the trip ahead of its first store)
#pragma unroll
        for (int u = 0; u < UQ; ++u) {
            const int64_t t = t0 + (int64_t)u * blockDim.x;
            if (t >= t_end) break;
            U4 o4;
            if constexpr (sizeof(T2) == 8) {
                o4.x = __float_as_uint(va[u].x); o4.y = __float_as_uint(va[u].y); o4.z = __float_as_uint(vb[u].x); o4.w = __float_as_uint(vb[u].y);
            } else {
                const unsigned long long lo = (unsigned long long)__double_as_longlong(va[u].x), hi = (unsigned long long)__double_as_longlong(va[u].y);
                o4.x = (unsigned)lo; o4.y = (unsigned)(lo >> 32); o4.z = (unsigned)hi; o4.w = (unsigned)(hi >> 32);
            }
            // (written once, read by other kernels much later: non-temporal, so that the 2 GB of output do not evict the source columns
            // the neighbouring cell columns and the next cell row are about to re-read from this XCD's L2)
            __builtin_nontemporal_store(o4, reinterpret_cast<U4*>(q) + t);
        }
    }
}

// the same interpolant as interp_points_kernel / trilinear<> (cell search, weights, summation order: bit-identical results), corners
// from the quad copy: 64 contiguous bytes of one block for an f32 cube, the whole 128 B block for an f64 one
template <typename T2>
__global__ __launch_bounds__(256) void interp_points_quad_kernel(CubeView<T2> c, const uint4* __restrict__ q, int nblk, PointQuery Q,
                                                                 int64_t n, double* __restrict__ wet, double* __restrict__ hyd, int axes_in_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const double* s_y = c.axes;
    if (axes_in_lds) {
        double* t = reinterpret_cast<double*>(smem_raw);
        for (int i = threadIdx.x; i < c.ny + c.nx + c.nz; i += blockDim.x) t[i] = c.axes[i];
        __syncthreads();
        s_y = t;
    }
    const double* s_x = s_y + c.ny;
    const double* s_z = s_x + c.nx;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double y, x, z;
        Q.point(i, y, x, z);
        double sw = qnan(), sh = qnan();
        const bool inside = (y >= c.y_lo) && (y <= c.y_hi) && (x >= c.x_lo) && (x <= c.x_hi) && (z >= c.z_lo) && (z <= c.z_hi);
        if (inside) {
            const int iy = find_cell(s_y, c.ny, y, c.y_lo, c.inv_dy, c.uni_y);
            const int ix = find_cell(s_x, c.nx, x, c.x_lo, c.inv_dx, c.uni_x);
            const int iz = find_cell(s_z, c.nz, z, c.z_lo, c.inv_dz, c.uni_z);
            const double ty = (y - s_y[iy]) / (s_y[iy + 1] - s_y[iy]);
            const double tx = (x - s_x[ix]) / (s_x[ix + 1] - s_x[ix]);
            const double tz = (z - s_z[iz]) / (s_z[iz + 1] - s_z[iz]);
            const int j = iz / Quad<T2>::CPB, l0 = iz - j * Quad<T2>::CPB;
            const uint4* blk = q + (((int64_t)iy * (c.nx - 1) + ix) * nblk + j) * 8;
            double w[8], h[8];
            if constexpr (sizeof(T2) == 8) {
                const uint4 A = blk[2 * l0], B = blk[2 * l0 + 1], C = blk[2 * l0 + 2], D = blk[2 * l0 + 3];
                w[0] = (double)__uint_as_float(A.x); h[0] = (double)__uint_as_float(A.y); w[2] = (double)__uint_as_float(A.z); h[2] = (double)__uint_as_float(A.w);
                w[4] = (double)__uint_as_float(B.x); h[4] = (double)__uint_as_float(B.y); w[6] = (double)__uint_as_float(B.z); h[6] = (double)__uint_as_float(B.w);
                w[1] = (double)__uint_as_float(C.x); h[1] = (double)__uint_as_float(C.y); w[3] = (double)__uint_as_float(C.z); h[3] = (double)__uint_as_float(C.w);
                w[5] = (double)__uint_as_float(D.x); h[5] = (double)__uint_as_float(D.y); w[7] = (double)__uint_as_float(D.z); h[7] = (double)__uint_as_float(D.w);
            } else {
#pragma unroll
                for (int cc = 0; cc < 4; ++cc)
#pragma unroll
                    for (int l = 0; l < 2; ++l) {
                        const uint4 P = blk[l * 4 + cc];
                        w[2 * cc + l] = __longlong_as_double((long long)(((unsigned long long)P.y << 32) | P.x));
                        h[2 * cc + l] = __longlong_as_double((long long)(((unsigned long long)P.w << 32) | P.z));
                    }
            }
            const double wy0 = 1.0 - ty, wx0 = 1.0 - tx, wz0 = 1.0 - tz;
            const double a00 = wy0 * wx0, a01 = wy0 * tx, a10 = ty * wx0, a11 = ty * tx;
            const double k0 = a00 * wz0, k1 = a00 * tz, k2 = a01 * wz0, k3 = a01 * tz;
            const double k4 = a10 * wz0, k5 = a10 * tz, k6 = a11 * wz0, k7 = a11 * tz;
            sw = 0.0; sh = 0.0;
            sw += w[0] * k0; sh += h[0] * k0;
            sw += w[1] * k1; sh += h[1] * k1;
            sw += w[2] * k2; sh += h[2] * k2;
            sw += w[3] * k3; sh += h[3] * k3;
            sw += w[4] * k4; sh += h[4] * k4;
            sw += w[5] * k5; sh += h[5] * k5;
            sw += w[6] * k6; sh += h[6] * k6;
            sw += w[7] * k7; sh += h[7] * k7;
        }
        Q.store(i, sw, sh, wet, hyd);
    }
}

constexpr int BUILD_STAGE_BYTES = 32 << 10;   // LDS staging area of build_cube_kernel: (2 levels x heights per round) x footprint columns x 16 B (8 B for f32 cubes)
constexpr int BUILD_NCOL_MAX = 256;           // a tile footprint of more cube columns than this takes the direct loads

// _build_cube (delay.py:196-216): points generated on the fly from (xpts, ypts, zpts); the arithmetic per point is that of
// trilinear<> (scipy's weight order, _rgi.py:490-498), so the values are the same bit for bit.  Output (nz, ny, nx).  Two kernels:
//
// build_cube_setup_kernel - everything that depends on ONE coordinate only, once: per output node (iy, ix) the model-CRS projection
//   (LCC / polar stereographic cubes), its x / y cell and the two weights; per output height its z cell and weight.  24 B per node
//   and 16 B per height in a scratch buffer.  (Fused into the gather kernel this code - bisections, four fp64 divisions, the
//   projection's libm calls - set that kernel's register allocation and spilled into its loop.)
// build_cube_kernel - the gather.  What bounds it is neither HBM nor arithmetic but the EIGHT 16 B corner loads per point through the
//   vector L1's 64 B/clk/CU return path (config 2, f64 cube: 128 B per point = 0.15 ms by themselves, 0.27 ms with their latency
//   exposed; tools/probes/buildcube_probe.hip), while the output alone can be written in 0.10-0.115 ms (tools/probes/write_probe.hip).
//   A workgroup takes a 64 x 4-node TILE of the output grid (a wave = 64 consecutive nodes of one row: 512 B per store); the nodes
//   of a tile interpolate from a handful of cube columns (config 2: 3 x 12 of them for 256 nodes), so the tile's FOOTPRINT - columns
//   [cy_min, cy_max + 1] x [cx_min, cx_max + 1], two LDS min / max atomics per thread - is staged in LDS for as many heights as fit
//   (2 level slots per height x columns; usually the whole chunk in one round) and the corners are read from there at the LDS rate.
//   Same operands, same arithmetic order: bit-identical values.  A tile whose footprint does not fit (an output grid much coarser
//   than the model's) takes the direct loads, batched ahead of the stores.
struct BuildNode { int cy, cx; double ty, tx; };      // cy < 0: the node lies outside the cube's x / y range (or is NaN) -> fill value
struct BuildLevel { double tz; int cz; int pad; };    // cz < 0: the height lies outside the z axis -> fill value

template <typename T2>
__global__ __launch_bounds__(256) void build_cube_setup_kernel(CubeView<T2> c, LccParams proj, const double* __restrict__ xpts, int64_t nx,
                                                               const double* __restrict__ ypts, int64_t ny, const double* __restrict__ zpts, int64_t nz,
                                                               BuildNode* __restrict__ nodes_out, BuildLevel* __restrict__ levels_out, int axes_in_lds) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    const double* s_y = c.axes;                       // very long axes stay in global memory (L1 / L2 hits)
    if (axes_in_lds) {
        double* t = reinterpret_cast<double*>(smem_raw);
        for (int i = threadIdx.x; i < c.ny + c.nx + c.nz; i += blockDim.x) t[i] = c.axes[i];
        __syncthreads();
        s_y = t;
    }
    const double* s_x = s_y + c.ny;
    const double* s_z = s_x + c.nx;
    const int64_t nodes = nx * ny;
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < nodes + nz; i += (int64_t)gridDim.x * blockDim.x) {
        if (i >= nodes) {                             // the heights: the same for every node
            const double z = zpts[i - nodes];
            BuildLevel L; L.cz = -1; L.tz = 0.0; L.pad = 0;
            if ((z >= c.z_lo) && (z <= c.z_hi)) {
                L.cz = find_cell(s_z, c.nz, z, c.z_lo, c.inv_dz, c.uni_z);
                L.tz = (z - s_z[L.cz]) / (s_z[L.cz + 1] - s_z[L.cz]);
            }
            levels_out[i - nodes] = L;
            continue;
        }
        const int64_t ix = i % nx, iy = i / nx;
        double qy = ypts[iy], qx = xpts[ix];
        if (proj.kind == 1) { double px_, py_; lcc_forward(proj, qy, qx, px_, py_); qx = px_; qy = py_; }   // transformPoints, delay.py:207-209
        BuildNode N; N.cy = -1; N.cx = 0; N.ty = 0.0; N.tx = 0.0;
        if ((qy >= c.y_lo) && (qy <= c.y_hi) && (qx >= c.x_lo) && (qx <= c.x_hi)) {
            N.cy = find_cell(s_y, c.ny, qy, c.y_lo, c.inv_dy, c.uni_y);
            N.cx = find_cell(s_x, c.nx, qx, c.x_lo, c.inv_dx, c.uni_x);
            N.ty = (qy - s_y[N.cy]) / (s_y[N.cy + 1] - s_y[N.cy]);
            N.tx = (qx - s_x[N.cx]) / (s_x[N.cx + 1] - s_x[N.cx]);
        }
        nodes_out[i] = N;
    }
}

template <typename T2>
__global__ __launch_bounds__(256) void build_cube_kernel(const T2* __restrict__ cv, int cny, int cnx, int cnz, const BuildNode* __restrict__ nrec,
                                                         const BuildLevel* __restrict__ lrec, int64_t nx, int64_t ny, int64_t nz, int64_t zchunk,
                                                         double* __restrict__ wet, double* __restrict__ hyd) {
    extern __shared__ __attribute__((aligned(16))) unsigned char smem_raw[];
    constexpr int STAGE_ELEMS = BUILD_STAGE_BYTES / (int)sizeof(T2);
    T2* s_stage = reinterpret_cast<T2*>(smem_raw);
    double* s_tz = reinterpret_cast<double*>(smem_raw + BUILD_STAGE_BYTES);
    int* s_cz = reinterpret_cast<int*>(s_tz + zchunk);
    int* s_fp = s_cz + zchunk;                                                 // cy_min, cy_max, cx_min, cx_max, "a height outside the z axis"
    const int64_t nodes = nx * ny;
    const int64_t z0 = (int64_t)blockIdx.y * zchunk, z1 = min(z0 + zchunk, nz);
    const int nzc = (int)(z1 - z0);
    if (threadIdx.x == 0) s_fp[4] = 0;
    __syncthreads();
    for (int k = threadIdx.x; k < nzc; k += blockDim.x) {
        const BuildLevel L = lrec[z0 + k];
        s_tz[k] = L.tz; s_cz[k] = L.cz;
        if (L.cz < 0) s_fp[4] = 1;
    }
    __syncthreads();
    const bool all_z_inside = s_fp[4] == 0;
    const int64_t tiles_x = (nx + 63) / 64, tiles_y = (ny + 3) / 4;
    for (int64_t t = blockIdx.x; t < tiles_x * tiles_y; t += gridDim.x) {       // (workgroup-uniform trip count: barriers inside)
        const int64_t ix = (t % tiles_x) * 64 + (threadIdx.x & 63), iy = (t / tiles_x) * 4 + (threadIdx.x >> 6);
        const bool act = ix < nx && iy < ny;
        const int64_t i = iy * nx + ix;
        BuildNode N; N.cy = -1; N.cx = 0; N.ty = 0.0; N.tx = 0.0;
        if (act) N = nrec[i];
        const bool in_xy = N.cy >= 0;
        const int cy = max(N.cy, 0), cx = N.cx;
        const double wy0 = 1.0 - N.ty, wx0 = 1.0 - N.tx;
        const double a00 = wy0 * wx0, a01 = wy0 * N.tx, a10 = N.ty * wx0, a11 = N.ty * N.tx;
        if (threadIdx.x == 0) { s_fp[0] = 0x7fffffff; s_fp[1] = -1; s_fp[2] = 0x7fffffff; s_fp[3] = -1; }
        __syncthreads();                               // (also: the previous tile's readers are done with the staging area)
        if (in_xy) { atomicMin(&s_fp[0], cy); atomicMax(&s_fp[1], cy); atomicMin(&s_fp[2], cx); atomicMax(&s_fp[3], cx); }
        __syncthreads();
        const int fy0 = s_fp[0], fy1 = s_fp[1], fx0 = s_fp[2], fx1 = s_fp[3];
        const int nfx = fx1 - fx0 + 2, ncol = fy1 >= 0 ? (fy1 - fy0 + 2) * nfx : 0;       // (no node inside the cube: nothing to stage)
        const bool staged = ncol > 0 && ncol <= BUILD_NCOL_MAX && 2 * ncol <= STAGE_ELEMS;    // workgroup-uniform
        const int U = staged ? max(1, min(nzc, STAGE_ELEMS / (2 * ncol))) : 1;                // heights per staging round (often the whole chunk)
        const int lc = in_xy ? (cy - fy0) * nfx + (cx - fx0) : 0;                         // this node's (y0, x0) column in the staging area
        // one output point from its eight corners.  CLEAN: every lane of the wave is an output node inside the cube's x / y range
        // and every height of the chunk lies inside the z axis (the usual case) - no validity selects, no store guards
        auto emit = [&](auto clean, int k, const T2* v) {
            constexpr bool CLEAN = decltype(clean)::value;
            const double tz = s_tz[k];
            const double wz0 = 1.0 - tz;
            const double k0 = a00 * wz0, k1 = a00 * tz, k2 = a01 * wz0, k3 = a01 * tz;
            const double k4 = a10 * wz0, k5 = a10 * tz, k6 = a11 * wz0, k7 = a11 * tz;
            double sw = 0.0, sh = 0.0;
            sw += (double)v[0].x * k0; sh += (double)v[0].y * k0;
            sw += (double)v[1].x * k1; sh += (double)v[1].y * k1;
            sw += (double)v[2].x * k2; sh += (double)v[2].y * k2;
            sw += (double)v[3].x * k3; sh += (double)v[3].y * k3;
            sw += (double)v[4].x * k4; sh += (double)v[4].y * k4;
            sw += (double)v[5].x * k5; sh += (double)v[5].y * k5;
            sw += (double)v[6].x * k6; sh += (double)v[6].y * k6;
            sw += (double)v[7].x * k7; sh += (double)v[7].y * k7;
            if (!CLEAN) {
                // (a select, not a branch around the arithmetic: under `if (valid)` the compiler sinks a height's loads into the branch)
                const bool ok = in_xy && s_cz[k] >= 0;
                sw = ok ? sw : qnan(); sh = ok ? sh : qnan();
            }
            if (CLEAN || act) {
                const int64_t o = (z0 + k) * nodes + i;
                __builtin_nontemporal_store(sw, wet + o);
                __builtin_nontemporal_store(sh, hyd + o);
            }
        };
        const bool wave_clean = __all(act && in_xy) && all_z_inside;
        if (staged) {
            for (int kb = 0; kb < nzc; kb += U) {
                if (kb > 0) __syncthreads();           // the previous round has been read
                // lane -> footprint column (one integer division per column and thread), wave -> every fourth level slot
                const int nu = min(U, nzc - kb);
                for (int col = threadIdx.x & 63; col < ncol; col += 64) {
                    const int fy = col / nfx, fx = col - fy * nfx;
                    // (rows / columns beyond the cube's last node belong to no node's corners: clamped, never read)
                    const T2* colp = cv + ((int64_t)min(fy0 + fy, cny - 1) * cnx + min(fx0 + fx, cnx - 1)) * cnz;
                    for (int slot = threadIdx.x >> 6; slot < 2 * nu; slot += 4)
                        s_stage[slot * ncol + col] = colp[max(s_cz[kb + (slot >> 1)], 0) + (slot & 1)];
                }
                __syncthreads();
                auto round = [&](auto clean) {
                    const T2* lo = s_stage + lc;
#pragma unroll 2
                    for (int u = 0; u < nu; ++u, lo += 2 * ncol) {
                        const T2* hi = lo + ncol;
                        T2 v[8];
                        v[0] = lo[0]; v[1] = hi[0];
                        v[2] = lo[1]; v[3] = hi[1];
                        v[4] = lo[nfx]; v[5] = hi[nfx];
                        v[6] = lo[nfx + 1]; v[7] = hi[nfx + 1];
                        emit(clean, kb + u, v);
                    }
                };
                if (wave_clean) round(std::true_type{}); else round(std::false_type{});
            }
        } else {
            // direct loads: UD heights per trip, their 8 UD corner loads issued back to back ahead of the first store
            constexpr int UD = sizeof(T2) == 8 ? 4 : 2;
            const T2* col00 = cv + ((int64_t)cy * cnx + cx) * cnz;             // column (y0, x0); the others at fixed strides
            const T2* col01 = col00 + cnz;
            const T2* col10 = col00 + (int64_t)cnx * cnz;
            const T2* col11 = col10 + cnz;
            for (int kb = 0; kb < nzc; kb += UD) {
                T2 v[UD][8];
#pragma unroll
                for (int u = 0; u < UD; ++u) {
                    const int cz = max(s_cz[min(kb + u, nzc - 1)], 0);
                    v[u][0] = col00[cz]; v[u][1] = col00[cz + 1];
                    v[u][2] = col01[cz]; v[u][3] = col01[cz + 1];
                    v[u][4] = col10[cz]; v[u][5] = col10[cz + 1];
                    v[u][6] = col11[cz]; v[u][7] = col11[cz + 1];
                }
                asm volatile("" ::: "memory");      // (the compiler sank the later heights' loads behind the first store otherwise)
#pragma unroll
                for (int u = 0; u < UD; ++u) {
                    if (kb + u >= nzc) break;
                    emit(std::false_type{}, kb + u, v[u]);
                }
            }
        }
    }
}

constexpr int BUILD_ZCHUNK_MAX = 1024;

// Conventional.__call__ tail (losreader.py:130-133): delay / LOS_enu[..., -1].  pmode as in project_divisor (1: inc[i] deg, 2: inc0, 3:
// the divisor itself); either field may be absent (the reference projects wet and hydro in two calls).
__global__ void project_kernel(double* wet, double* hyd, int pmode, const double* __restrict__ proj, double inc0, int64_t n) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double up = project_divisor(pmode, proj, inc0, i);   // inc_hd_to_enu(...)[..., -1] = cosd(inc)
        if (wet) wet[i] = wet[i] / up;
        if (hyd) hyd[i] = hyd[i] / up;
    }
}

__global__ void lcc_kernel(LccParams proj, const double* __restrict__ lat, const double* __restrict__ lon, int64_t n,
                           double* __restrict__ y, double* __restrict__ x) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double px, py; lcc_forward(proj, lat[i], lon[i], px, py);
        x[i] = px; y[i] = py;
    }
}

__global__ void lla2ecef_kernel(const double* __restrict__ lat, const double* __restrict__ lon, const double* __restrict__ h,
                                int64_t n, double* __restrict__ xyz) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double x, y, z; lla2ecef(lat[i], lon[i], h[i], x, y, z);
        xyz[3 * i] = x; xyz[3 * i + 1] = y; xyz[3 * i + 2] = z;
    }
}

__global__ void ecef2lla_kernel(const double* __restrict__ xyz, int64_t n, double* __restrict__ lon, double* __restrict__ lat,
                                double* __restrict__ h) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double lo, la, hh; ecef2lla(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], lo, la, hh);
        lon[i] = lo; lat[i] = la; h[i] = hh;
    }
}

__global__ void look_kernel(RayParams P, double* __restrict__ los) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < P.n; i += (int64_t)gridDim.x * blockDim.x) {
        double lat, lon;
        if (P.origin_mode == 0) { lat = P.ypts[i / P.nx]; lon = P.xpts[i % P.nx]; }
        else { lat = P.lat[i]; lon = P.lon[i]; }
        double u, v, w;
        if (P.los_mode == 0) { u = P.los[3 * i]; v = P.los[3 * i + 1]; w = P.los[3 * i + 2]; }
        else if (P.los_mode == 1) inc_hd_to_ecef(P.inc[i], P.hd ? P.hd[i] : P.hd0, lat, lon, u, v, w);
        else if (P.los_mode == 2) inc_hd_to_ecef(P.inc0, P.hd0, lat, lon, u, v, w);
        else {
            double sla, cla, slo, clo;
            sincos(lat * DEG_TO_RAD, &sla, &cla); sincos(lon * DEG_TO_RAD, &slo, &clo);
            u = cla * clo; v = cla * slo; w = sla;
        }
        los[3 * i] = u; los[3 * i + 1] = v; los[3 * i + 2] = w;
    }
}

__global__ void toa_kernel(const double* __restrict__ xyz, const double* __restrict__ los, int64_t n, double h,
                           const double* __restrict__ factor, double* __restrict__ pos) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double px, py, pz;
        toa_newton(xyz[3 * i], xyz[3 * i + 1], xyz[3 * i + 2], los[3 * i], los[3 * i + 1], los[3 * i + 2], h,
                   factor ? 3 : 10, factor ? factor[i] : 1.0, px, py, pz);
        pos[3 * i] = px; pos[3 * i + 1] = py; pos[3 * i + 2] = pz;
    }
}

// build_ray materialised (losreader.py:772-835): levels passed in a small device table
__global__ void build_ray_kernel(const double* __restrict__ xyz, const double* __restrict__ los, int64_t n, int K,
                                 const double* __restrict__ lo_hi, double* __restrict__ lengths, double* __restrict__ low,
                                 double* __restrict__ high) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        const double ox = xyz[3 * i], oy = xyz[3 * i + 1], oz = xyz[3 * i + 2];
        const double lx = los[3 * i], ly = los[3 * i + 1], lz = los[3 * i + 2];
        double hx = 0, hy = 0, hz = 0, cosf = 1.0;
        for (int k = 0; k < K; ++k) {
            const double lo = lo_hi[k], hi = lo_hi[K + k];
            double bx, by, bz;
            if (k == 0) toa_newton(ox, oy, oz, lx, ly, lz, lo, 10, 1.0, bx, by, bz);
            else { bx = hx; by = hy; bz = hz; }
            toa_newton(ox, oy, oz, lx, ly, lz, hi, k == 0 ? 10 : 3, cosf, hx, hy, hz);
            const double dx = hx - bx, dy = hy - by, dz = hz - bz;
            const double L = sqrt(dx * dx + dy * dy + dz * dz);
            if (k == 0) cosf = (hi - lo) / L;
            lengths[(int64_t)k * n + i] = L;
            double* pl = low + ((int64_t)k * n + i) * 3; pl[0] = bx; pl[1] = by; pl[2] = bz;
            double* ph = high + ((int64_t)k * n + i) * 3; ph[0] = hx; ph[1] = hy; ph[2] = hz;
        }
    }
}
