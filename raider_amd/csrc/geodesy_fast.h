// Device-only "light fp64" geodesy for the ray kernels (gfx950).
//
// Same quantities as geodesy.h (PROJ `cart` inverse: single-pass Bowring latitude, ellipsoidal height, longitude),
// evaluated with cheaper instruction sequences:
//   * 1/sqrt comes from v_rsq_f64 + ONE Newton-Raphson step (4e-15 relative; the seed alone is 2^-24) instead of the
//     correctly rounded sqrt()/div expansions (each ~4x the instructions);
//   * the height uses the support-function identity h = p cos(phi) + z sin(phi) - a sqrt(1 - es sin^2(phi)): exact
//     for the exact latitude, second-order in the latitude error, no division; sqrt(1-u), u <= es, is a 6-term
//     polynomial;
//   * latitude / longitude of a ray sample are computed RELATIVE to the ray origin (whose geodetic coordinates are
//     the kernel's exact inputs): sin(phi-phi0) and sin(lam-lam0) come from 2 FMAs each and the angle from an odd
//     asin series (rays that could travel more than 0.03 rad, or come within 2 deg of the +-180 meridian, are
//     classified "slow" up front and handled by the generic-geodesy kernels).  No atan/atan2 per sample.
// Accuracy is pinned by tests/test_gpu_parity.py (ray-traced delays against the CPU restatement of the reference:
// 1e-9 m across latitudes, hemispheres, the dateline and a polar scene).
#pragma once
#include "geodesy.h"

namespace rdr {

template <int NR>
__device__ __forceinline__ double rsq_nr(double a) {
    double y = __builtin_amdgcn_rsq(a);
#pragma unroll
    for (int i = 0; i < NR; ++i) {
        const double t = a * y;
        const double e = fma(-t, y, 1.0);
        y = fma(0.5 * y, e, y);
    }
    return y;
}

constexpr double E2S_B = WGS84_E2S * WGS84_B;
constexpr double ES_A = WGS84_ES * WGS84_A;

// sqrt(1-u) for 0 <= u <= es (0.0067): degree-6 Taylor, truncation 1e-17
__device__ __forceinline__ double sqrt_one_minus_small(double u) {
    double q = fma(u, -21.0 / 1024.0, -7.0 / 256.0);
    q = fma(u, q, -5.0 / 128.0);
    q = fma(u, q, -1.0 / 16.0);
    q = fma(u, q, -1.0 / 8.0);
    q = fma(u, q, -0.5);
    return fma(u, q, 1.0);
}

struct GeoF {
    double p, rp;          // hypot(x,y) and its reciprocal
    double cphi, sphi;     // cos / sin of the (Bowring) geodetic latitude, unit-normalised
    double h;              // ellipsoidal height
    bool regular;          // false -> caller must use the generic path (poles, Earth's centre, NaN)
};

// Height is evaluated as the support-function identity  h = p cos(phi) + z sin(phi) - a sqrt(1 - es sin^2(phi)),
// which is EXACT for the exact latitude and only second-order sensitive to the latitude error (PROJ's
// h = p/cos(phi) - N is first-order sensitive: with the single-pass Bowring latitude it is off by up to 1.7e-5 m
// at 40 km, 1e-6 m at 10 km - far below anything that reaches the delay: < 1e-9 m).
// PRECISE_PHI: refine the theta-stage normalisation (needed when the LATITUDE itself is used, i.e. for samples;
// the height alone tolerates the 2^-24 seed).
template <bool PRECISE_PHI>
__device__ __forceinline__ GeoF geo_fast(double x, double y, double z) {
    GeoF g;
    const double p2 = fma(x, x, y * y);
    g.rp = rsq_nr<1>(p2);
    g.p = p2 * g.rp;
    const double xt = g.p * WGS84_B, yt = z * WGS84_A;
    const double n2t = fma(xt, xt, yt * yt);
    const double rn = PRECISE_PHI ? rsq_nr<1>(n2t) : __builtin_amdgcn_rsq(n2t);
    const double c = xt * rn, s = yt * rn;
    const double y_phi = fma(E2S_B, s * s * s, z);
    const double x_phi = fma(-ES_A, c * c * c, g.p);
    const double r = rsq_nr<1>(fma(x_phi, x_phi, y_phi * y_phi));
    g.sphi = y_phi * r; g.cphi = x_phi * r;
    const double sq = sqrt_one_minus_small(WGS84_ES * g.sphi * g.sphi);
    g.h = fma(g.p, g.cphi, fma(z, g.sphi, -WGS84_A * sq));
    g.regular = (g.cphi >= 1e-6) && (p2 > 1.0);
    return g;
}

// asin(s) for the small angles between a fit node and the ray origin: odd series through s^11.  The static classification
// admits only rays that turn by less than LON_TRAVEL_MAX = 0.2 rad in longitude (0.035 rad in latitude), where the truncation
// (231/13312 s^13) is < 1.4e-11 rad (9e-5 m on the ground at the far end of the longest admitted ray).  Used only at the six
// fit nodes of a ray, never per sample.
__device__ __forceinline__ double asin_small(double s) {
    const double s2 = s * s;
    double q = fma(s2, 63.0 / 2816.0, 35.0 / 1152.0);
    q = fma(s2, q, 5.0 / 112.0);
    q = fma(s2, q, 3.0 / 40.0);
    q = fma(s2, q, 1.0 / 6.0);
    return fma(s * s2, q, s);
}

// Geodetic frame of the ray origin: everything the per-sample delta formulas need.
struct RayBase {
    double lat0, lon0;     // degrees (kernel inputs, exact)
    double s0, c0;         // sin/cos(lat0)
    double sl0, cl0;       // sin/cos(lon0)
};

__device__ __forceinline__ RayBase make_base(double lat_deg, double lon_deg) {
    RayBase b;
    b.lat0 = lat_deg; b.lon0 = lon_deg;
    sincos(lat_deg * DEG_TO_RAD, &b.s0, &b.c0);
    sincos(lon_deg * DEG_TO_RAD, &b.sl0, &b.cl0);
    return b;
}

// ---- ray polynomials ------------------------------------------------------------------------------------------------
// Along a straight ray o + t l the geodetic height, latitude and longitude are smooth functions of t: over the rays the
// static classification admits (angular travel < 0.035 rad, < 0.2 rad of longitude, away from the poles) their degree-5
// interpolants at the 6 Chebyshev nodes of the ray's parameter range reproduce them to < 3e-7 m (h) and < 2.2e-5 m on the
// ground (lat, lon) in the worst admitted case, and to 5e-9 m / 5e-6 m for rays shorter than 100 km
// (sweep: tools/ray_poly_probe.py).  The ray kernels therefore evaluate the full geodesy 6 times per ray and replace
// every later evaluation (3 per model level in the Newton level crossings, 1 per integration sample) by 5 FMAs per
// quantity.  u = su * t + ou maps the range to [-1, 1]; coefficients are monomial in u (well conditioned on [-1, 1]).
constexpr int PN = 6;
struct RayPoly { double h[PN], lat[PN], lon[PN]; };

__device__ __forceinline__ double poly5(const double* c, double u) {
    double r = fma(c[5], u, c[4]);
    r = fma(r, u, c[3]);
    r = fma(r, u, c[2]);
    r = fma(r, u, c[1]);
    return fma(r, u, c[0]);
}

// ECEF -> (lon - lon0, lat - lat0) in degrees and h of a point within 0.03 rad of the ray origin (guaranteed by the per-ray
// static classification in crossings_kernel; rays that fail it never come here): sin(dphi), sin(dlam) from 2 FMAs each
__device__ __forceinline__ void ecef2lla_delta(const RayBase& b, double x, double y, double z, double& dlon, double& dlat, double& h) {
    const GeoF g = geo_fast<true>(x, y, z);
    const double sd = fma(g.sphi, b.c0, -g.cphi * b.s0);
    const double sl = fma(b.cl0, y, -b.sl0 * x) * g.rp;
    dlat = asin_small(sd) * RAD_TO_DEG;
    dlon = asin_small(sl) * RAD_TO_DEG;
    h = g.h;
}

// The same point projected onto a SPHERICAL Lambert conformal conic grid (HRRR: e = 0), RELATIVE to the projection of the
// ray origin (done once per ray with the full lcc_forward): with t = tan(pi/4 - phi/2) = cos(phi) / (1 + sin(phi)),
//   rho / rho_origin = (t / t_origin)^n = exp(n D),  D = log(t / t_origin) = 2 atanh(z),  z = (t - t_o) / (t + t_o)
//   theta - theta_origin = n (lam - lam_origin)
// and both D and the angle are small for a node of an admitted ray (|z| < 0.05 by the LCC clause of the static
// classification, |n dlam| < 0.2), so short series replace log, exp and sincos: atanh through z^9 (next term 9e-15
// relative), exp through x^9 (3e-17), sin / cos through x^11 / x^12 (< 1e-19).  No libm call per node - which is what lets
// the LCC instantiation of crossings_kernel keep the register budget of the lon/lat one.
struct LccOrigin { double rho, st, ct; };      // rho_origin and sin / cos of theta_origin (either sign convention of rho works:
                                               // only rho * st and rho * ct are used)

__device__ __forceinline__ void ecef2lcc_sphere(const RayBase& b, const LccParams& L, const LccOrigin& o, double x, double y, double z,
                                                double& dpx, double& dpy, double& h) {
    const GeoF g = geo_fast<true>(x, y, z);
    const double sl = fma(b.cl0, y, -b.sl0 * x) * g.rp;
    const double ang = L.n * asin_small(sl);                                   // theta - theta_origin
    // z = (t - t_o) / (t + t_o) with the common denominator (1 + sin phi)(1 + sin phi_o) removed
    const double ta = g.cphi * (1.0 + b.s0), tb = b.c0 * (1.0 + g.sphi);
    const double den = ta + tb;
    double iden = __builtin_amdgcn_rcp(den);
    iden = fma(fma(-den, iden, 1.0), iden, iden);
    iden = fma(fma(-den, iden, 1.0), iden, iden);
    const double zz = (ta - tb) * iden, z2 = zz * zz;
    double q = fma(z2, 1.0 / 9.0, 1.0 / 7.0);
    q = fma(z2, q, 1.0 / 5.0);
    q = fma(z2, q, 1.0 / 3.0);
    const double xe = L.n * (2.0 * fma(zz * z2, q, zz));                       // n log(t / t_o)
    double e = fma(xe, 1.0 / 362880.0, 1.0 / 40320.0);
    e = fma(xe, e, 1.0 / 5040.0); e = fma(xe, e, 1.0 / 720.0); e = fma(xe, e, 1.0 / 120.0); e = fma(xe, e, 1.0 / 24.0);
    e = fma(xe, e, 1.0 / 6.0); e = fma(xe, e, 0.5); e = fma(xe, e, 1.0);
    const double em1 = xe * e;                                                 // exp(xe) - 1
    const double a2 = ang * ang;
    double sn = fma(a2, -1.0 / 39916800.0, 1.0 / 362880.0);
    sn = fma(a2, sn, -1.0 / 5040.0); sn = fma(a2, sn, 1.0 / 120.0); sn = fma(a2, sn, -1.0 / 6.0);
    sn = fma(ang * a2, sn, ang);                                               // sin(ang)
    double cm = fma(a2, 1.0 / 479001600.0, -1.0 / 3628800.0);
    cm = fma(a2, cm, 1.0 / 40320.0); cm = fma(a2, cm, -1.0 / 720.0); cm = fma(a2, cm, 1.0 / 24.0); cm = fma(a2, cm, -0.5);
    cm = a2 * cm;                                                              // cos(ang) - 1
    // rho sin(theta) - rho_o sin(theta_o) and rho cos(theta) - rho_o cos(theta_o), kept as DIFFERENCES (no cancellation):
    //   rho/rho_o = 1 + em1,  sin(theta) = st (1 + cm) + ct sn,  cos(theta) = ct (1 + cm) - st sn
    const double ds = fma(o.st, cm, o.ct * sn), dc = fma(o.ct, cm, -o.st * sn);
    dpx = o.rho * fma(em1, o.st + ds, ds);                                     //  x - x_origin
    dpy = -o.rho * fma(em1, o.ct + dc, dc);                                    //  y - y_origin
    h = g.h;
}

// Interpolant through the Chebyshev nodes u_j = cos(pi (2j+1)/12) of t in [mid - half, mid + half]: coefficient n =
// sum_j VINV[n][j] f(u_j) (inverse Vandermonde matrix of the nodes).  One node at a time (rolled loop) to keep the
// register footprint of this once-per-ray step below that of the per-level loop.
__device__ const double RAY_POLY_NODES[PN] = {0.9659258262890682867, 0.7071067811865475244, 0.2588190451025207623,
                                              -0.2588190451025207623, -0.7071067811865475244, -0.9659258262890682867};
__device__ const double RAY_POLY_VINV[PN][PN] = {   // [node j][power n]
    {0.04465819873852045108, 0.04623356941400984176, -0.7559830641437075688, -0.7826512591014083831, 1.333333333333333333, 1.380368240546777399},
    {-0.1666666666666666667, -0.2357022603955158415, 2.666666666666666667, 3.771236166328253463, -2.666666666666666667, -3.771236166328253463},
    {0.6220084679281462156, 2.403256173369168256, -1.910683602522959098, -7.382314550175851944, 1.333333333333333333, 5.151604406875030863},
    {0.6220084679281462156, -2.403256173369168256, -1.910683602522959098, 7.382314550175851944, 1.333333333333333333, -5.151604406875030863},
    {-0.1666666666666666667, 0.2357022603955158415, 2.666666666666666667, -3.771236166328253463, -2.666666666666666667, 3.771236166328253463},
    {0.04465819873852045108, -0.04623356941400984176, -0.7559830641437075688, 0.7826512591014083831, 1.333333333333333333, -1.380368240546777399}};

// proj.kind == 1: the cube lives on a Lambert-conformal-conic grid (HRRR) - the "lat" / "lon" polynomials are then fitted to
// the projected northing / easting of the node points (ecef_to_model, delay.py:253,295), which are just as smooth along a ray.
// Projection of a ray origin onto a spherical cone: rho = aF t^n with t = cos(phi) / (1 + sin(phi)) from the origin's own sine /
// cosine - one log and one exp per LATITUDE - and sin / cos of theta = n (lam - lam0) - one sincos per LONGITUDE.  On a GRID batch
// a tile has 16 of each: crossings_kernel computes them once per tile and shares them through LDS (lcc_rho / lcc_theta).
__device__ __forceinline__ double lcc_rho(const LccParams& proj, double s0, double c0) { return proj.aF * exp(proj.n * log(c0 / (1.0 + s0))); }
__device__ __forceinline__ void lcc_theta(const LccParams& proj, double lon_deg, double& st, double& ct) {
    double dlam = lon_deg * DEG_TO_RAD - proj.lam0;
    if (dlam > 3.141592653589793) dlam -= 6.283185307179586;
    else if (dlam < -3.141592653589793) dlam += 6.283185307179586;
    sincos(proj.n * dlam, &st, &ct);
}

template <bool LCC>
__device__ __forceinline__ void fit_ray_poly(const RayBase& b, double ox, double oy, double oz, double lx, double ly, double lz,
                                             double mid, double half, const LccParams& proj, const LccOrigin* shared_org, RayPoly& q) {
    double x0 = b.lon0, y0 = b.lat0;
    LccOrigin org = {0.0, 0.0, 0.0};
    if (LCC) {
        // Spherical cone (the static classification sends every ray of an ELLIPSOIDAL LCC cube to the generic kernels)
        if (shared_org) org = *shared_org;
        else { org.rho = lcc_rho(proj, b.s0, b.c0); lcc_theta(proj, b.lon0, org.st, org.ct); }
        x0 = fma(org.rho, org.st, proj.x0);
        y0 = proj.y0 + proj.rho0 - org.rho * org.ct;
    }
#pragma unroll
    for (int n = 0; n < PN; ++n) { q.h[n] = 0.0; q.lat[n] = 0.0; q.lon[n] = 0.0; }
#pragma unroll 1
    for (int j = 0; j < PN; ++j) {
        const double t = fma(half, RAY_POLY_NODES[j], mid);
        double dx, dy, h;
        if (LCC) ecef2lcc_sphere(b, proj, org, fma(t, lx, ox), fma(t, ly, oy), fma(t, lz, oz), dx, dy, h);
        else ecef2lla_delta(b, fma(t, lx, ox), fma(t, ly, oy), fma(t, lz, oz), dx, dy, h);
#pragma unroll
        for (int n = 0; n < PN; ++n) {
            const double v = RAY_POLY_VINV[j][n];
            q.h[n] = fma(v, h, q.h[n]); q.lat[n] = fma(v, dy, q.lat[n]); q.lon[n] = fma(v, dx, q.lon[n]);
        }
    }
    q.lat[0] += y0;
    q.lon[0] += x0;
}

}  // namespace rdr
