// Transverse Mercator (UTM and friends) on the device - the pyproj step of transformPoints (delay.py:404-436) for output grids
// that are not lon/lat (tropo_delay(out_proj=EPSG:326xx): delay.py:207-209,259-263).  PROJ is not in /root/reference: this is the
// published Krueger series in the third flattening n to order n^6 (L. Krueger 1912; C. Karney, "Transverse Mercator with an
// accuracy of a few nanometers", J. Geodesy 85, 2011, eqs. 7-11, 35, 36), the same formulation PROJ's `etmerc` / `utm` uses:
// nanometre-level within 35 deg of the central meridian.  Part of libraider_hip.so (included by raider_hip.hip).  gfx950 only.
#pragma once
#include <hip/hip_runtime.h>
#include <math.h>
#include <stdint.h>

namespace rdr {

struct TmParams {
    double e, es;          // first eccentricity and its square
    double kA;             // k_0 * A, A = a / (1+n) (1 + n^2/4 + n^4/64 + n^6/256): the rectifying radius times the scale
    double lam0, x0, y0;   // central meridian (rad), false easting / northing (m)
    double xi0;            // xi of the latitude of origin on the central meridian
    double al[6], be[6];   // forward / inverse series coefficients
};

inline void tm_series(double n, double* al, double* be) {
    const double n2 = n * n, n3 = n2 * n, n4 = n3 * n, n5 = n4 * n, n6 = n5 * n;
    al[0] = n / 2 - 2 * n2 / 3 + 5 * n3 / 16 + 41 * n4 / 180 - 127 * n5 / 288 + 7891 * n6 / 37800;
    al[1] = 13 * n2 / 48 - 3 * n3 / 5 + 557 * n4 / 1440 + 281 * n5 / 630 - 1983433 * n6 / 1935360;
    al[2] = 61 * n3 / 240 - 103 * n4 / 140 + 15061 * n5 / 26880 + 167603 * n6 / 181440;
    al[3] = 49561 * n4 / 161280 - 179 * n5 / 168 + 6601661 * n6 / 7257600;
    al[4] = 34729 * n5 / 80640 - 3418889 * n6 / 1995840;
    al[5] = 212378941 * n6 / 319334400;
    be[0] = n / 2 - 2 * n2 / 3 + 37 * n3 / 96 - n4 / 360 - 81 * n5 / 512 + 96199 * n6 / 604800;
    be[1] = n2 / 48 + n3 / 15 - 437 * n4 / 1440 + 46 * n5 / 105 - 1118711 * n6 / 3870720;
    be[2] = 17 * n3 / 480 - 37 * n4 / 840 - 209 * n5 / 4480 + 5569 * n6 / 90720;
    be[3] = 4397 * n4 / 161280 - 11 * n5 / 504 - 830251 * n6 / 7257600;
    be[4] = 4583 * n5 / 161280 - 108847 * n6 / 3991680;
    be[5] = 20648693 * n6 / 638668800;
}

// (xi', eta') -> (xi, eta) or back: xi +- sum c_j sin(2j xi) cosh(2j eta), eta +- sum c_j cos(2j xi) sinh(2j eta)
__host__ __device__ inline void tm_apply(const double* c, double sign, double xi, double eta, double& oxi, double& oeta) {
    double sx = 0.0, se = 0.0;
    for (int j = 0; j < 6; ++j) {
        const double k = 2.0 * (j + 1);
        sx += c[j] * sin(k * xi) * cosh(k * eta);
        se += c[j] * cos(k * xi) * sinh(k * eta);
    }
    oxi = xi + sign * sx; oeta = eta + sign * se;
}

__host__ __device__ inline void tm_conformal(const TmParams& T, double phi, double lam, double& xip, double& etap) {
    const double tau = tan(phi);
    const double sig = sinh(T.e * atanh(T.e * tau / sqrt(1.0 + tau * tau)));
    const double taup = tau * sqrt(1.0 + sig * sig) - sig * sqrt(1.0 + tau * tau);
    const double cl = cos(lam);
    xip = atan2(taup, cl);
    etap = asinh(sin(lam) / sqrt(taup * taup + cl * cl));
}

inline TmParams tm_setup(double a, double es, double lat0_deg, double lon0_deg, double k0, double x0, double y0) {
    TmParams T;
    T.es = es; T.e = sqrt(es);
    const double f = 1.0 - sqrt(1.0 - es), n = f / (2.0 - f);
    const double n2 = n * n;
    T.kA = k0 * a / (1.0 + n) * (1.0 + n2 / 4 + n2 * n2 / 64 + n2 * n2 * n2 / 256);
    T.lam0 = lon0_deg * 0.017453292519943296; T.x0 = x0; T.y0 = y0;
    tm_series(n, T.al, T.be);
    double xip, etap, eta;
    tm_conformal(T, lat0_deg * 0.017453292519943296, 0.0, xip, etap);
    tm_apply(T.al, 1.0, xip, etap, T.xi0, eta);
    return T;
}

__host__ __device__ inline void tm_forward(const TmParams& T, double lat_deg, double lon_deg, double& x, double& y) {
    double lam = lon_deg * 0.017453292519943296 - T.lam0;
    if (lam > 3.141592653589793) lam -= 6.283185307179586;
    else if (lam < -3.141592653589793) lam += 6.283185307179586;
    double xip, etap, xi, eta;
    tm_conformal(T, lat_deg * 0.017453292519943296, lam, xip, etap);
    tm_apply(T.al, 1.0, xip, etap, xi, eta);
    x = T.x0 + T.kA * eta;
    y = T.y0 + T.kA * (xi - T.xi0);
}

__host__ __device__ inline void tm_inverse(const TmParams& T, double x, double y, double& lat_deg, double& lon_deg) {
    const double xi = (y - T.y0) / T.kA + T.xi0, eta = (x - T.x0) / T.kA;
    double xip, etap;
    tm_apply(T.be, -1.0, xi, eta, xip, etap);
    const double sh = sinh(etap), cx = cos(xip);
    const double taup = sin(xip) / sqrt(sh * sh + cx * cx);
    const double lam = atan2(sh, cx);
    // tau from tau' (conformal -> geodetic latitude): Newton on tau'(tau) (Karney 2011 eqs. 19-21); converges in 2-3 steps
    double tau = taup;
    const double e2m = 1.0 - T.es;
    for (int it = 0; it < 6; ++it) {
        const double t1 = sqrt(1.0 + tau * tau);
        const double sig = sinh(T.e * atanh(T.e * tau / t1));
        const double tpi = tau * sqrt(1.0 + sig * sig) - sig * t1;
        const double dt = (taup - tpi) / sqrt(1.0 + tpi * tpi) * (1.0 + e2m * tau * tau) / (e2m * t1);
        tau += dt;
        if (fabs(dt) < 1e-15 * fmax(1.0, fabs(tau))) break;
    }
    lat_deg = atan(tau) * 57.295779513082321;
    lon_deg = (lam + T.lam0) * 57.295779513082321;
    if (lon_deg > 180.0) lon_deg -= 360.0;
    else if (lon_deg < -180.0) lon_deg += 360.0;
}

// dir 0: (lat, lon) deg -> (y, x) m;  dir 1: (y, x) m -> (lat, lon) deg.   (argument order as transformPoints stacks them)
__global__ __launch_bounds__(256) void tm_kernel(TmParams T, int dir, const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                 double* __restrict__ oa, double* __restrict__ ob) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double u, v;
        if (dir == 0) { tm_forward(T, a[i], b[i], v, u); oa[i] = u; ob[i] = v; }       // u = y, v = x
        else { tm_inverse(T, b[i], a[i], u, v); oa[i] = u; ob[i] = v; }                 // u = lat, v = lon
    }
}

// The same for the two cones (LCC, polar stereographic): dir 0: (lat, lon) deg -> (y, x) m;  dir 1: (y, x) m -> (lat, lon) deg.
__global__ __launch_bounds__(256) void cone_kernel(LccParams L, int dir, const double* __restrict__ a, const double* __restrict__ b, int64_t n,
                                                   double* __restrict__ oa, double* __restrict__ ob) {
    for (int64_t i = blockIdx.x * (int64_t)blockDim.x + threadIdx.x; i < n; i += (int64_t)gridDim.x * blockDim.x) {
        double u, v;
        if (dir == 0) { lcc_forward(L, a[i], b[i], v, u); oa[i] = u; ob[i] = v; }      // u = y, v = x
        else { lcc_inverse(L, b[i], a[i], u, v); oa[i] = u; ob[i] = v; }                // u = lat, v = lon
    }
}

}  // namespace rdr
