// WGS84 geodetic <-> ECEF arithmetic shared by every kernel (and by the host-side helpers).
//
// Replaces the reference's calls into pyproj/PROJ:
//   tools/RAiDER/utilFcns.py:77-88   lla2ecef / ecef2lla  (Transformer 4326<->4978, always_xy)
//   tools/RAiDER/delay.py:238,253,295  T / ecef_to_model inside _build_cube_ray
// PROJ is a third-party dependency that is not under /root/reference; the formulas are PROJ's
// published `cart` conversion (src/conversions/cart.cpp): forward = Heiskanen & Moritz 5-27,
// inverse = single-pass Bowring written with normalised (cos,sin) pairs instead of trig calls.
// The inverse needs no trig at all for the HEIGHT (what the Newton ray/level intersection needs),
// one atan for latitude and one atan2 for longitude.
#pragma once
#include <math.h>

#ifndef RDR_HD
#if defined(__HIPCC__)
#define RDR_HD __host__ __device__ __forceinline__
#else
#define RDR_HD inline
#endif
#endif

namespace rdr {

constexpr double WGS84_A = 6378137.0;
constexpr double WGS84_F = 1.0 / 298.257223563;
constexpr double WGS84_ES = 2.0 * WGS84_F - WGS84_F * WGS84_F;   // e^2
constexpr double WGS84_B = (1.0 - WGS84_F) * WGS84_A;
constexpr double WGS84_E2S = WGS84_ES / (1.0 - WGS84_ES);         // e'^2
constexpr double DEG_TO_RAD = 0.017453292519943296;
constexpr double RAD_TO_DEG = 57.295779513082321;

// geodetic (deg, deg, m) -> ECEF.  PROJ cart.cpp `cartesian()`.
RDR_HD void lla2ecef(double lat_deg, double lon_deg, double h, double& x, double& y, double& z) {
    const double lam = lon_deg * DEG_TO_RAD;
    const double phi = lat_deg * DEG_TO_RAD;
    double sphi, cphi, slam, clam;
    sincos(phi, &sphi, &cphi);
    sincos(lam, &slam, &clam);
    const double N = WGS84_A / sqrt(1.0 - WGS84_ES * sphi * sphi);
    x = (N + h) * cphi * clam;
    y = (N + h) * cphi * slam;
    z = (N * (1.0 - WGS84_ES) + h) * sphi;
}

// Shared front half of PROJ cart.cpp `geodetic()`: returns p, and (cos phi, sin phi) of the geodetic
// latitude plus the un-normalised (x_phi, y_phi) pair (for atan).  Polar / degenerate branches of the
// original are kept (x_phi <= 0 -> +-90 deg; cos phi < 1e-6 -> geocentric-radius height).
struct GeoFront {
    double p, cosphi, sinphi, x_phi, y_phi;
};

RDR_HD GeoFront geo_front(double x, double y, double z) {
    GeoFront g;
    g.p = sqrt(x * x + y * y);
    const double y_theta = z * WGS84_A;
    const double x_theta = g.p * WGS84_B;
    const double norm = sqrt(y_theta * y_theta + x_theta * x_theta);
    const double c = norm == 0 ? 1.0 : x_theta / norm;
    const double s = norm == 0 ? 0.0 : y_theta / norm;
    g.y_phi = z + WGS84_E2S * WGS84_B * s * s * s;
    g.x_phi = g.p - WGS84_ES * WGS84_A * c * c * c;
    const double norm_phi = sqrt(g.y_phi * g.y_phi + g.x_phi * g.x_phi);
    g.cosphi = norm_phi == 0 ? 1.0 : g.x_phi / norm_phi;
    g.sinphi = norm_phi == 0 ? 0.0 : g.y_phi / norm_phi;
    if (g.x_phi <= 0) {
        g.cosphi = 0;
        g.sinphi = z >= 0 ? 1.0 : -1.0;
    }
    return g;
}

RDR_HD double geo_height(const GeoFront& g, double z) {
    if (g.cosphi < 1e-6) {
        const double a2c = WGS84_A * WGS84_A * g.cosphi, b2s = WGS84_B * WGS84_B * g.sinphi;
        const double ac = WGS84_A * g.cosphi, bs = WGS84_B * g.sinphi;
        return fabs(z) - sqrt(a2c * a2c + b2s * b2s) / sqrt(ac * ac + bs * bs);
    }
    return g.p / g.cosphi - WGS84_A / sqrt(1.0 - WGS84_ES * g.sinphi * g.sinphi);
}

// ECEF -> ellipsoidal height only (the Newton iteration of getTopOfAtmosphere, losreader.py:729-731).
RDR_HD double ecef_height(double x, double y, double z) {
    const GeoFront g = geo_front(x, y, z);
    return geo_height(g, z);
}

// ECEF -> (lon deg, lat deg, h).
RDR_HD void ecef2lla(double x, double y, double z, double& lon_deg, double& lat_deg, double& h) {
    const GeoFront g = geo_front(x, y, z);
    double phi;
    if (g.x_phi <= 0) phi = z >= 0 ? 1.5707963267948966 : -1.5707963267948966;
    else phi = atan(g.y_phi / g.x_phi);
    lon_deg = atan2(y, x) * RAD_TO_DEG;
    lat_deg = phi * RAD_TO_DEG;
    h = geo_height(g, z);
}

// inc/heading (deg) -> local ENU unit vector (losreader.py:374-396) -> ECEF (utilFcns.py:91-121).
RDR_HD void inc_hd_to_ecef(double inc_deg, double hd_deg, double lat_deg, double lon_deg,
                           double& u, double& v, double& w) {
    double si, ci, sh, ch, sla, cla, slo, clo;
    sincos(inc_deg * DEG_TO_RAD, &si, &ci);
    sincos((hd_deg + 90.0) * DEG_TO_RAD, &sh, &ch);
    const double east = si * ch, north = si * sh, up = ci;
    sincos(lat_deg * DEG_TO_RAD, &sla, &cla);
    sincos(lon_deg * DEG_TO_RAD, &slo, &clo);
    const double t = cla * up - sla * north;
    w = sla * up + cla * north;
    u = clo * t - slo * east;
    v = slo * t + clo * east;
}

}  // namespace rdr
