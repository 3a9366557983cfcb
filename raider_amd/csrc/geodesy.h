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

// inc/heading (deg) -> local ENU unit vector (losreader.py:374-396) -> ECEF (utilFcns.py:91-121); the caller supplies
// sin / cos of the origin's latitude and longitude.
RDR_HD void inc_hd_to_ecef_sc(double inc_deg, double hd_deg, double sla, double cla, double slo, double clo,
                              double& u, double& v, double& w) {
    double si, ci, sh, ch;
    sincos(inc_deg * DEG_TO_RAD, &si, &ci);
    sincos((hd_deg + 90.0) * DEG_TO_RAD, &sh, &ch);
    const double east = si * ch, north = si * sh, up = ci;
    const double t = cla * up - sla * north;
    w = sla * up + cla * north;
    u = clo * t - slo * east;
    v = slo * t + clo * east;
}

RDR_HD void inc_hd_to_ecef(double inc_deg, double hd_deg, double lat_deg, double lon_deg,
                           double& u, double& v, double& w) {
    double sla, cla, slo, clo;
    sincos(lat_deg * DEG_TO_RAD, &sla, &cla);
    sincos(lon_deg * DEG_TO_RAD, &slo, &clo);
    inc_hd_to_ecef_sc(inc_deg, hd_deg, sla, cla, slo, clo, u, v, w);
}

// ---- Lambert conformal conic (PROJ `lcc`, Snyder 15-1..15-4 / 14-1..14-4 for the sphere) ---------------------------
// Replaces pyproj's geodetic -> model-CRS step for projected weather models (HRRR: models/hrrr.py:248-259;
// call sites delay.py:207-209,253,295).  Host code derives n, a*F and rho0 once (lcc_setup); the device evaluates
// rho = a F t(phi)^n, theta = n (lam - lam0), x = x0 + rho sin(theta), y = y0 + rho0 - rho cos(theta).
struct LccParams {
    int kind;                 // 0 = cube is on lon/lat (no projection), 1 = LCC
    double n, aF, rho0;       // cone constant, a*F, radius of the origin parallel
    double lam0, x0, y0;      // central meridian (rad), false easting / northing (m)
    double e;                 // first eccentricity (0 for a sphere)
};

inline double lcc_tsfn(double phi, double e) {       // PROJ pj_tsfn
    const double s = sin(phi);
    double t = tan(0.5 * (1.5707963267948966 - phi));
    if (e != 0.0) t /= pow((1.0 - e * s) / (1.0 + e * s), 0.5 * e);
    return t;
}

inline LccParams lcc_setup(double a, double es, double lat1_deg, double lat2_deg, double lat0_deg, double lon0_deg, double x0, double y0) {
    LccParams L;
    L.kind = 1; L.e = sqrt(es); L.x0 = x0; L.y0 = y0; L.lam0 = lon0_deg * DEG_TO_RAD;
    const double p1 = lat1_deg * DEG_TO_RAD, p2 = lat2_deg * DEG_TO_RAD, p0 = lat0_deg * DEG_TO_RAD;
    const double m1 = cos(p1) / sqrt(1.0 - es * sin(p1) * sin(p1));
    const double t1 = lcc_tsfn(p1, L.e);
    if (fabs(p1 - p2) >= 1e-10) {
        const double m2 = cos(p2) / sqrt(1.0 - es * sin(p2) * sin(p2));
        L.n = log(m1 / m2) / log(t1 / lcc_tsfn(p2, L.e));
    } else {
        L.n = sin(p1);
    }
    const double F = m1 * pow(t1, -L.n) / L.n;
    L.aF = a * F;
    L.rho0 = (fabs(fabs(p0) - 1.5707963267948966) < 1e-10) ? 0.0 : L.aF * pow(lcc_tsfn(p0, L.e), L.n);
    return L;
}

// Polar stereographic (PROJ `stere` with lat_0 = +-90; Snyder 21-33 / 21-34 / 21-39; HRRR-AK: models/hrrr.py:22-25,359) is the
// cone of constant n = +1 (north) or -1 (south): rho = a m_c t / t_c (true scale at lat_ts) or 2 a k_0 t / sqrt((1+e)^(1+e)
// (1-e)^(1-e)) (scale k_0 at the pole), theta = n (lam - lam0), x = x0 + rho sin(theta), y = y0 - rho cos(theta) - i.e. exactly
// lcc_forward with rho0 = 0.  For the southern aspect t(phi)^-1 = t(-phi) does the mirroring and a NEGATIVE a F restores the
// signs (x = |rho| sin(dlam), y = +|rho| cos(dlam)).  So every kernel that handles LCC cubes handles these unchanged.
// lat_ts: NaN -> use k0.
inline LccParams stere_setup(double a, double es, double lat0_deg, double lat_ts_deg, double k0, double lon0_deg, double x0, double y0) {
    LccParams L;
    L.kind = 1; L.e = sqrt(es); L.x0 = x0; L.y0 = y0; L.lam0 = lon0_deg * DEG_TO_RAD; L.rho0 = 0.0;
    const bool south = lat0_deg < 0;
    L.n = south ? -1.0 : 1.0;
    double scale;
    if (lat_ts_deg == lat_ts_deg && fabs(fabs(lat_ts_deg) - 90.0) > 1e-9) {
        const double pc = fabs(lat_ts_deg) * DEG_TO_RAD;
        scale = cos(pc) / sqrt(1.0 - es * sin(pc) * sin(pc)) / lcc_tsfn(pc, L.e);
    } else {
        scale = 2.0 * k0 / sqrt(pow(1.0 + L.e, 1.0 + L.e) * pow(1.0 - L.e, 1.0 - L.e));
    }
    L.aF = (south ? -1.0 : 1.0) * a * scale;
    return L;
}

RDR_HD void lcc_forward(const LccParams& L, double lat_deg, double lon_deg, double& x, double& y) {
    const double phi = lat_deg * DEG_TO_RAD;
    double dlam = lon_deg * DEG_TO_RAD - L.lam0;
    // PROJ reduces lam - lam0 to (-pi, pi]
    if (dlam > 3.141592653589793) dlam -= 6.283185307179586;
    else if (dlam < -3.141592653589793) dlam += 6.283185307179586;
    const double s = sin(phi);
    double t = tan(0.5 * (1.5707963267948966 - phi));
    if (L.e != 0.0) t /= pow((1.0 - L.e * s) / (1.0 + L.e * s), 0.5 * L.e);
    const double rho = L.aF * pow(t, L.n);
    double st, ct;
    sincos(L.n * dlam, &st, &ct);
    x = L.x0 + rho * st;
    y = L.y0 + L.rho0 - rho * ct;
}

// Inverse of lcc_forward for both cones (Snyder 15-10, 15-11, 14-9 / 15-9 with the 7-9 iteration for the ellipsoid; PROJ's
// `lcc` / polar `stere` inverse): rho carries the sign of a F (southern cones and the southern stereographic aspect have a
// negative one), theta = atan2(+-x', +-y''), t = (rho / a F)^(1/n), lam = lam0 + theta / n, phi from t.
RDR_HD void lcc_inverse(const LccParams& L, double x, double y, double& lat_deg, double& lon_deg) {
    const double sg = L.aF < 0.0 ? -1.0 : 1.0;
    const double xp = sg * (x - L.x0), yp = sg * (L.rho0 - (y - L.y0));
    const double rho = sg * sqrt(xp * xp + yp * yp);
    if (rho == 0.0) { lat_deg = L.n > 0.0 ? 90.0 : -90.0; lon_deg = L.lam0 * 57.295779513082321; }
    else {
        const double t = pow(rho / L.aF, 1.0 / L.n);
        double phi = 1.5707963267948966 - 2.0 * atan(t);
        if (L.e != 0.0) {
            for (int it = 0; it < 15; ++it) {                    // contracts by ~e^2 per step
                const double es_ = L.e * sin(phi);
                const double nphi = 1.5707963267948966 - 2.0 * atan(t * pow((1.0 - es_) / (1.0 + es_), 0.5 * L.e));
                const double d = nphi - phi;
                phi = nphi;
                if (fabs(d) < 1e-15) break;
            }
        }
        lat_deg = phi * 57.295779513082321;
        lon_deg = (L.lam0 + atan2(xp, yp) / L.n) * 57.295779513082321;
    }
    if (lon_deg > 180.0) lon_deg -= 360.0;
    else if (lon_deg < -180.0) lon_deg += 360.0;
}

}  // namespace rdr
