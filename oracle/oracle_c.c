/* oracle_c.c - plain C (OpenMP) restatement of RAiDER's ray-traced delay path.
 *
 * *** TEST INFRASTRUCTURE - NOT PART OF THE PRODUCT. ***  Only tests/, __graft_entry__.smoke() and bench.py's
 * cpu_baseline leg may load this library (as the checker / the reported multi-core CPU baseline).
 *
 * Same algorithm as oracle/raider_oracle.py (which is pinned against the reference's golden vectors), written as
 * per-ray loops so that it can use every host core:
 *   lla2ecef / ecef2lla            PROJ `cart` as called from utilFcns.py:77-88, delay.py:238,253,295
 *   getTopOfAtmosphere             losreader.py:706-733   (10 iterations factor 1, or 3 with the per-ray cos factor)
 *   build_ray                      losreader.py:772-835   (levels table computed by the caller: raider_oracle.ray_levels)
 *   nParts                         delay.py:283           (max over the whole slice, reduced across threads)
 *   sample loop + scipy linear RGI delay.py:285-323, scipy _rgi.py:405-499
 * Parity: checked against raider_oracle.build_cube_ray (and therefore the reference goldens) in
 * tests/test_oracle_c.py to 1e-11 m.  The WGS84<->ECEF arithmetic is the same restatement of PROJ as in the NumPy
 * oracle ("parity unpinned" for that conversion).
 *
 * build: gcc -O3 -fopenmp -shared -fPIC -ffp-contract=off oracle_c.c -o liboracle_c.so -lm   (__graft_entry__.build)
 */
#include <math.h>
#include <stdint.h>
#include <stdlib.h>
#ifdef _OPENMP
#include <omp.h>
#endif

static const double A = 6378137.0;
static const double RF = 298.257223563;
#define F (1.0 / RF)
#define ES (2.0 * F - F * F)
#define B ((1.0 - F) * A)
#define E2S (ES / (1.0 - ES))
static const double D2R = 0.017453292519943296;
static const double R2D = 57.295779513082321;

void orc_set_threads(int n) {
#ifdef _OPENMP
    if (n > 0) omp_set_num_threads(n);
#else
    (void)n;
#endif
}

int orc_num_threads(void) {
#ifdef _OPENMP
    return omp_get_max_threads();
#else
    return 1;
#endif
}

static void lla2ecef(double lat, double lon, double h, double* x, double* y, double* z) {
    const double phi = lat * D2R, lam = lon * D2R;
    const double sp = sin(phi), cp = cos(phi);
    const double N = A / sqrt(1.0 - ES * sp * sp);
    *x = (N + h) * cp * cos(lam);
    *y = (N + h) * cp * sin(lam);
    *z = (N * (1.0 - ES) + h) * sp;
}

/* PROJ cart.cpp geodetic(): returns lon, lat (deg), h */
static void ecef2lla(double x, double y, double z, double* lon, double* lat, double* h) {
    const double p = sqrt(x * x + y * y);
    const double yt = z * A, xt = p * B;
    const double nrm = sqrt(yt * yt + xt * xt);
    const double c = nrm == 0 ? 1.0 : xt / nrm, s = nrm == 0 ? 0.0 : yt / nrm;
    const double yphi = z + E2S * B * s * s * s;
    const double xphi = p - ES * A * c * c * c;
    const double nphi = sqrt(yphi * yphi + xphi * xphi);
    double cphi = nphi == 0 ? 1.0 : xphi / nphi, sphi = nphi == 0 ? 0.0 : yphi / nphi;
    double phi;
    if (xphi <= 0) { phi = z >= 0 ? M_PI_2 : -M_PI_2; cphi = 0; sphi = z >= 0 ? 1.0 : -1.0; }
    else phi = atan(yphi / xphi);
    *lon = atan2(y, x) * R2D;
    *lat = phi * R2D;
    if (cphi < 1e-6) {
        const double r = hypot(A * A * cphi, B * B * sphi) / hypot(A * cphi, B * sphi);
        *h = fabs(z) - r;
    } else {
        *h = p / cphi - A / sqrt(1.0 - ES * sphi * sphi);
    }
}

/* height only (what the Newton iteration of getTopOfAtmosphere reads, losreader.py:729-731): same formulas as
 * ecef2lla without the two arctangents; sqrt(x*x+y*y) instead of hypot (1 ulp) because glibc's hypot is ~10x slower */
static double ecef_h(double x, double y, double z) {
    const double p = sqrt(x * x + y * y);
    const double yt = z * A, xt = p * B;
    const double nrm = sqrt(yt * yt + xt * xt);
    const double c = nrm == 0 ? 1.0 : xt / nrm, s = nrm == 0 ? 0.0 : yt / nrm;
    const double yphi = z + E2S * B * s * s * s;
    const double xphi = p - ES * A * c * c * c;
    const double nphi = sqrt(yphi * yphi + xphi * xphi);
    double cphi = nphi == 0 ? 1.0 : xphi / nphi, sphi = nphi == 0 ? 0.0 : yphi / nphi;
    if (xphi <= 0) { cphi = 0; sphi = z >= 0 ? 1.0 : -1.0; }
    if (cphi < 1e-6) return fabs(z) - hypot(A * A * cphi, B * B * sphi) / hypot(A * cphi, B * sphi);
    return p / cphi - A / sqrt(1.0 - ES * sphi * sphi);
}

static void toa(const double* o, const double* l, double hgt, int iters, double factor, double* pos) {
    for (int k = 0; k < 3; ++k) pos[k] = o[k] + hgt * l[k];
    for (int it = 0; it < iters; ++it) {
        const double step = (hgt - ecef_h(pos[0], pos[1], pos[2])) / factor;
        for (int k = 0; k < 3; ++k) pos[k] = pos[k] + l[k] * step;
    }
}

/* first index with x < g[i] */
static int upper(const double* g, int n, double x) {
    int lo = 0, hi = n;
    while (lo < hi) { const int mid = (lo + hi) / 2; if (x < g[mid]) hi = mid; else lo = mid + 1; }
    return lo;
}

/* scipy linear RGI on both fields; cube (y,x,z) C-order, dtype 0 = f32, 1 = f64 */
static void rgi2(const double* ys, int ny, const double* xs, int nx, const double* zs, int nz, const void* wet, const void* hyd, int dtype,
                 double y, double x, double z, double* ow, double* oh) {
    if (!(y >= ys[0] && y <= ys[ny - 1] && x >= xs[0] && x <= xs[nx - 1] && z >= zs[0] && z <= zs[nz - 1])) { *ow = NAN; *oh = NAN; return; }
    int iy = upper(ys, ny, y) - 1; if (iy > ny - 2) iy = ny - 2; if (iy < 0) iy = 0;
    int ix = upper(xs, nx, x) - 1; if (ix > nx - 2) ix = nx - 2; if (ix < 0) ix = 0;
    int iz = upper(zs, nz, z) - 1; if (iz > nz - 2) iz = nz - 2; if (iz < 0) iz = 0;
    const double ty = (y - ys[iy]) / (ys[iy + 1] - ys[iy]);
    const double tx = (x - xs[ix]) / (xs[ix + 1] - xs[ix]);
    const double tz = (z - zs[iz]) / (zs[iz + 1] - zs[iz]);
    double sw = 0.0, sh = 0.0;
    for (int a = 0; a < 2; ++a)
        for (int b = 0; b < 2; ++b)
            for (int c = 0; c < 2; ++c) {
                const double wgt = ((1.0 * (a ? ty : 1 - ty)) * (b ? tx : 1 - tx)) * (c ? tz : 1 - tz);
                const int64_t idx = ((int64_t)(iy + a) * nx + (ix + b)) * nz + (iz + c);
                const double vw = dtype ? ((const double*)wet)[idx] : (double)((const float*)wet)[idx];
                const double vh = dtype ? ((const double*)hyd)[idx] : (double)((const float*)hyd)[idx];
                sw = sw + vw * wgt; sh = sh + vh * wgt;
            }
    *ow = sw; *oh = sh;
}

/* Pass 1: per-level ray lengths' maximum over all rays (delay.py:283) -> maxlen[K], and the two all-pixels z-clamp
 * predicates of delay.py:306-311 for the first / last sample of the rays: clamp[0] = every first sample is below zmin,
 * clamp[1] = every last sample is above zmax.  origin = lla2ecef(lat, lon, ht). */
void orc_prepass(const double* lat, const double* lon, const double* los, int64_t n, double ht,
                 const double* lo, const double* hi, int K, double zmin, double zmax, double* maxlen, int* clamp) {
    for (int k = 0; k < K; ++k) maxlen[k] = 0.0;
    int all_below = 1, all_above = 1;
#pragma omp parallel reduction(&& : all_below, all_above)
    {
        double* loc = (double*)calloc((size_t)K, sizeof(double));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            double o[3], low[3], high[3], cosf = 1.0;
            lla2ecef(lat[i], lon[i], ht, &o[0], &o[1], &o[2]);
            const double* l = los + 3 * i;
            for (int k = 0; k < K; ++k) {
                if (k == 0) toa(o, l, lo[0], 10, 1.0, low); else { low[0] = high[0]; low[1] = high[1]; low[2] = high[2]; }
                toa(o, l, hi[k], k == 0 ? 10 : 3, cosf, high);
                const double dx = high[0] - low[0], dy = high[1] - low[1], dz = high[2] - low[2];
                const double L = sqrt(dx * dx + dy * dy + dz * dz);
                if (k == 0) cosf = (hi[0] - lo[0]) / L;
                if (!isnan(loc[k]) && (isnan(L) || L > loc[k])) loc[k] = L;      /* NaN poisons the max like ndarray.max */
                if (k == 0) all_below = all_below && (ecef_h(low[0] + 0.0 * dx, low[1] + 0.0 * dy, low[2] + 0.0 * dz) < zmin);
                if (k == K - 1) all_above = all_above && (ecef_h(low[0] + 1.0 * dx, low[1] + 1.0 * dy, low[2] + 1.0 * dz) > zmax);
            }
        }
#pragma omp critical
        for (int k = 0; k < K; ++k) if (!isnan(maxlen[k]) && (isnan(loc[k]) || loc[k] > maxlen[k])) maxlen[k] = loc[k];
        free(loc);
    }
    clamp[0] = all_below; clamp[1] = all_above;
}

/* Pass 2: trapezoid integration with the given partition (delay.py:285-323).  clamp_lo / clamp_hi: the all-pixels
 * z-clamp decisions (delay.py:306-311) for the first / last sample, made by the caller. */
/* Lambert conformal conic forward (PROJ `lcc`, Snyder 15-1..15-4) as oracle/raider_oracle.py:lcc_forward restates it - same
 * operations in the same order; parity with PROJ itself is unpinned there and here.  par = a, es, lat_1, lat_2, lat_0, lon_0, x_0, y_0. */
typedef struct { double a, e, es, n, Fc, rho0, lon0, x0, y0; } lcc_t;
static double lcc_tsfn(double phi, double e) {
    const double s = sin(phi), t = tan(0.5 * (M_PI / 2 - phi));
    return e != 0 ? t / pow((1 - e * s) / (1 + e * s), 0.5 * e) : t;
}
static double lcc_msfn(double phi, double es) { return cos(phi) / sqrt(1 - es * sin(phi) * sin(phi)); }
static void lcc_setup(const double* par, lcc_t* L) {
    L->a = par[0]; L->es = par[1]; L->e = sqrt(par[1]);
    const double p1 = par[2] * D2R, p2 = par[3] * D2R, p0 = par[4] * D2R;
    L->n = fabs(p1 - p2) >= 1e-10 ? log(lcc_msfn(p1, L->es) / lcc_msfn(p2, L->es)) / log(lcc_tsfn(p1, L->e) / lcc_tsfn(p2, L->e)) : sin(p1);
    L->Fc = lcc_msfn(p1, L->es) * pow(lcc_tsfn(p1, L->e), -L->n) / L->n;
    L->rho0 = L->a * L->Fc * pow(lcc_tsfn(p0, L->e), L->n);
    L->lon0 = par[5] * D2R; L->x0 = par[6]; L->y0 = par[7];
}
static void lcc_fwd(const lcc_t* L, double lat, double lon, double* x, double* y) {
    double dlam = lon * D2R - L->lon0;
    dlam = dlam > M_PI ? dlam - 2 * M_PI : (dlam < -M_PI ? dlam + 2 * M_PI : dlam);
    const double rho = L->a * L->Fc * pow(lcc_tsfn(lat * D2R, L->e), L->n);
    *x = L->x0 + rho * sin(L->n * dlam); *y = L->y0 + L->rho0 - rho * cos(L->n * dlam);
}

/* proj: NULL (a lon/lat cube) or the eight LCC parameters of the cube's CRS (ecef_to_model, delay.py:253,295) */
void orc_march_proj(const double* lat, const double* lon, const double* los, int64_t n, double ht,
                    const double* lo, const double* hi, int K, const int* nparts, int clamp_lo, int clamp_hi,
                    const double* ys, int ny, const double* xs, int nx, const double* zs, int nz, const void* wet, const void* hyd, int dtype,
                    const double* proj, double* out_w, double* out_h) {
    lcc_t LC;
    if (proj) lcc_setup(proj, &LC);
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double o[3], low[3], high[3], cosf = 1.0, aw = 0.0, ah = 0.0;
        lla2ecef(lat[i], lon[i], ht, &o[0], &o[1], &o[2]);
        const double* l = los + 3 * i;
        for (int k = 0; k < K; ++k) {
            if (k == 0) toa(o, l, lo[0], 10, 1.0, low); else { low[0] = high[0]; low[1] = high[1]; low[2] = high[2]; }
            toa(o, l, hi[k], k == 0 ? 10 : 3, cosf, high);
            const double dx = high[0] - low[0], dy = high[1] - low[1], dz = high[2] - low[2];
            const double L = sqrt(dx * dx + dy * dy + dz * dz);
            if (k == 0) cosf = (hi[0] - lo[0]) / L;
            const int np = nparts[k];
            const double step = 1.0 / (np - 1.0);
            for (int j = 0; j < np; ++j) {
                const double f = (j == np - 1) ? 1.0 : j * step;
                double plon, plat, ph;
                ecef2lla(low[0] + f * dx, low[1] + f * dy, low[2] + f * dz, &plon, &plat, &ph);
                if (clamp_lo && k == 0 && j == 0) ph = zs[0];
                if (clamp_hi && k == K - 1 && j == np - 1) ph = zs[nz - 1];
                if (proj) { double px, py; lcc_fwd(&LC, plat, plon, &px, &py); plon = px; plat = py; }
                double vw, vh;
                rgi2(ys, ny, xs, nx, zs, nz, wet, hyd, dtype, plat, plon, ph, &vw, &vh);
                double wt = (j == 0 || j == np - 1) ? 0.5 : 1.0;
                wt = wt * (L * 1.0e-6 / (np - 1.0));
                aw += wt * vw; ah += wt * vh;
            }
        }
        out_w[i] = aw; out_h[i] = ah;
    }
}

void orc_march(const double* lat, const double* lon, const double* los, int64_t n, double ht,
               const double* lo, const double* hi, int K, const int* nparts, int clamp_lo, int clamp_hi,
               const double* ys, int ny, const double* xs, int nx, const double* zs, int nz, const void* wet, const void* hyd, int dtype,
               double* out_w, double* out_h) {
    orc_march_proj(lat, lon, los, n, ht, lo, hi, K, nparts, clamp_lo, clamp_hi, ys, ny, xs, nx, zs, nz, wet, hyd, dtype, NULL, out_w, out_h);
}

/* ---- per-ray origin heights (BASELINE configs' "c3b") ---------------------------------------------
 * The reference integrates one (ny,nx) slice at ONE height (delay.py:256-323); a batch whose rays start at their own heights has
 * no reference semantics.  The rule restated here applies the reference's per-slice algorithm ray by ray
 * wherever it is per-ray, and keeps its slice-level reductions batch-level:
 *   - every ray is build_ray'd with ITS height: model interval zz contributes to ray i under the tests of losreader.py:785-808
 *     with ht = hts[i] (clipped to [hts[i], zref], top interval shortened by 0.01 m, < 1 m skipped); the ray's first contributing
 *     interval gets the 10-iteration factor-1 crossings and fixes its cos_factor, later ones 3 iterations (losreader.py:812-825);
 *   - nParts[zz] = ceil(max over the rays interval zz contributes to (length) / MAX_SEGMENT_LENGTH) + 1   (delay.py:283);
 *   - the all-pixels z-clamp (delay.py:306-311) asks about every ray's own first / last sample.
 * With all heights equal this IS the slice algorithm (orc_prepass / orc_march): tests/test_oracle_c.py.
 * maxlen / nparts are indexed by the model interval zz (nz-1 entries; 0 / unused where no ray contributes). */
static int ray_level(const double* zs, int nz, int zz, double ht, double zref, double* lo, double* hi) {
    double l = zs[zz], h = zs[zz + 1];
    if (h == zs[nz - 1]) h = h - 0.01;
    if (h < ht || l >= zref) return 0;
    if (l < ht) l = ht;
    if (h > zref) h = zref;
    if (fabs(h - l) < 1.0) return 0;
    *lo = l; *hi = h;
    return 1;
}

void orc_prepass_pp(const double* lat, const double* lon, const double* hts, const double* los, int64_t n, const double* zs, int nz,
                    double zref, double zmin, double zmax, double* maxlen, int* clamp, int* any_level) {
    const int M = nz - 1;
    for (int k = 0; k < M; ++k) maxlen[k] = 0.0;
    int all_below = 1, all_above = 1, any = 0;
#pragma omp parallel reduction(&& : all_below, all_above) reduction(|| : any)
    {
        double* loc = (double*)calloc((size_t)M, sizeof(double));
#pragma omp for schedule(static)
        for (int64_t i = 0; i < n; ++i) {
            double o[3], low[3], high[3] = {0, 0, 0}, cosf = 1.0, dx = 0, dy = 0, dz = 0;
            lla2ecef(lat[i], lon[i], hts[i], &o[0], &o[1], &o[2]);
            const double* l = los + 3 * i;
            int first = 1;
            for (int zz = 0; zz < M; ++zz) {
                double lo, hi;
                if (!ray_level(zs, nz, zz, hts[i], zref, &lo, &hi)) continue;
                if (first) toa(o, l, lo, 10, 1.0, low); else { low[0] = high[0]; low[1] = high[1]; low[2] = high[2]; }
                toa(o, l, hi, first ? 10 : 3, cosf, high);
                dx = high[0] - low[0]; dy = high[1] - low[1]; dz = high[2] - low[2];
                const double L = sqrt(dx * dx + dy * dy + dz * dz);
                if (first) cosf = (hi - lo) / L;
                if (!isnan(loc[zz]) && (isnan(L) || L > loc[zz])) loc[zz] = L;
                if (first) all_below = all_below && (ecef_h(low[0] + 0.0 * dx, low[1] + 0.0 * dy, low[2] + 0.0 * dz) < zmin);
                first = 0;
            }
            if (!first) { any = 1; all_above = all_above && (ecef_h(low[0] + 1.0 * dx, low[1] + 1.0 * dy, low[2] + 1.0 * dz) > zmax); }
        }
#pragma omp critical
        for (int k = 0; k < M; ++k) if (!isnan(maxlen[k]) && (isnan(loc[k]) || loc[k] > maxlen[k])) maxlen[k] = loc[k];
        free(loc);
    }
    clamp[0] = all_below; clamp[1] = all_above; *any_level = any;
}

void orc_march_pp(const double* lat, const double* lon, const double* hts, const double* los, int64_t n, const double* zs_lev, double zref,
                  const int* nparts, int clamp_lo, int clamp_hi,
                  const double* ys, int ny, const double* xs, int nx, const double* zs, int nz, const void* wet, const void* hyd, int dtype,
                  double* out_w, double* out_h) {
    (void)zs_lev;
    const int M = nz - 1;
#pragma omp parallel for schedule(static)
    for (int64_t i = 0; i < n; ++i) {
        double o[3], low[3], high[3] = {0, 0, 0}, cosf = 1.0, aw = 0.0, ah = 0.0;
        lla2ecef(lat[i], lon[i], hts[i], &o[0], &o[1], &o[2]);
        const double* l = los + 3 * i;
        int first = 1, last_zz = -1;
        for (int zz = 0; zz < M; ++zz) { double lo, hi; if (ray_level(zs, nz, zz, hts[i], zref, &lo, &hi)) last_zz = zz; }
        for (int zz = 0; zz < M; ++zz) {
            double lo, hi;
            if (!ray_level(zs, nz, zz, hts[i], zref, &lo, &hi)) continue;
            if (first) toa(o, l, lo, 10, 1.0, low); else { low[0] = high[0]; low[1] = high[1]; low[2] = high[2]; }
            toa(o, l, hi, first ? 10 : 3, cosf, high);
            const double dx = high[0] - low[0], dy = high[1] - low[1], dz = high[2] - low[2];
            const double L = sqrt(dx * dx + dy * dy + dz * dz);
            if (first) cosf = (hi - lo) / L;
            const int np = nparts[zz];
            const double step = 1.0 / (np - 1.0);
            for (int j = 0; j < np; ++j) {
                const double f = (j == np - 1) ? 1.0 : j * step;
                double plon, plat, ph;
                ecef2lla(low[0] + f * dx, low[1] + f * dy, low[2] + f * dz, &plon, &plat, &ph);
                if (clamp_lo && first && j == 0) ph = zs[0];
                if (clamp_hi && zz == last_zz && j == np - 1) ph = zs[nz - 1];
                double vw, vh;
                rgi2(ys, ny, xs, nx, zs, nz, wet, hyd, dtype, plat, plon, ph, &vw, &vh);
                double wt = (j == 0 || j == np - 1) ? 0.5 : 1.0;
                wt = wt * (L * 1.0e-6 / (np - 1.0));
                aw += wt * vw; ah += wt * vh;
            }
            first = 0;
        }
        out_w[i] = aw; out_h[i] = ah;
    }
}
