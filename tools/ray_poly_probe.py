#!/usr/bin/env python3
"""How well do degree-N polynomials in the ray parameter reproduce height / latitude / longitude along a straight ray?
(design probe for the ray-polynomial kernels, CPU only).  For random rays it interpolates the exact geodetic
coordinates at the N+1 Chebyshev nodes of the ray's parameter range [min(0,ht)-1, max(zref,(zref-ht)/cos(inc))+1] and
reports the worst interpolation error over the rays the kernels' static classification admits
(cos(inc) > 0.05, cos(lat) > gam + 0.02, gam < 0.08 (cos(lat) - gam), gam < 0.035, gam = (zref-ht)/(6.3e6 cos(inc))).
usage: ray_poly_probe.py [degree=5] [nrays=3000] [lon_travel_limit=0.2]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from oracle import raider_oracle as O  # noqa: E402

A = 6378137.0
F = 1 / 298.257223563
ES = 2 * F - F * F


def exact_llh(o, l, t):
    """iterated (converged) geodetic latitude, longitude and the support-function height of o + t l"""
    p = o[None, :] + t[:, None] * l[None, :]
    x, y, z = p.T
    pp = np.hypot(x, y)
    phi = np.arctan2(z, pp * (1 - ES))
    for _ in range(6):
        n = A / np.sqrt(1 - ES * np.sin(phi) ** 2)
        hh = pp / np.cos(phi) - n
        phi = np.arctan2(z, pp * (1 - ES * n / (n + hh)))
    h = pp * np.cos(phi) + z * np.sin(phi) - A * np.sqrt(1 - ES * np.sin(phi) ** 2)
    return h, phi, np.unwrap(np.arctan2(y, x))


def main():
    deg = int(sys.argv[1]) if len(sys.argv) > 1 else 5
    nrays = int(sys.argv[2]) if len(sys.argv) > 2 else 3000
    lon_lim = float(sys.argv[3]) if len(sys.argv) > 3 else 0.2
    rng = np.random.default_rng(0)
    n = deg + 1
    u = np.cos(np.pi * (2 * np.arange(n) + 1) / (2 * n))
    V = np.vander(u, n, increasing=True)
    rows = []
    for _ in range(nrays):
        lat0 = rng.uniform(-89.5, 89.5); inc = rng.uniform(0, 86); hd = rng.uniform(-180, 180)
        zref = rng.choice([15000., 40000., 80000.]); ht = rng.choice([0., -100., 3000.]); lon0 = rng.uniform(-170, 170)
        o = np.array(O.lla2ecef(np.array([lat0]), np.array([lon0]), np.array([ht]))).ravel()
        l = O.look_vectors_from_inc_hd(np.array([inc]), np.array([hd]), np.array([lat0]), np.array([lon0]), ht).ravel()
        cosi, c0 = np.cos(np.radians(inc)), np.cos(np.radians(lat0))
        gam = (zref - ht) / (cosi * 6.3e6)
        tb = max(zref, (zref - ht) / cosi) + 1; ta = min(0.0, ht) - 1.0
        tn = 0.5 * (ta + tb) + 0.5 * (tb - ta) * u
        hn, pn, ln = exact_llh(o, l, tn)
        td = np.linspace(ta, tb, 801); ud = (2 * td - (ta + tb)) / (tb - ta)
        hd_, pd_, ld_ = exact_llh(o, l, td)
        ld_ += np.round((ln.mean() - ld_.mean()) / (2 * np.pi)) * 2 * np.pi
        ch = np.linalg.solve(V, hn); cp = np.linalg.solve(V, pn - pn[n // 2]); cl = np.linalg.solve(V, ln - ln[n // 2])
        eh = np.abs(np.polyval(ch[::-1], ud) - hd_).max()
        ep = np.abs(np.polyval(cp[::-1], ud) - (pd_ - pn[n // 2])).max() * 6.4e6
        el = np.abs(np.polyval(cl[::-1], ud) - (ld_ - ln[n // 2])).max() * 6.4e6 * max(c0, 0.01)
        rows.append((gam, c0, cosi, tb, eh, ep, el))
    r = np.array(rows)
    gam, c0, cosi, tb, eh, ep, el = r.T
    ok = (cosi > 0.05) & (c0 > gam + 0.02) & (gam < lon_lim * (c0 - gam)) & (gam < 0.035)
    print(f'degree {deg}, longitude-travel limit {lon_lim}: {ok.sum()} of {nrays} rays admitted by the classification')
    for lim in (100e3, 1e9):
        m = ok & (tb < lim)
        print(f'  rays shorter than {lim / 1e3:.0f} km: n={m.sum():5d}  max interpolation error  h {eh[m].max():.1e} m   lat {ep[m].max():.1e} m   lon {el[m].max():.1e} m (on the ground)')


if __name__ == '__main__':
    main()
