#!/usr/bin/env python3
"""Orbit-based ray tracing end to end (host API): Raytracing(<state-vector file>) -> per-height zero-Doppler look vectors -> batched
ray tracing.  usage: e2e_orbit.py [ny nx nheights]"""
import datetime as dt
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from oracle import raider_oracle as O                                   # noqa: E402  (synthetic cube + orbit mid point only)
from raider_amd.delay import _build_cube_ray                            # noqa: E402
from raider_amd.delayFcns import getInterpolators                       # noqa: E402
from raider_amd.losreader import Raytracing                             # noqa: E402

ny, nx, nh = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 1000, 8)
d = REPO / 'tests' / 'golden' / 'orbit_files'
t0 = dt.datetime(2018, 11, 12, 23, 0, 2)
los = Raytracing(str(d / 'S1_sv_file.txt'), time=t0 + dt.timedelta(seconds=35))
orb = los._orbit
mid, _ = O.orbit_hermite(orb.time, orb.position, orb.velocity, [35.0])
lon_s, lat_s, _ = O.ecef2lla(mid[:, 0], mid[:, 1], mid[:, 2])
ypts = lat_s[0] + np.linspace(0.12, -0.12, ny); xpts = lon_s[0] - np.linspace(2.4, 4.6, nx)
c = O.synthetic_cube(120, 300, 80, seed=4, y0=lat_s[0] - 2, y1=lat_s[0] + 2, x0=lon_s[0] - 7, x1=lon_s[0] - 0.5)
wm = dict(x=c['xs'], y=c['ys'], z=c['zs'], wet=c['wet'], hydro=c['hydro'])
ip = list(getInterpolators(wm))
zref = float(c['zs'].max() - 1)
zpts = np.linspace(0.0, 7000.0, nh)
for rep in range(4):
    t = time.perf_counter()
    w, h = _build_cube_ray(xpts, ypts, zpts, los, 4326, 4326, ip, MAX_TROPO_HEIGHT=zref)
    el = time.perf_counter() - t
    print(f'rep {rep}: {ny}x{nx}x{nh} = {ny * nx * nh / 1e6:.1f} M rays in {el * 1e3:.1f} ms = {ny * nx * nh / el / 1e6:.1f} M rays/s; mean hydro {np.nanmean(h):.6f} nan {np.isnan(h).mean():.3f}')
