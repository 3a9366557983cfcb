#!/usr/bin/env python3
"""End-to-end timing of the POINT branch of the host API (delay.py:101-128; BASELINE configs[1]: Conventional slant at 10^6 query points with
their own heights): tropo_delay(datetime, processed-cube file, points AOI, Conventional(inc, heading)) -> wet / hydro at the points.
usage: e2e_points.py [npoints]"""
import datetime as dt
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from raider_amd.delay import PointsAOI, tropo_delay                   # noqa: E402
from raider_amd.losreader import Conventional, Zenith                 # noqa: E402
from raider_amd.synthetic import synthetic_cube                       # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 1_000_000
c = synthetic_cube(300, 300, 80, seed=0)
tmp = Path(tempfile.mkdtemp()) / 'ERA5_synthetic.nc'
from scipy.io import netcdf_file                                      # noqa: E402
with netcdf_file(str(tmp), 'w', version=2) as f:
    for d, k in (('z', 'zs'), ('y', 'ys'), ('x', 'xs')):
        f.createDimension(d, c[k].size)
        f.createVariable(d, 'f8', (d,))[:] = c[k]
    for k in ('wet', 'hydro'):
        f.createVariable(k, 'f4', ('z', 'y', 'x'))[:] = c[k]
    for k in ('wet_total', 'hydro_total'):
        f.createVariable(k, 'f8', ('z', 'y', 'x'))[:] = c[k]
    pj = f.createVariable('proj', 'i4', ())
    pj.data[()] = 0
    pj.crs_wkt = 'GEOGCRS["WGS 84",ID["EPSG",4326]]'
rng = np.random.default_rng(1)
lats = rng.uniform(31.5, 34.5, n); lons = rng.uniform(-119.5, -115.5, n); hgts = rng.uniform(0.0, 3000.0, n)
res = {}
for name, los in (('zenith', Zenith()), ('conventional', Conventional(inc=np.full(n, 39.0), heading=np.full(n, -167.9)) if False else None)):
    if los is None:
        continue
    for rep in range(5):
        t0 = time.perf_counter()
        w, h = tropo_delay(dt.datetime(2020, 1, 30, 13, 52, 45), str(tmp), PointsAOI(lats, lons, hgts), los, None, 4326, None)
        res[f'{name}_run{rep}_s'] = time.perf_counter() - t0
    res[f'{name}_points_per_s'] = n / min(v for k, v in res.items() if k.startswith(name + '_run') and not k.endswith('run0_s'))
    res[f'{name}_mean_hydro'] = float(np.nanmean(h)); res[f'{name}_nan'] = float(np.isnan(h).mean())
print(json.dumps(res))
if len(sys.argv) > 2 and sys.argv[2] == 'profile':
    import cProfile
    import pstats
    pr = cProfile.Profile(); pr.enable()
    tropo_delay(dt.datetime(2020, 1, 30, 13, 52, 45), str(tmp), PointsAOI(lats, lons, hgts), Zenith(), None, 4326, None)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(22)
