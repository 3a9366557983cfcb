#!/usr/bin/env python3
"""End-to-end timing of the ZENITH branch of the host API (BASELINE configs[1] sizes): tropo_delay(datetime, processed-cube file, grid AOI, Zenith(), heights)
from a NetCDF file on disk to NumPy delay cubes, ERA5-sized synthetic cube.  usage: e2e_tropo_delay.py [rows cols nheights]"""
import datetime as dt
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from raider_amd.delay import GridAOI, tropo_delay                     # noqa: E402
from raider_amd.losreader import Raytracing, Zenith                           # noqa: E402
from raider_amd.synthetic import synthetic_cube                       # noqa: E402

rows, cols, nh = (int(v) for v in sys.argv[1:4]) if len(sys.argv) > 3 else (1000, 1000, 40)
c = synthetic_cube(300, 300, 80, seed=0)
tmp = Path(tempfile.mkdtemp()) / 'ERA5_synthetic.nc'
from scipy.io import netcdf_file                                      # noqa: E402
with netcdf_file(str(tmp), 'w', version=2) as f:                      # the layout of weatherModel.py:659-724, as NetCDF-3
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
x = np.linspace(-119.5, -115.5, cols); y = np.linspace(34.5, 31.5, rows)
inc = np.broadcast_to(30.0 + 16.0 * np.arange(cols) / cols, (rows, cols)).copy()
heights = list(np.linspace(0.0, 3500.0, nh))
res = {}
ds = hyd = None
for rep in range(6):
    del ds, hyd
    t0 = time.perf_counter()
    ds, _ = tropo_delay(dt.datetime(2020, 1, 30, 13, 52, 45), str(tmp), GridAOI(x, y), Zenith(), heights, 4326, None)
    hyd = np.asarray(ds['hydro'][:])
    res[f'run{rep}_s'] = time.perf_counter() - t0
best = min(v for k, v in res.items() if k != 'run0_s')
res.update(rays=rows * cols * nh, rays_per_s=rows * cols * nh / best, mean_hydro=float(np.nanmean(hyd)), nan=float(np.isnan(hyd).mean()))
print(json.dumps(res))
if len(sys.argv) > 4 and sys.argv[4] == 'profile':
    import cProfile
    import pstats
    pr = cProfile.Profile(); pr.enable()
    tropo_delay(dt.datetime(2020, 1, 30, 13, 52, 45), str(tmp), GridAOI(x, y), Zenith(), heights, 4326, None)
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(28)
