import datetime as dt, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
from raider_amd.delay import GridAOI, tropo_delay, _get_delays_on_cube
from raider_amd.losreader import Zenith
from raider_amd.synthetic import synthetic_cube
import raider_amd.delay as D, raider_amd.engine as E
c = synthetic_cube(300, 300, 80, seed=0)
tmp = Path(tempfile.mkdtemp()) / 'c.nc'
from scipy.io import netcdf_file
with netcdf_file(str(tmp), 'w', version=2) as f:
    for d, k in (('z', 'zs'), ('y', 'ys'), ('x', 'xs')):
        f.createDimension(d, c[k].size); f.createVariable(d, 'f8', (d,))[:] = c[k]
    for k in ('wet', 'hydro'): f.createVariable(k, 'f4', ('z', 'y', 'x'))[:] = c[k]
    for k in ('wet_total', 'hydro_total'): f.createVariable(k, 'f8', ('z', 'y', 'x'))[:] = c[k]
    pj = f.createVariable('proj', 'i4', ()); pj.data[()] = 0; pj.crs_wkt = 'GEOGCRS["WGS 84",ID["EPSG",4326]]'
x = np.linspace(-119.5, -115.5, 1000); y = np.linspace(34.5, 31.5, 1000); heights = list(np.linspace(0, 3500, 40))
orig_bc = E.Cube.build_cube
def timed_bc(self, *a, **k):
    t0 = time.perf_counter(); r = orig_bc(self, *a, **k); print('   build_cube %.1f ms' % ((time.perf_counter() - t0) * 1e3)); return r
E.Cube.build_cube = timed_bc
orig_hn = D._has_nan
def timed_hn(a):
    t0 = time.perf_counter(); r = orig_hn(a); print('   has_nan %.1f ms' % ((time.perf_counter() - t0) * 1e3)); return r
D._has_nan = timed_hn
ds = hyd = None
for rep in range(6):
    t0 = time.perf_counter(); del ds, hyd; t1 = time.perf_counter()
    ds, _ = tropo_delay(dt.datetime(2020, 1, 30, 13, 52, 45), str(tmp), GridAOI(x, y), Zenith(), heights, 4326, None); t2 = time.perf_counter()
    hyd = np.asarray(ds['hydro'][:]); t3 = time.perf_counter()
    print(f'rep {rep}: del {1e3*(t1-t0):.1f} tropo {1e3*(t2-t1):.1f} asarray {1e3*(t3-t2):.1f} ms')
