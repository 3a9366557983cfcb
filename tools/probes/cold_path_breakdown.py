"""Where a COLD tropo_delay call spends its time: the file cache off, 1 M points, ERA5-sized NetCDF-3 model file.
Pieces: open + parse the file (no data read), map the two f64 totals, Cube() from the mapping (2 x hipMemcpyAsync from pageable
memory + pack kernel + one sync), the rest of the call.  Prints one JSON line."""
import datetime as dt, json, os, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
os.environ['RAIDER_HIP_FILE_CACHE'] = '0'
from raider_amd.delay import PointsAOI, tropo_delay
from raider_amd.losreader import Zenith
from raider_amd.synthetic import synthetic_cube
from raider_amd import delayFcns as F
import raider_amd as R
c = synthetic_cube(300, 300, 80, seed=0)
tmp = Path(tempfile.mkdtemp()) / 'ERA5_synthetic.nc'
from scipy.io import netcdf_file
with netcdf_file(str(tmp), 'w', version=2) as f:
    for d, k in (('z', 'zs'), ('y', 'ys'), ('x', 'xs')):
        f.createDimension(d, c[k].size); f.createVariable(d, 'f8', (d,))[:] = c[k]
    for k in ('wet', 'hydro'):
        f.createVariable(k, 'f4', ('z', 'y', 'x'))[:] = c[k]
    for k in ('wet_total', 'hydro_total'):
        f.createVariable(k, 'f8', ('z', 'y', 'x'))[:] = c[k]
    pj = f.createVariable('proj', 'i4', ()); pj.data[()] = 0; pj.crs_wkt = 'GEOGCRS["WGS 84",ID["EPSG",4326]]'
n = 1_000_000
rng = np.random.default_rng(1)
lats = rng.uniform(31.5, 34.5, n); lons = rng.uniform(-119.5, -115.5, n); hgts = rng.uniform(0.0, 3000.0, n)
aoi = PointsAOI(lats, lons, hgts)
def best(fn, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); r = fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return min(ts[1:]), r
res = {}
res['cold_call_ms'], _ = best(lambda: tropo_delay(dt.datetime(2020, 1, 1), str(tmp), aoi, Zenith(), None, 4326, None))
res['open_and_parse_ms'], var = best(lambda: F._load_fields(str(tmp))[0])
def raw():
    v = F._load_fields(str(tmp))[0]
    return v['wet_total'].raw(), v['hydro_total'].raw()
res['open_parse_map_ms'], (w, h) = best(raw)
xs, ys, zs = (np.array(var[k][:]) for k in ('x', 'y', 'z'))
res['cube_from_mapping_ms'], _ = best(lambda: R.Cube(ys, xs, zs, *raw(), order='zyx'))
wr, hr = np.array(w), np.array(h)          # resident native copies
res['cube_from_resident_bigendian_copy_ms'], _ = best(lambda: R.Cube(ys, xs, zs, wr, hr, order='zyx'))
os.environ['RAIDER_HIP_FILE_CACHE'] = '4'
F.clear_file_cache()
res['warm_call_ms'], _ = best(lambda: tropo_delay(dt.datetime(2020, 1, 1), str(tmp), aoi, Zenith(), None, 4326, None))
res['bytes_uploaded'] = int(w.nbytes + h.nbytes)
res['upload_GBps_if_all_of_cube_from_mapping'] = res['bytes_uploaded'] / (res['cube_from_mapping_ms'] - res['open_parse_map_ms']) / 1e6
print(json.dumps(res))
