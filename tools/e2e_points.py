#!/usr/bin/env python3
"""End-to-end timing of the POINT branch of the host API (delay.py:96-128) - NumPy arrays in, NumPy arrays out, everything between
them on the device (round 4: intermediate cube kept on the GPU, one gather for both fields, Conventional's division in the same launch).

  c2   BASELINE configs[1] through tropo_delay: N query points with their own heights (default 10^6), processed ERA5-sized
       300x300x80 cube read from a NetCDF file on disk, Zenith and Conventional (incidence raster + heading) lines of sight.
  c5   BASELINE configs[4] through tropo_delay: N stations (default 5*10^6) on the two-epoch blend of HRRR-sized 1000x1000x50
       cubes on a 3-km Lambert-conformal-conic grid (device-resident ProcessedModel, blended on the device), zenith.

usage: e2e_points.py [c2|c5] [npoints] [profile]        -> one JSON line"""
import datetime as dt
import json
import sys
import tempfile
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
from raider_amd.delay import PointsAOI, tropo_delay, transformPoints      # noqa: E402
from raider_amd.losreader import Conventional, Zenith                     # noqa: E402
from raider_amd.synthetic import synthetic_cube                           # noqa: E402

mode = sys.argv[1] if len(sys.argv) > 1 else 'c2'
WHEN = dt.datetime(2020, 1, 30, 13, 52, 45)
res = {'mode': mode}


def timed_calls(name, fn, n, reps=6):
    runs = []
    out = None
    for _ in range(reps):
        t0 = time.perf_counter()
        out = fn()
        runs.append(time.perf_counter() - t0)
    res[f'{name}_first_call_ms'] = runs[0] * 1e3
    res[f'{name}_warm_ms'] = [round(r * 1e3, 3) for r in runs[1:]]
    res[f'{name}_ms'] = min(runs[1:]) * 1e3
    res[f'{name}_points_per_s'] = n / min(runs[1:])
    w, h = out
    res[f'{name}_mean_hydro'] = float(np.nanmean(h)); res[f'{name}_nan'] = float(np.isnan(h).mean())
    return out


if mode == 'c2':
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 1_000_000
    c = synthetic_cube(300, 300, 80, seed=0)
    tmp = Path(tempfile.mkdtemp()) / 'ERA5_synthetic.nc'
    from scipy.io import netcdf_file
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
    inc = rng.uniform(30.0, 46.0, n)
    res.update(points=n, cube='300x300x80 (NetCDF-3 file on disk)')
    # the AOI object of a job is made once and serves every date (cli/raider.py:347-355 loops over the dates with one aoi; its output
    # grid is laid out by the first call, delay.py:142-151): the timed calls reuse it.  `*_fresh_aoi`: a new AOI per call as well
    # (four min / max passes over the points to find their bounding box - host work of the AOI provider, outside the path).
    aoi = PointsAOI(lats, lons, hgts)
    wz, hz = timed_calls('zenith', lambda: tropo_delay(WHEN, str(tmp), aoi, Zenith(), None, 4326, None), n)
    los = Conventional(inc=inc, heading=np.full(n, -167.9))
    wc, hc = timed_calls('conventional', lambda: tropo_delay(WHEN, str(tmp), aoi, los, None, 4326, None), n)
    timed_calls('conventional_fresh_aoi', lambda: tropo_delay(WHEN, str(tmp), PointsAOI(lats, lons, hgts), los, None, 4326, None), n)
    # where the time goes: the two library calls of the branch by themselves
    from raider_amd.delayFcns import getInterpolators
    cube = getInterpolators(str(tmp), 'total')[0].cube
    zl = np.asarray(c['zs'], dtype=np.float64)
    def best(fn, reps=6):
        ts = []
        for _ in range(reps):
            t0 = time.perf_counter(); r = fn(); ts.append(time.perf_counter() - t0)
        return min(ts[1:]) * 1e3, r
    res['piece_build_delay_cube_ms'], dcube = best(lambda: cube.build_delay_cube(aoi.xpts, aoi.ypts, zl))
    res['piece_interp_project_ms'], _ = best(lambda: dcube.interp_project(lats, lons, hgts, inc=inc))
    res['piece_interp_ms'], _ = best(lambda: dcube.interp_project(lats, lons, hgts))
    res['intermediate_grid'] = [int(zl.size), int(aoi.ypts.size), int(aoi.xpts.size)]
    up = np.cos(np.radians(inc))
    res['conventional_vs_zenith_over_cos_max_rel'] = float(np.nanmax(np.abs(hc * up / hz - 1.0)))
    # the same job with the file cache off: every call opens the file and uploads the 115 MB of f64 totals again
    import os
    os.environ['RAIDER_HIP_FILE_CACHE'] = '0'
    timed_calls('zenith_nocache', lambda: tropo_delay(WHEN, str(tmp), PointsAOI(lats, lons, hgts), Zenith(), None, 4326, None), n, reps=4)
    del os.environ['RAIDER_HIP_FILE_CACHE']
    prof = lambda: tropo_delay(WHEN, str(tmp), PointsAOI(lats, lons, hgts), los, None, 4326, None)
else:
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 5_000_000
    import torch
    from raider_amd.engine import Cube
    from raider_amd.weather import ProcessedModel
    dev = torch.device('cuda:0')
    lcc = '+proj=lcc +lat_1=38.5 +lat_2=38.5 +lat_0=38.5 +lon_0=262.5 +x_0=0 +y_0=0 +a=6371229 +b=6371229 +units=m +no_defs'
    ny = nx = 1000; nz = 50
    xs = -1.5e6 + 3000.0 * np.arange(nx); ys = -1.5e6 + 3000.0 * np.arange(ny)
    zs = np.round(-100.0 + 26100.0 * np.linspace(0, 1, nz) ** 2, 3)
    zt = torch.from_numpy(zs).to(dev)

    def epoch(seed):
        g = torch.Generator(device=dev); g.manual_seed(seed)
        gh = torch.randn((ny, nx, 1), generator=g, device=dev); gw = torch.randn((ny, nx, 1), generator=g, device=dev)
        hyd = (270.0 * torch.exp(-zt / 8000.0) * (1 + 0.01 * gh)).float(); wet = (60.0 * torch.exp(-zt / 2000.0) * (1 + 0.1 * gw)).float()
        # totals: 1e-6 * trapz(X[..., l:], zs[l:]) (weatherModel.py:398-401)
        def tot(f):
            f = f.double(); d = zt[1:] - zt[:-1]
            t = 0.5 * (f[..., 1:] + f[..., :-1]) * d
            s = torch.flip(torch.cumsum(torch.flip(t, (-1,)), -1), (-1,))
            return 1e-6 * torch.cat([s, torch.zeros_like(s[..., :1])], -1)
        pw = Cube(ys, xs, zs, wet.contiguous(), hyd.contiguous(), order='yxz')
        tt = Cube(ys, xs, zs, tot(wet).contiguous(), tot(hyd).contiguous(), order='yxz')
        return pw, tt
    t0 = time.perf_counter()
    (p1, t1), (p2, t2) = epoch(0), epoch(1)
    torch.cuda.synchronize()
    res['epochs_setup_ms'] = (time.perf_counter() - t0) * 1e3
    t0 = time.perf_counter()
    model = ProcessedModel(p1.blend(0.25, p2, 0.75), t1.blend(0.25, t2, 0.75), zs, proj=lcc)      # cli/raider.py:817-819, on the device
    model.total.ctx.synchronize()
    res['blend_ms'] = (time.perf_counter() - t0) * 1e3
    del p1, p2, t1, t2
    rng = np.random.default_rng(3)
    px = rng.uniform(-1.2e6, 1.2e6, n); py = rng.uniform(-1.2e6, 1.2e6, n); hgts = rng.uniform(0.0, 4000.0, n)
    ll = transformPoints(py, px, hgts, lcc, 4326)
    lats, lons = np.ascontiguousarray(ll[..., 0]), np.ascontiguousarray(ll[..., 1])
    sp = 0.03                                                                    # ~3 km: the model's own resolution in degrees
    xg = np.arange(lons.min() - sp, lons.max() + 2 * sp, sp); yg = np.arange(lats.max() + sp, lats.min() - 2 * sp, -sp)
    hl = list(zs[zs <= 6000.0]) + [8000.0]                                       # station heights end at 4 km
    res.update(points=n, cube='blend(0.25, 0.75) of two 1000x1000x50 epochs, 3-km LCC grid, device-resident', intermediate_grid=[len(hl), yg.size, xg.size])
    import logging
    logging.getLogger('RAiDER').setLevel(logging.CRITICAL + 1)               # (the lon/lat bounding grid leaves the LCC cube at its corners: NaN nodes, logged per call)
    wz, hz = timed_calls('zenith', lambda: tropo_delay(WHEN, model, PointsAOI(lats, lons, hgts, xg, yg), Zenith(), hl, 4326, None), n)
    prof = lambda: tropo_delay(WHEN, model, PointsAOI(lats, lons, hgts, xg, yg), Zenith(), hl, 4326, None)

print(json.dumps(res))
if 'profile' in sys.argv:
    import cProfile
    import pstats
    pr = cProfile.Profile(); pr.enable()
    prof()
    pr.disable()
    pstats.Stats(pr).sort_stats('cumulative').print_stats(25)
