#!/usr/bin/env python3
"""Pass-level timing on the bench scene: crossings_kernel with and without its workspace stores, march_kernel."""
import json, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import torch
import raider_amd as R
from raider_amd.synthetic import synthetic_cube, scene_grid

rows = cols = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
dev = torch.device('cuda:0')
ctx = R.Context(0)
c = synthetic_cube(300, 300, 80, seed=0)
cube = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet']).to(dev), torch.from_numpy(c['hydro']).to(dev), order='zyx', ctx=ctx)
zref = float(c['zs'].max() - 1)
xpts, ypts, inc_cols, hd = scene_grid(rows, cols)
xt, yt = torch.from_numpy(xpts).to(dev), torch.from_numpy(ypts).to(dev)
inc_t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
los = R.Rays.grid(xt, yt, inc=inc_t, hd=torch.full((rows, cols), hd, dtype=torch.float64, device=dev)).look_vectors(ctx)
rays = R.Rays.grid(xt, yt, los=los)
ow = torch.empty((rows, cols), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
_, _, nparts, flags = cube.raytrace(rays, 0.0, zref, out=(ow, oh))
def timed(fn, reps=5):
    fn(); torch.cuda.synchronize()
    ctx.set_profiling(True)
    for _ in range(reps): fn()
    torch.cuda.synchronize()
    r = [ctx.profile_get(i) for i in range(2)]
    ctx.set_profiling(False)
    return {('crossings', 'march')[i]: (r[i][1] / r[i][0] if r[i][0] else None) for i in range(2)}
print(json.dumps({'prepass_only(no workspace stores)': timed(lambda: cube.ray_prepass(rays, 0.0, zref)),
                  'raytrace': timed(lambda: cube.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=False)),
                  'march_only(reusing records)': timed(lambda: cube.ray_march(rays, 0.0, zref, nparts, flags, out=(ow, oh)))}))

# ---- the same scene through an HRRR-like Lambert-conformal-conic cube (projected model coordinates) ----------------
if len(sys.argv) > 2 and sys.argv[2] == 'lcc':
    hr = dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5 - 360.0)
    xs = np.linspace(-2.0e6, -1.0e6, 300); ys = np.linspace(-9.0e5, 1.0e5, 300)
    cube2 = R.Cube(ys, xs, c['zs'], torch.from_numpy(c['wet']).to(dev), torch.from_numpy(c['hydro']).to(dev), order='zyx', ctx=ctx)
    cube2.set_projection_lcc(**hr)
    # lon/lat box well inside that grid
    lon = np.linspace(-118.5, -112.5, cols); lat = np.linspace(31.5, 36.5, rows)
    py, px = cube2.project(np.array([lat[0], lat[-1], lat[0], lat[-1]]), np.array([lon[0], lon[0], lon[-1], lon[-1]]))
    print('corner x', px, 'y', py, 'grid x', xs[[0, -1]], 'y', ys[[0, -1]])
    rays2 = R.Rays.grid(torch.from_numpy(lon).to(dev), torch.from_numpy(lat).to(dev), inc=36.0, hd=-167.9)
    _, _, np2, fl2 = cube2.raytrace(rays2, 0.0, zref, out=(ow, oh))
    print(json.dumps({'lcc raytrace': timed(lambda: cube2.raytrace(rays2, 0.0, zref, out=(ow, oh), want_nparts=False)), 'nan_fraction': float(torch.isnan(ow).double().mean())}))
