#!/usr/bin/env python3
"""c3b with a SMOOTH DEM (what a real scene has) next to SURVEY 8(d)'s white-noise heights: the per-ray-height marcher idles a lane until the
level loop reaches that lane's first level, so the spread of heights INSIDE a 64-pixel wave is what costs."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import raider_amd as R
from raider_amd.synthetic import synthetic_cube, scene_grid
dev = torch.device('cuda', 0)
c = synthetic_cube(300, 300, 80, seed=0)
cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
zref = float(c['zs'].max() - 1)
n = 4000
xp, yp, inc_cols, hd = scene_grid(n, n)
xt, yt = torch.from_numpy(xp).to(dev), torch.from_numpy(yp).to(dev)
inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (n, n)))).to(dev)
los = R.Rays.grid(xt, yt, inc=inc, hd=hd).look_vectors()
los = los if hasattr(los, 'is_cuda') else torch.from_numpy(los).to(dev)
yy, xx = np.meshgrid(np.arange(n), np.arange(n), indexing='ij')
dems = {'white noise 0..3000 m (SURVEY 8d)': np.random.default_rng(2).uniform(0.0, 3000.0, (n, n)),
        'smooth: 1500 + 1400 sin(x/300) cos(y/400) m': 1500.0 + 1400.0 * np.sin(xx / 300.0) * np.cos(yy / 400.0),
        'flat 0 m (per-ray kernels, equal heights)': np.zeros((n, n))}
w = torch.empty((n, n), dtype=torch.float64, device=dev); h = torch.empty_like(w)
for name, dem in dems.items():
    rays = R.Rays.grid(xt, yt, los=los, hts=torch.from_numpy(np.ascontiguousarray(dem)).to(dev))
    for _ in range(3):
        cube.raytrace(rays, None, zref, out=(w, h), want_nparts=False)
    torch.cuda.synchronize(); t0 = time.perf_counter()
    for _ in range(10):
        cube.raytrace(rays, None, zref, out=(w, h), want_nparts=False)
    torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
    print(f'{name}: {dt * 1e3:.3f} ms per 16 M rays = {n * n / dt / 1e9:.3f} G rays/s')
rays = R.Rays.grid(xt, yt, los=los)
for _ in range(3):
    cube.raytrace(rays, 0.0, zref, out=(w, h), want_nparts=False)
torch.cuda.synchronize(); t0 = time.perf_counter()
for _ in range(10):
    cube.raytrace(rays, 0.0, zref, out=(w, h), want_nparts=False)
torch.cuda.synchronize(); dt = (time.perf_counter() - t0) / 10
print(f'slice kernels at ht = 0: {dt * 1e3:.3f} ms = {n * n / dt / 1e9:.3f} G rays/s')
