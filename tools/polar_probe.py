#!/usr/bin/env python3
"""GPU ray tracer vs the NumPy oracle on an 80 km cube at high latitude, by latitude band (where the static classification hands
rays from the polynomial kernels to the generic ones).  usage: polar_probe.py [lat_lo=76 lat_hi=86]"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R                      # noqa: E402
from oracle import raider_oracle as O       # noqa: E402

lat_lo, lat_hi = (float(v) for v in sys.argv[1:3]) if len(sys.argv) > 2 else (76.0, 86.0)
rng = np.random.default_rng(4)
c = O.synthetic_cube(90, 120, 40, seed=8, ztop=80000.0, y0=70.0, y1=89.5, x0=-175.0, x1=-95.0)
ypts = np.linspace(lat_hi, lat_lo, 41); xpts = np.linspace(-150.0, -120.0, 24)
inc = rng.uniform(25, 46, (41, 24)); hd = rng.uniform(-180, 180, (41, 24)); ht = 0.0
zref = float(c['zs'].max() - 1)
cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro']))
look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(inc, hd, llh[1], llh[0], llh[2])
(ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([ht]), look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True)
xx, yy = np.meshgrid(xpts, ypts)
los = look(ht, [xx, yy, np.full(yy.shape, ht)], None, yy)
wet, hyd, nparts, _ = cube.raytrace(R.Rays.grid(xpts, ypts, los=np.ascontiguousarray(los)), ht, zref)
print('nparts equal:', np.array_equal(nparts, onp[0]), ' NaN pattern equal:', np.array_equal(np.isnan(hyd), np.isnan(oh[0])), ' finite:', np.isfinite(oh[0]).mean())
gam = (zref - ht) / (np.cos(np.radians(inc)) * 6.3e6)
c0 = np.cos(np.radians(yy))
light = (c0 > gam + 0.02) & (gam < 0.2 * (c0 - gam)) & (gam < 0.035)
for a in np.arange(lat_lo, lat_hi, 1.0):
    m = (yy >= a) & (yy < a + 1.0)
    dw = np.nanmax(np.abs(wet - ow[0])[m]); dh = np.nanmax(np.abs(hyd - oh[0])[m])
    print(f'lat {a:4.0f}-{a + 1:4.0f}: light-path share {light[m].mean():.2f}   max |d wet| {dw:.2e} m   max |d hydro| {dh:.2e} m')

# ---- timing: 2000 x 2000 device-resident scenes through the same cube ------------------------------------------------------
if len(sys.argv) > 3 and sys.argv[3] == 'time':
    import time
    import torch
    dev = torch.device('cuda:0')
    for name, (a, b) in {'78-84 N (light path)': (84.0, 78.0), '86-89 N (generic kernels)': (89.0, 86.0)}.items():
        yp = torch.linspace(a, b, 2000, dtype=torch.float64, device=dev); xp = torch.linspace(-150.0, -120.0, 2000, dtype=torch.float64, device=dev)
        inc_t = (30.0 + 16.0 * torch.arange(2000, dtype=torch.float64, device=dev) / 2000).expand(2000, 2000).contiguous()
        rays = R.Rays.grid(xp, yp, inc=inc_t, hd=-167.9)
        ow_ = torch.empty((2000, 2000), dtype=torch.float64, device=dev); oh_ = torch.empty_like(ow_)
        cube.raytrace(rays, 0.0, zref, out=(ow_, oh_)); torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(5):
            cube.raytrace(rays, 0.0, zref, out=(ow_, oh_), want_nparts=False)
        torch.cuda.synchronize()
        dt_ = (time.perf_counter() - t0) / 5
        print(f'{name}: {dt_ * 1e3:.2f} ms per 4 M rays = {4e6 / dt_ / 1e6:.0f} M rays/s, NaN share {float(torch.isnan(oh_).double().mean()):.3f}')
