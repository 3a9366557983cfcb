#!/usr/bin/env python3
"""Device / pinned-host memory after N create-use-destroy cycles of contexts and cubes (every entry family once per cycle)."""
import gc, sys
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import raider_amd as R
from raider_amd import _pinned
from raider_amd.synthetic import synthetic_cube
N = int(sys.argv[1]) if len(sys.argv) > 1 else 100
c = synthetic_cube(120, 130, 40, seed=0)
zref = float(c['zs'].max() - 1)
xp = np.linspace(-119.0, -116.0, 700); yp = np.linspace(34.0, 32.0, 600)
rng = np.random.default_rng(0)
pts = np.stack([rng.uniform(31, 35, 300000), rng.uniform(-120, -115, 300000), rng.uniform(0, 5000, 300000)], -1)
def cycle():
    ctx = R.Context(0)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx', ctx=ctx)
    w, h, npz, _ = cube.raytrace(R.Rays.grid(xp, yp, inc=35.0, hd=-167.9), 0.0, zref)
    ws, hs, K, nps, fl = cube.raytrace_slices(R.Rays.grid(xp, yp, inc=35.0, hd=-167.9), np.array([0.0, 500.0, 2000.0]), zref)
    cube.point_index(True); a, b = cube.interp(pts)
    tot = R.Cube(c['ys'], c['xs'], c['zs'], c['wet_total'], c['hydro_total'], order='zyx', ctx=ctx)
    zw, zh = tot.build_cube(xp, yp, np.array([0.0, 100.0, 1000.0]))
    m = cube.blend(0.25, cube, 0.75)
    # round 4: the point branch (fused / two-call / ray-traced), the on-the-fly blend
    pw, ph, pn = tot.point_delays(xp, yp, np.array([0.0, 100.0, 1000.0, 4000.0]), pts[:, 0], pts[:, 1], pts[:, 2], inc=39.0)
    d = tot.build_delay_cube(xp, yp, np.array([0.0, 100.0, 1000.0, 4000.0])); dw, dh = d.interp_project(pts)
    rc, K2, n2, f2 = cube.raytrace_slices_to_cube(R.Rays.grid(xp, yp, inc=35.0, hd=-167.9), np.array([0.0, 500.0, 2000.0]), zref)
    bw, bh = cube.interp_blend(0.25, cube, 0.75, pts)
    # round 5: views (source destroyed first), device-source cubes made asynchronously with pending verdicts never asked for, trim
    v = cube.view(dict(proj='lcc', lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0)); v2 = v.view(None)
    dev = torch.device('cuda:0')
    wt, ht = torch.from_numpy(c['wet']).to(dev), torch.from_numpy(c['hydro']).to(dev)
    asy = [R.Cube(c['ys'], c['xs'], c['zs'], wt, ht, order='zyx', ctx=ctx) for _ in range(3)]
    asy[0].has_nan()
    va, vb = v2.interp(pts)
    del cube
    vc = v.interp(pts)
    ctx.trim(1 << 20)
    del tot, m, d, rc, v, v2, asy, ctx
for i in range(3): cycle()
gc.collect(); torch.cuda.synchronize()
free0 = torch.cuda.mem_get_info()[0]
import resource
rss0 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
for i in range(N): cycle()
gc.collect(); torch.cuda.synchronize()
free1 = torch.cuda.mem_get_info()[0]
rss1 = resource.getrusage(resource.RUSAGE_SELF).ru_maxrss
print(f'{N} cycles: device free {free0 >> 20} -> {free1 >> 20} MiB (delta {(free0 - free1) / (1 << 20):.1f} MiB), max RSS {rss0 >> 10} -> {rss1 >> 10} MiB, '
      f'pinned pool free {_pinned.free_bytes() >> 20} MiB')
