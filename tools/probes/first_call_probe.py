#!/usr/bin/env python3
"""Where does the FIRST call of a big scene go?  100 M rays (configs[3]) traced three times in one process, with the default
workspace limit and with smaller ones (chunked integration), each in a fresh context.  usage: first_call_probe.py [rows cols]"""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import raider_amd as R
from raider_amd.synthetic import synthetic_cube, scene_grid
rows = int(sys.argv[1]) if len(sys.argv) > 1 else 10000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 10000
dev = torch.device('cuda', 0)
c = synthetic_cube(300, 300, 80, seed=0)
zref = float(c['zs'].max() - 1)
xp, yp, inc_cols, hd = scene_grid(rows, cols)
t0 = time.perf_counter(); torch.zeros(1, device=dev); torch.cuda.synchronize(); print(f'torch init {time.perf_counter()-t0:.3f} s')
for limit in (48 << 30, 8 << 30, 2 << 30, 1 << 30):
    ctx = R.Context(0)
    ctx.set_workspace_limit(limit)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx', ctx=ctx)
    xt, yt = torch.from_numpy(xp).to(dev), torch.from_numpy(yp).to(dev)
    inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
    rays = R.Rays.grid(xt, yt, inc=inc, hd=hd)
    wet = torch.empty((rows, cols), dtype=torch.float64, device=dev); hyd = torch.empty_like(wet)
    torch.cuda.synchronize()
    ts = []
    for rep in range(3):
        t0 = time.perf_counter()
        cube.raytrace(rays, 0.0, zref, out=(wet, hyd))
        torch.cuda.synchronize()
        ts.append(time.perf_counter() - t0)
    print(f'workspace limit {limit >> 30:3d} GiB: calls {[round(t * 1e3, 1) for t in ts]} ms')
    del cube, rays, wet, hyd, ctx
    torch.cuda.empty_cache()
