#!/usr/bin/env python3
"""Production-sized ray tracing (aria/prepFromGUNW.py:173,180: 20 heights -500..9000 m x ~1e5 rays per date): the height loop of
_build_cube_ray as ONE batched launch pair (rdr_raytrace_slices) against the slice-by-slice loop, device-resident.
usage: bench_slices.py [ny=316] [nx=316] [cube=300x300x80]"""
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R  # noqa: E402
from raider_amd.synthetic import synthetic_cube  # noqa: E402

ny = int(sys.argv[1]) if len(sys.argv) > 1 else 316
nx = int(sys.argv[2]) if len(sys.argv) > 2 else 316
cy, cx, cz = (int(v) for v in (sys.argv[3] if len(sys.argv) > 3 else '300x300x80').split('x'))
dev = torch.device('cuda:0')
ctx = R.Context(0)
c = synthetic_cube(cy, cx, cz, seed=0)
cube = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet']).to(dev), torch.from_numpy(c['hydro']).to(dev), order='zyx', ctx=ctx)
zref = float(c['zs'].max() - 1)
hts = np.arange(-500.0, 9000.0 + 1, 500.0)                       # 20 heights
S = hts.size
xt = torch.linspace(-119.5, -115.5, nx, dtype=torch.float64, device=dev)
yt = torch.linspace(34.5, 31.5, ny, dtype=torch.float64, device=dev)
inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(30.0 + 16.0 * np.arange(nx) / nx, (ny, nx)))).to(dev)
rays = R.Rays.grid(xt, yt, inc=inc, hd=-167.9)
ow = torch.empty((S, ny, nx), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
ow2 = torch.empty_like(ow); oh2 = torch.empty_like(ow)


def batched():
    cube.raytrace_slices(rays, hts, zref, out=(ow, oh), want_partition=False)


def looped():
    for s, ht in enumerate(hts):
        cube.raytrace(rays, float(ht), zref, out=(ow2[s], oh2[s]), want_nparts=False)


def timed(fn, reps=20):
    fn(); fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


tb, tl = timed(batched), timed(looped)
ctx.set_profiling(True); batched(); torch.cuda.synchronize()
kp, km = ctx.profile_get(0), ctx.profile_get(1); ctx.set_profiling(False)
n = S * ny * nx
print(json.dumps({'workload': f'{S} heights x {ny}x{nx} rays = {n} rays, {cy}x{cx}x{cz} cube, device-resident', 'batched_ms': tb * 1e3, 'batched_rays_per_s': n / tb,
                  'slice_loop_ms': tl * 1e3, 'slice_loop_rays_per_s': n / tl, 'speedup': tl / tb,
                  'batched_kernel_ms': {'crossings': kp[1], 'march': km[1]},
                  'bit_identical': bool(torch.equal(ow, ow2) and torch.equal(oh, oh2))}))
