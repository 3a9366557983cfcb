#!/usr/bin/env python3
"""A/B of the LDS-staged f64 marcher (round 4): the bench scene through an f64 cube (the f32 refractivities widened - what an
azimuth-time-grid blend leaves on the device), march_kernel<double2,false,1> with RAIDER_HIP_F64_STAGE = 0 / 1 (the library reads the
variable once: one process per setting).  Prints march / crossings ms per step, the shader clock and a digest of the delays.
usage: f64_stage_probe.py [rows cols]   (run it twice: RAIDER_HIP_F64_STAGE=0 and =1)"""
import hashlib
import json
import os
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import raider_amd as R                                         # noqa: E402
from raider_amd.synthetic import scene_grid, synthetic_cube    # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
dev = torch.device('cuda:0')
ctx = R.Context(0)
c = synthetic_cube(300, 300, 80, seed=0)
cube = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet']).to(dev).double(), torch.from_numpy(c['hydro']).to(dev).double(), order='zyx', ctx=ctx)
xpts, ypts, inc_cols, hd = scene_grid(rows, cols)
xt, yt = torch.from_numpy(xpts).to(dev), torch.from_numpy(ypts).to(dev)
inc_t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
los = R.Rays.grid(xt, yt, inc=inc_t, hd=torch.full((rows, cols), hd, dtype=torch.float64, device=dev)).look_vectors(ctx)
rays = R.Rays.grid(xt, yt, los=los)
zref = float(c['zs'].max() - 1)
ow = torch.empty((rows, cols), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
cube.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=True)
for _ in range(2):
    cube.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=False)
torch.cuda.synchronize()
steps = 8
ctx.set_profiling(True)
end = ctx.clock_sample(40.0)
t0 = time.perf_counter()
for _ in range(steps):
    cube.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=False)
torch.cuda.synchronize()
dt = time.perf_counter() - t0
n1, ms1 = ctx.profile_get(0); n2, ms2 = ctx.profile_get(1)
ghz = end()
h = hashlib.sha256(); h.update(ow.cpu().numpy().tobytes()); h.update(oh.cpu().numpy().tobytes())
print(json.dumps(dict(stage=os.environ.get('RAIDER_HIP_F64_STAGE', '(default 1)'), scene=[rows, cols], step_ms=dt / steps * 1e3, march_ms=ms2 / steps, crossings_ms=ms1 / steps,
                      clock_GHz=ghz, mean_hydro=float(oh.mean()), nan=float(torch.isnan(oh).double().mean()), digest=h.hexdigest()[:16],
                      vgpr=cube.ray_kernel_attributes(1)['vgpr'], lds=cube.ray_kernel_attributes(1)['lds_static'])))
