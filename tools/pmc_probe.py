#!/usr/bin/env python3
"""Workload for the HBM-traffic PMC passes (rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate runs):
one calibration copy of known size (1 GiB read + 1 GiB write, 16 B/lane and 8 B/lane variants) followed by
two full config-3 ray-tracing steps.  tools/pmc_summary.py turns the two CSVs into profiles/*.txt."""
import sys
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R  # noqa: E402
from raider_amd.synthetic import synthetic_cube, scene_grid  # noqa: E402

dev = torch.device('cuda')
ctx = R.Context.default()
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
# calibration: 2^27 float64 = 1 GiB; copy_ is a vectorised 16 B/lane stream; complex? keep simple
a = torch.zeros(1 << 27, dtype=torch.float64, device=dev)
b = torch.empty_like(a)
b.copy_(a)
torch.cuda.synchronize()
ny_, nx_, nz_ = (int(v) for v in (sys.argv[1] if len(sys.argv) > 1 else '300x300x80').split('x'))
c = synthetic_cube(ny_, nx_, nz_, seed=0)
cube = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet']).to(dev), torch.from_numpy(c['hydro']).to(dev), order='zyx')
rows = cols = 4000
xpts, ypts, inc_cols, hd = scene_grid(rows, cols)
xt, yt = torch.from_numpy(xpts).to(dev), torch.from_numpy(ypts).to(dev)
inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
hdt = torch.full((rows, cols), hd, dtype=torch.float64, device=dev)
los = R.Rays.grid(xt, yt, inc=inc, hd=hdt).look_vectors(ctx)
rays = R.Rays.grid(xt, yt, los=los)
out = (torch.empty((rows, cols), dtype=torch.float64, device=dev), torch.empty((rows, cols), dtype=torch.float64, device=dev))
zref = float(c['zs'].max() - 1)
for _ in range(2):
    cube.raytrace(rays, 0.0, zref, out=out, want_nparts=False)
torch.cuda.synchronize()
print('done', float(out[1].mean()))
