#!/usr/bin/env python3
"""More than 2^31 rays in ONE call (46400 x 46400 = 2.15e9; outputs 2 x 17 GB, ray records integrated in chunks of the 48 GiB workspace):
index arithmetic past 32 bits everywhere.  Row bands of the result must equal the same bands traced alone with the scene's partition
(ray_prepass / ray_march) bit for bit - first rows, rows around ray 2^31, last rows."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import raider_amd as R
from raider_amd.synthetic import synthetic_cube
n = int(sys.argv[1]) if len(sys.argv) > 1 else 46400
dev = torch.device('cuda', 0)
c = synthetic_cube(300, 300, 80, seed=0)
cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
zref = float(c['zs'].max() - 1)
xt = torch.linspace(-119.5, -115.5, n, dtype=torch.float64, device=dev); yt = torch.linspace(34.5, 31.5, n, dtype=torch.float64, device=dev)
inc = (30.0 + 16.0 * torch.arange(n, dtype=torch.float64, device=dev) / n).expand(n, n).contiguous()
rays = R.Rays.grid(xt, yt, inc=inc, hd=-167.9)
wet = torch.empty((n, n), dtype=torch.float64, device=dev); hyd = torch.empty_like(wet)
torch.cuda.synchronize(); t0 = time.perf_counter()
_, _, nparts, flags = cube.raytrace(rays, 0.0, zref, out=(wet, hyd))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'{n * n:.4e} rays in {dt:.3f} s = {n * n / dt / 1e9:.3f} G rays/s (first call), S = {int(nparts.sum())}, flags = {flags}')
assert bool(torch.isfinite(wet[::97, ::89]).all()) and bool(torch.isfinite(hyd[::97, ::89]).all())
row31 = (1 << 31) // n
ok = True
for r0 in (0, row31 - 8, n - 16):
    sl = slice(r0, r0 + 16)
    band = R.Rays.grid(xt, yt[sl].contiguous(), inc=inc[sl].contiguous(), hd=-167.9)
    ml, fl = cube.ray_prepass(band, 0.0, zref)
    w, h = cube.ray_march(band, 0.0, zref, nparts, flags)
    same = torch.equal(w, wet[sl]) and torch.equal(h, hyd[sl])
    print(f'rows {r0}..{r0 + 15} (rays {r0 * n:.4e}..): identical = {same}')
    ok = ok and same
# and the far corner against a tiny scene of its own
print('all bands identical:', ok)
sys.exit(0 if ok else 1)
