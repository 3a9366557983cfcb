#!/usr/bin/env python3
"""More than 2^31 points / nodes in ONE call of the gather entries (device-resident arrays): Cube.interp on 2.2e9 packed points and
Cube.build_cube on a 12000 x 12000 x 16 node grid (2.3e9 nodes).  Samples around element 2^31 and at the end must equal small calls."""
import sys, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import torch
import raider_amd as R
from raider_amd.synthetic import synthetic_cube
dev = torch.device('cuda', 0)
c = synthetic_cube(300, 300, 80, seed=0)
cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet_total'], c['hydro_total'], order='zyx')
ok = True
# ---- interp: 2.2e9 points made on the device
n = 2_200_000_000
g = torch.Generator(device=dev); g.manual_seed(1)
pts = torch.empty((n, 3), dtype=torch.float64, device=dev)
step = 100_000_000
for s0 in range(0, n, step):
    u = torch.rand((min(step, n - s0), 3), dtype=torch.float64, device=dev, generator=g)
    pts[s0:s0 + u.shape[0], 0] = 30.2 + 5.6 * u[:, 0]; pts[s0:s0 + u.shape[0], 1] = -120.8 + 7.6 * u[:, 1]; pts[s0:s0 + u.shape[0], 2] = 9000.0 * u[:, 2]
    del u
torch.cuda.synchronize(); t0 = time.perf_counter()
w, h = cube.interp(pts)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'interp: {n:.3e} points in {dt:.3f} s = {n / dt / 1e9:.2f} G points/s (first call: includes the allocation of 35 GB of outputs); finite {bool(torch.isfinite(w[::1000003]).all())}')
del w, h
torch.cuda.synchronize(); t0 = time.perf_counter()
w, h = cube.interp(pts)          # (the outputs come out of torch's cache now)
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'interp again: {dt:.3f} s = {n / dt / 1e9:.2f} G points/s')
for a in (0, (1 << 31) - 500, n - 1000):
    w2, h2 = cube.interp(pts[a:a + 1000].contiguous())
    same = torch.equal(w2, w[a:a + 1000]) and torch.equal(h2, h[a:a + 1000])
    print(f'  points {a}..: identical = {same}'); ok = ok and same
del pts, w, h
torch.cuda.empty_cache()
# ---- build_cube: 12000 x 12000 nodes x 16 heights
nx = ny = 12000
xp = np.linspace(-119.5, -115.5, nx); yp = np.linspace(34.5, 31.5, ny); zp = np.linspace(0.0, 3000.0, 16)
ow = torch.empty((16, ny, nx), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
torch.cuda.synchronize(); t0 = time.perf_counter()
cube.build_cube(torch.from_numpy(xp).to(dev), torch.from_numpy(yp).to(dev), torch.from_numpy(zp).to(dev), out=(ow, oh))
torch.cuda.synchronize(); dt = time.perf_counter() - t0
print(f'build_cube: {16 * nx * ny:.3e} nodes in {dt:.3f} s = {16 * nx * ny / dt / 1e9:.1f} G nodes/s')
for iz, r0 in ((0, 0), (14, 10000), (15, ny - 8)):          # (14*12000+10000)*12000 = 2.136e9 .. past 2^31 inside the band
    sw, sh = cube.build_cube(xp, yp[r0:r0 + 8], zp[iz:iz + 1])
    same = np.array_equal(sw[0], ow[iz, r0:r0 + 8].cpu().numpy()) and np.array_equal(sh[0], oh[iz, r0:r0 + 8].cpu().numpy())
    print(f'  height {iz}, rows {r0}..: identical = {same}'); ok = ok and same
print('all identical:', ok)
sys.exit(0 if ok else 1)
