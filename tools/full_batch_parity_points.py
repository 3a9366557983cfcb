#!/usr/bin/env python3
"""EVERY point of the two gather workloads of BASELINE against the NumPy oracle's restatement of scipy's RegularGridInterpolator:
  c2: configs[1] - _build_cube on 1000 x 1000 nodes x 40 heights of the 300 x 300 x 80 f64 totals cube (40 M points);
  c5: configs[4] - 5 M random stations on the two-epoch blend of 1000 x 1000 x 50 f32 cubes (LCC grid, stations given in lon / lat:
      the oracle interpolates at ITS OWN projected coordinates).
Records the largest |difference| and the NaN-mask agreement.   usage: full_batch_parity_points.py [out.json]"""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R                      # noqa: E402
from raider_amd import _lib                # noqa: E402
from oracle import raider_oracle as O       # noqa: E402

out_path = sys.argv[1] if len(sys.argv) > 1 else ''
res = dict(source_hash=_lib.source_hash())

# ---- c2
c = O.synthetic_cube(300, 300, 80, seed=0)
tot = R.Cube(c['ys'], c['xs'], c['zs'], c['wet_total'], c['hydro_total'], order='zyx')
xpts = np.linspace(-119.5, -115.5, 1000); ypts = np.linspace(34.5, 31.5, 1000)
zpts = np.linspace(0.0, 3900.0, 40)
t0 = time.perf_counter(); gw, gh = tot.build_cube(xpts, ypts, zpts); t_gpu = time.perf_counter() - t0
it = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet_total'], c['hydro_total']))
worst = [0.0, 0.0]; nanmis = 0; t0 = time.perf_counter()
for k in range(0, 40, 4):
    ow, oh = O.build_cube(xpts, ypts, zpts[k:k + 4], it)
    for j, (g, o) in enumerate(((gw[k:k + 4], ow), (gh[k:k + 4], oh))):
        nanmis += int((np.isnan(g) != np.isnan(o)).sum())
        worst[j] = max(worst[j], float(np.nanmax(np.abs(g - o) / np.abs(o))))
res['c2'] = dict(points=int(gw.size), what='configs[1]: 1000x1000 nodes x 40 heights, f64 totals cube 300x300x80', max_rel_wet=worst[0], max_rel_hydro=worst[1],
                 nan_mask_mismatches=nanmis, gpu_call_s=t_gpu, oracle_s=time.perf_counter() - t0)
print(json.dumps(res['c2']), flush=True)
del tot, gw, gh

# ---- c5
HRRR = dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5 - 360.0, a=6371229.0, es=0.0)
rng = np.random.default_rng(3)
xs = -1.5e6 + 3000.0 * np.arange(1000); ys = -1.5e6 + 3000.0 * np.arange(1000)
zs = np.round(-100 + 26100 * np.linspace(0, 1, 50) ** 2, 3)
hyd0 = 270 * np.exp(-zs / 8000)[:, None, None]; wet0 = 60 * np.exp(-zs / 2000)[:, None, None]
e = [(hyd0 * (1 + 0.01 * rng.standard_normal((1, 1000, 1000)))).astype(np.float32) if k % 2 else
     (wet0 * (1 + 0.1 * rng.standard_normal((1, 1000, 1000)))).astype(np.float32) for k in range(4)]
a = R.Cube(ys, xs, zs, e[0], e[1], order='zyx').set_projection_lcc(**HRRR)
b = R.Cube(ys, xs, zs, e[2], e[3], order='zyx').set_projection_lcc(**HRRR)
m = a.blend(0.25, b, 0.75)
bw = O.blend_cubes(0.25, e[0], 0.75, e[2]); bh = O.blend_cubes(0.25, e[1], 0.75, e[3])
n = 5_000_000
lat = rng.uniform(28.0, 49.0, n); lon = rng.uniform(-110.0, -85.0, n); hgt = rng.uniform(0, 4000, n)
t0 = time.perf_counter()
py, px = m.project(lat, lon)
gw, gh = m.interp(np.stack([py, px, hgt], -1))
t_gpu = time.perf_counter() - t0
pw, ph = a.interp_blend(0.25, b, 0.75, np.stack([py, px, hgt], -1), via_cube=True)        # round 6: the blend made for the gather (paired x columns, scratch)
pair_same = bool(np.array_equal(pw, gw, equal_nan=True) and np.array_equal(ph, gh, equal_nan=True))
del pw, ph
t0 = time.perf_counter()
ox, oy = O.lcc_forward(lat, lon, **HRRR)
iw, ih = O.getInterpolators(xs, ys, zs, bw, bh)
worst = [0.0, 0.0]; nanmis = 0
for s0 in range(0, n, 500000):
    q = np.stack([oy[s0:s0 + 500000], ox[s0:s0 + 500000], hgt[s0:s0 + 500000]], -1)
    for j, (g, f) in enumerate(((gw, iw), (gh, ih))):
        o = f(q); gg = g[s0:s0 + 500000]
        nanmis += int((np.isnan(gg) != np.isnan(o)).sum())
        worst[j] = max(worst[j], float(np.nanmax(np.abs(gg - o))))
res['c5'] = dict(points=n, what='configs[4]: 5 M stations (lon/lat, projected on the device) on the blend of two 1000x1000x50 f32 LCC epochs',
                 max_abs_wet=worst[0], max_abs_hydro=worst[1], max_abs_projection_m=float(max(np.abs(px - ox).max(), np.abs(py - oy).max())),
                 nan_mask_mismatches=nanmis, gpu_call_s=t_gpu, oracle_s=time.perf_counter() - t0, paired_blend_route_same_bits=pair_same,
                 note='the field differences are the projection difference (device libm vs NumPy) times the field gradient')
print(json.dumps(res['c5']), flush=True)
if out_path:
    Path(out_path).parent.mkdir(parents=True, exist_ok=True)
    Path(out_path).write_text(json.dumps(res, indent=1) + '\n')
