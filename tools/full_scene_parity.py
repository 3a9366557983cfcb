#!/usr/bin/env python3
"""EVERY ray of a BASELINE-sized scene against the multi-core C oracle (oracle/oracle_c.c), not a sample of blocks: the GPU traces
the whole scene once, the oracle re-traces it in row blocks driven with the whole-slice partition (delay.py:283), and the tool
records the largest |difference| of both delays, the NaN-mask agreement and the nParts agreement.  Not part of the suite (the
oracle needs 30 s for 16 M rays and 3 min for 100 M on the GPU box's 16 usable cores).
usage: full_scene_parity.py [rows=4000] [cols=4000] [out.json] [c5|c3b]
c3b: SURVEY 8(d)'s secondary workload - every pixel starts at its own height, rng(2).uniform(0, 3000) - against the oracle's per-ray
restatement of the rule in DESIGN.md 5c (no reference semantics).
c5: the ray scene of BASELINE configs[4] instead - an HRRR-like 1000 x 1000 x 50 cube on the 3-km Lambert-conformal-conic grid, two
epochs blended (0.25, 0.75) on the device (f32), lon / lat scene over the central US; the oracle projects every sample as delay.py:253,295."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R                      # noqa: E402
from raider_amd import _lib                # noqa: E402
from oracle import raider_oracle as O       # noqa: E402
from oracle import oracle_c as OC           # noqa: E402
from raider_amd.synthetic import scene_grid    # noqa: E402

rows = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
cols = int(sys.argv[2]) if len(sys.argv) > 2 else 4000
out_path = sys.argv[3] if len(sys.argv) > 3 else ''
C5 = len(sys.argv) > 4 and sys.argv[4] == 'c5'
PP = len(sys.argv) > 4 and sys.argv[4] == 'c3b'
HRRR = dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5 - 360.0, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0)       # models/hrrr.py:248-259

import torch                                # noqa: E402
dev = torch.device('cuda', 0)
if C5:
    rng = np.random.default_rng(3)
    xs = -1.5e6 + 3000.0 * np.arange(1000); ys = -1.5e6 + 3000.0 * np.arange(1000)
    zs = np.round(-100 + 26100 * np.linspace(0, 1, 50) ** 2, 3)
    hyd0 = 270 * np.exp(-zs / 8000)[:, None, None]; wet0 = 60 * np.exp(-zs / 2000)[:, None, None]
    e = [(hyd0 * (1 + 0.01 * rng.standard_normal((1, 1000, 1000)))).astype(np.float32) if k % 2 else
         (wet0 * (1 + 0.1 * rng.standard_normal((1, 1000, 1000)))).astype(np.float32) for k in range(4)]      # wet_a, hydro_a, wet_b, hydro_b
    a = R.Cube(ys, xs, zs, e[0], e[1], order='zyx').set_projection_lcc(**HRRR)
    b = R.Cube(ys, xs, zs, e[2], e[3], order='zyx').set_projection_lcc(**HRRR)
    cube = a.blend(0.25, b, 0.75)
    c = dict(ys=ys, xs=xs, zs=zs, wet=O.blend_cubes(0.25, e[0], 0.75, e[2]), hydro=O.blend_cubes(0.25, e[1], 0.75, e[3]))   # cli/raider.py:817-819 (f32)
    del a, b, e
    xpts = np.linspace(-104.0, -92.0, cols); ypts = np.linspace(44.0, 33.0, rows)
    inc_cols = 30.0 + 16.0 * (np.arange(cols) / float(cols)); hd = -167.9
else:
    c = O.synthetic_cube(300, 300, 80, seed=0)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    xpts, ypts, inc_cols, hd = scene_grid(rows, cols)
zref = float(c['zs'].max() - 1.0)
proj = HRRR if C5 else None
xt, yt = torch.from_numpy(xpts).to(dev), torch.from_numpy(ypts).to(dev)
inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
hts_np = np.random.default_rng(2).uniform(0.0, 3000.0, (rows, cols)) if PP else None
rays = R.Rays.grid(xt, yt, inc=inc, hd=hd, hts=torch.from_numpy(hts_np).to(dev) if PP else None)
wet = torch.empty((rows, cols), dtype=torch.float64, device=dev); hyd = torch.empty_like(wet)
t0 = time.perf_counter()
_, _, nparts, flags = cube.raytrace(rays, None if PP else 0.0, zref, out=(wet, hyd))
torch.cuda.synchronize()
t_gpu = time.perf_counter() - t0
wn, hn = wet.cpu().numpy(), hyd.cpu().numpy()
del wet, hyd

# the oracle's OWN partition of the whole scene (pass 1 over every ray), then its march block by block with it
block = max(1, (1 << 21) // cols)
t0 = time.perf_counter()
maxlen = clamp = None
worst_w = worst_h = 0.0
nan_mismatch = 0
sum_w = sum_h = 0.0
los_of = lambda r0, r1: O.look_vectors_from_inc_hd(np.broadcast_to(inc_cols, (r1 - r0, cols)), np.full((r1 - r0, cols), hd),
                                                   *np.meshgrid(ypts[r0:r1], xpts, indexing='ij'), 0.0)
ll_of = lambda r0, r1: np.meshgrid(ypts[r0:r1], xpts, indexing='ij')
for r0 in range(0, rows, block):            # pass 1: per-level maxima / clamp predicates of every block -> the scene's
    r1 = min(rows, r0 + block)
    if PP:
        ml, cl = OC.per_pixel_prepass(c, *ll_of(r0, r1), hts_np[r0:r1], los_of(r0, r1), zref)
    else:
        ml, cl = OC.ray_prepass(c, xpts, ypts[r0:r1], 0.0, los_of(r0, r1), zref)
    maxlen = ml if maxlen is None else np.maximum(maxlen, ml)
    clamp = cl if clamp is None else (clamp[0] & cl[0], clamp[1] & cl[1])
t_pass1 = time.perf_counter() - t0
if PP:          # the oracle's partition is indexed by model interval, the library's by entry of the batch's level table
    onp = OC.per_pixel_nparts(maxlen)
    kz = cube.ray_levels(float(hts_np.min()), zref)[2]
    mask = np.zeros(onp.size, bool); mask[kz] = True
    nparts_equal = bool(np.array_equal(onp[kz], nparts) and not onp[~mask].any())
else:
    onp = OC.nparts_of(maxlen)
    nparts_equal = bool(np.array_equal(onp, nparts))
t0 = time.perf_counter()
for r0 in range(0, rows, block):
    r1 = min(rows, r0 + block)
    if PP:
        ow, oh, _ = OC.build_cube_ray_per_pixel(c, *ll_of(r0, r1), hts_np[r0:r1], los_of(r0, r1), zref, nparts=onp, clamp=clamp)
    else:
        ow, oh, _ = OC.build_cube_ray_slice(c, xpts, ypts[r0:r1], 0.0, los_of(r0, r1), zref, nparts=onp, clamp=clamp, model_proj=proj)
    gw, gh = wn[r0:r1], hn[r0:r1]
    nan_mismatch += int((np.isnan(ow) != np.isnan(gw)).sum() + (np.isnan(oh) != np.isnan(gh)).sum())
    with np.errstate(invalid='ignore'):
        worst_w = max(worst_w, float(np.nanmax(np.abs(gw - ow)))); worst_h = max(worst_h, float(np.nanmax(np.abs(gh - oh))))
    sum_w += float(np.nansum(ow)); sum_h += float(np.nansum(oh))
t_march = time.perf_counter() - t0
res = dict(scene=f'{rows}x{cols}' + (' on a DEM: per-pixel origin heights rng(2).uniform(0, 3000) (c3b)' if PP else ''), rays=rows * cols, cube='1000x1000x50 f32 LCC, two epochs blended (configs[4])' if C5 else '300x300x80 f32 (SURVEY 8d, seed 0)', S=int(np.sum(nparts)), K=int(len(nparts)),
           nparts_equal=nparts_equal, max_abs_wet_m=worst_w, max_abs_hydro_m=worst_h, nan_mask_mismatches=nan_mismatch,
           gpu_mean_wet_m=float(np.nanmean(wn)), gpu_mean_hydro_m=float(np.nanmean(hn)), oracle_mean_wet_m=sum_w / (rows * cols),
           oracle_mean_hydro_m=sum_h / (rows * cols), gpu_call_s=t_gpu, oracle_pass1_s=t_pass1, oracle_march_s=t_march,
           oracle_threads=OC.num_threads(), tolerance_m=1e-6, source_hash=_lib.source_hash())
print(json.dumps(res))
if out_path:
    Path(out_path).parent.mkdir(parents=True, exist_ok=True)
    Path(out_path).write_text(json.dumps(res, indent=1) + '\n')
ok = nparts_equal and nan_mismatch == 0 and max(worst_w, worst_h) < 1e-6
sys.exit(0 if ok else 1)
