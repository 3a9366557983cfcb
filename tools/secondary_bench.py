#!/usr/bin/env python3
"""The secondary kernels of SURVEY 8(a) on the sizes of BASELINE configs 2 and 5, device-resident, each priced with the byte model
of SURVEY 8(d): build_cube_kernel / interp_points_kernel (168 B per point on the f64 totals cube, 104 B on an f32 cube),
blend_kernel (24 B per f32 cell), producer_kernel (32 B per model level + 24 B per output level, per column), orbit_los_fast_kernel
(48 B per target).  Run under `rocprofv3 --kernel-trace --stats` (tools/profile_secondary.sh) - tools/secondary_digest.py joins the
per-kernel durations with these byte counts into profiles/r02_secondary.json.  Prints ONE JSON line."""
import datetime as dt
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R  # noqa: E402
from raider_amd.synthetic import synthetic_cube, scene_grid  # noqa: E402

dev = torch.device('cuda')
ctx = R.Context.default()
ctx.set_stream(torch.cuda.current_stream().cuda_stream)
REPS = 5
res = {}


def timed(fn, reps=REPS):
    fn(); torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


# ---- config 2: _build_cube 1000 x 1000 x 40 heights on the f64 totals cube ---------------------------------------------------
c = synthetic_cube(300, 300, 80, seed=0)
tot = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet_total']).to(dev), torch.from_numpy(c['hydro_total']).to(dev), order='zyx')
x2, y2, _, _ = scene_grid(1000, 1000)
xt, yt = torch.from_numpy(x2).to(dev), torch.from_numpy(y2).to(dev)
zt = torch.from_numpy(c['zs'][:40].copy()).to(dev)
ow = torch.empty((40, 1000, 1000), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
t = timed(lambda: tot.build_cube(xt, yt, zt, out=(ow, oh)))
n = 40 * 1000 * 1000
res['build_cube_kernel'] = dict(what='configs[1]: _build_cube 1000x1000 nodes x 40 heights, 300x300x80 f64 totals cube', units=n, unit='points', bytes_per_unit=168,
                                wall_ms=t * 1e3, reps=REPS + 1)
res['build_cube_setup_kernel'] = dict(what='configs[1]: per-node cells / weights (24 B) and per-height cells / weights of the same _build_cube call (first of its two kernels)',
                                      units=1000 * 1000, unit='nodes', bytes_per_unit=24, wall_ms=0.0, reps=REPS + 1)
# ---- config 5: two-epoch blend of 1000 x 1000 x 50 f32 cubes on the HRRR 3-km LCC grid, 5 M stations ---------------------------
rng = np.random.default_rng(3)
xs = -1.5e6 + 3000.0 * np.arange(1000); ys = -1.5e6 + 3000.0 * np.arange(1000)
zs = np.round(-100 + 26100 * np.linspace(0, 1, 50) ** 2, 3)
e = [torch.from_numpy(rng.standard_normal((50, 1000, 1000)).astype(np.float32)).to(dev) for _ in range(4)]
a = R.Cube(ys, xs, zs, e[0], e[1], order='zyx'); b = R.Cube(ys, xs, zs, e[2], e[3], order='zyx')
del e
t = timed(lambda: a.blend(0.25, b, 0.75))
cells = 50 * 1000 * 1000
res['blend_kernel'] = dict(what='configs[4]: blend of two 1000x1000x50 f32 epochs (2 fields)', units=cells, unit='cells', bytes_per_unit=24, wall_ms=t * 1e3, reps=REPS + 1,
                           note='wall time includes the hipMalloc of the 400 MB result cube; the kernel time is the rocprof figure')
m = a.blend(0.25, b, 0.75)
npt = 5_000_000
pts = torch.from_numpy(np.stack([rng.uniform(-1.4e6, 1.4e6, npt), rng.uniform(-1.4e6, 1.4e6, npt), rng.uniform(0, 4000, npt)], -1)).to(dev)
m.point_index(build=False)
import os  # noqa: E402
os.environ['RAIDER_HIP_POINT_INDEX'] = '0'          # (read once by the library: set before the first large call; the copy is built explicitly below)
t = timed(lambda: m.interp(pts))
res['interp_points_kernel'] = dict(what='configs[4]: 5 M random station points on the blended 1000x1000x50 f32 cube, gathered from the (y,x,z) cube', units=npt, unit='points', bytes_per_unit=104,
                                   wall_ms=t * 1e3, reps=REPS + 1)
# round 6: the same query with the blend made FOR the gather - x columns paired, in scratch (rdr_interp3_blend_cube: blend_pair_kernel + interp_points_pair_kernel)
t = timed(lambda: a.interp_blend(0.25, b, 0.75, pts, via_cube=True))
res['blend_pair_kernel'] = dict(what='configs[4]: blend of two 1000x1000x50 f32 epochs written pair-interleaved into the context scratch', units=cells, unit='cells', bytes_per_unit=24,
                                wall_ms=t * 1e3, reps=REPS + 1, note='wall time is the whole rdr_interp3_blend_cube call (blend + gather)')
res['interp_points_pair_kernel'] = dict(what='configs[4]: 5 M random station points gathered from the pair-interleaved blend (2 lines per point from an even cell, 4 from an odd one)',
                                        units=npt, unit='points', bytes_per_unit=104, wall_ms=t * 1e3, reps=REPS + 1, note='wall time is the whole rdr_interp3_blend_cube call (blend + gather)')
rp = a.interp_blend(0.25, b, 0.75, pts, via_cube=True); rq = m.interp(pts)
res['pair_route_same_bits'] = bool(torch.equal(rp[0], rq[0]) and torch.equal(rp[1], rq[1]))
del rp, rq
tb = timed(lambda: (m.point_index(build=False), m.point_index()), reps=2)
res['quad_build_kernel'] = dict(what='corner-quad copy of the 1000x1000x50 f32 cube (built once per cube)', units=cells, unit='cells', bytes_per_unit=8 + 128.0 / 3.0,
                                wall_ms=tb * 1e3, reps=3, note='wall time includes hipMalloc / hipFree of the 2.2 GB copy')
t = timed(lambda: m.interp(pts))
res['interp_points_quad_kernel'] = dict(what='configs[4]: the same 5 M points gathered from the corner-quad copy (one 128 B line per point)', units=npt, unit='points', bytes_per_unit=104,
                                        wall_ms=t * 1e3, reps=REPS + 1)
# ---- the FIRST large call on a fresh cube, both ways: device time from HIP events around the calls -----------
def first_call(env_mode):
    m2 = a.blend(0.25, b, 0.75)
    torch.cuda.synchronize()
    e0, e1 = torch.cuda.Event(enable_timing=True), torch.cuda.Event(enable_timing=True)
    e0.record()
    if env_mode == 'build':
        m2.point_index()
    r = m2.interp(pts)
    e1.record(); torch.cuda.synchronize()
    return e0.elapsed_time(e1), r
fd = min(first_call('direct')[0] for _ in range(3))
fb = min(first_call('build')[0] for _ in range(3))
rd, rb = first_call('direct')[1], first_call('build')[1]
res['first_call_5M_stations'] = dict(what='ONE rdr_interp3 call of 5 M stations on a FRESH blended 1000x1000x50 f32 cube (device events, best of 3): gathered from the (y,x,z) '
                                          'cube, against building the corner-quad copy first', direct_ms=fd, build_then_quad_ms=fb,
                                     same_bits=bool(torch.equal(rd[0], rb[0]) and torch.equal(rd[1], rb[1])),
                                     policy='time-based (raider_hip.hip quad_wanted): build at the first call only when n x 175 B > the copy\'s bytes (12 M points here)')
del a, b, m, pts
# ---- cube producer: 300 x 300 columns, 137 model levels -> 145 levels -------------------------------------------------------------
from raider_amd.weather import cubes_from_model_levels, MODEL_LEVEL_HEIGHTS  # noqa: E402
A = B = 300; nl = 137
base = np.linspace(0, 1, nl)[None, None, :] ** 1.8
zz = -100.0 + 200.0 * rng.uniform(0, 1, (A, B, 1)) + 80000.0 * base
tt = np.maximum(288.0 - 0.0065 * zz, 200.0); pp = 101325.0 * np.exp(-zz / 7600.0); qq = 0.012 * np.exp(-zz / 2400.0)
arrs = [torch.from_numpy(v).to(dev) for v in (zz, pp, tt, qq)]
newz = np.concatenate([MODEL_LEVEL_HEIGHTS, np.linspace(42000, 80000, 65)])
t = timed(lambda: cubes_from_model_levels(np.linspace(-120, -110, B), np.linspace(30, 40, A), *arrs, 'q', newz))
res['producer_kernel'] = dict(what=f'cube producer: {A}x{B} columns, {nl} model levels -> {newz.size} levels', units=A * B, unit='columns',
                              bytes_per_unit=32 * nl + 24 * int(newz.size), wall_ms=t * 1e3, reps=REPS + 1)
# ---- zero-Doppler look vectors: 4000 x 4000 targets ---------------------------------------------------------------------------------
from raider_amd.orbits import Orbit  # noqa: E402
from raider_amd.utilFcns import lla2ecef  # noqa: E402
ts = np.arange(-120.0, 121.0, 10.0)
r, w = 7.07e6, 2 * np.pi / 5900.0
lat0, lon0 = np.radians(33.0), np.radians(-100.0)
pos = np.stack([r * np.cos(lat0 + w * ts) * np.cos(lon0), r * np.cos(lat0 + w * ts) * np.sin(lon0), r * np.sin(lat0 + w * ts)], -1)
vel = np.stack([-r * w * np.sin(lat0 + w * ts) * np.cos(lon0), -r * w * np.sin(lat0 + w * ts) * np.sin(lon0), r * w * np.cos(lat0 + w * ts)], -1)
epoch = dt.datetime(2021, 1, 1, 6, 57, 0)
orb = Orbit([epoch + dt.timedelta(seconds=float(x)) for x in ts], pos, vel)
nn = 4000
xx, yy = np.meshgrid(np.linspace(-119.5, -115.5, nn), np.linspace(34.5, 31.5, nn))
xyz = torch.from_numpy(np.stack(lla2ecef(yy, xx, np.zeros_like(yy)), -1)).to(dev)
t = timed(lambda: orb.look_vectors(xyz))
res['orbit_los_fast_kernel'] = dict(what='zero-Doppler look vectors of a 4000x4000 scene, 25 state vectors', units=nn * nn, unit='targets', bytes_per_unit=48, wall_ms=t * 1e3, reps=REPS + 1)
print(json.dumps(res))
