#!/usr/bin/env python3
"""Secondary measurements quoted in DESIGN.md (NOT the driver's bench): PCIe-inclusive ray tracing through the
host-buffer (NumPy) boundary, config 2 (zenith/projected cube, 1000x1000), config 5-like (two-epoch blend + 5 M
station points on a 1000x1000x50 cube), makePoints and the native interpolate.  Prints one JSON line each."""
import json
import sys
import time
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))

import torch  # noqa: E402
import raider_amd as R  # noqa: E402
from raider_amd.synthetic import synthetic_cube, scene_grid  # noqa: E402


def timeit(fn, reps=3):
    fn()
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(reps):
        fn()
    torch.cuda.synchronize()
    return (time.perf_counter() - t0) / reps


def main():
    ctx = R.Context.default()
    c = synthetic_cube(300, 300, 80, seed=0)
    zref = float(c['zs'].max() - 1)
    # ---- host-buffer boundary, config 3 geometry (PCIe inclusive) ------------------------------------------
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    rows = cols = 4000
    xpts, ypts, inc_cols, hd = scene_grid(rows, cols)
    inc = np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))
    los = R.Rays.grid(xpts, ypts, inc=inc, hd=np.full((rows, cols), hd)).look_vectors()     # NumPy (rows, cols, 3)
    out = (np.empty((rows, cols)), np.empty((rows, cols)))
    dt = timeit(lambda: cube.raytrace(R.Rays.grid(xpts, ypts, los=los), 0.0, zref, out=out), reps=2)
    print(json.dumps({'what': 'config 3 through the NumPy (host buffer) boundary: H2D look vectors 384 MB + kernels + D2H 256 MB',
                      'rays_per_s': rows * cols / dt, 'ms': dt * 1e3}))
    dt = timeit(lambda: cube.raytrace(R.Rays.grid(xpts, ypts, inc=inc_cols[None, :].repeat(1, 0)[0].mean(), hd=hd), 0.0, zref, out=out), reps=2)
    print(json.dumps({'what': 'same, scalar inc/heading (no look-vector upload, D2H 256 MB only)', 'rays_per_s': rows * cols / dt, 'ms': dt * 1e3}))
    # ---- config 2: Conventional slant = ZTD gather on the f64 totals cube, then / cos(inc) ------------------
    tot = R.Cube(c['ys'], c['xs'], c['zs'], c['wet_total'], c['hydro_total'], order='zyx')
    x2, y2, _, _ = scene_grid(1000, 1000)
    dev = torch.device('cuda')
    xt, yt = torch.from_numpy(x2).to(dev), torch.from_numpy(y2).to(dev)
    zt = torch.from_numpy(c['zs'][:40].copy()).to(dev)
    ow = torch.empty((40, 1000, 1000), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    dt = timeit(lambda: tot.build_cube(xt, yt, zt, out=(ow, oh)), reps=5)
    npts = 40 * 1000 * 1000
    print(json.dumps({'what': 'config 2: _build_cube 1000x1000 x 40 heights on the f64 totals cube (device resident)',
                      'points_per_s': npts / dt, 'ms': dt * 1e3, 'algorithmic_GBps': npts * 168 / dt / 1e9}))
    # ---- config 5-like: two-epoch blend + 5 M station points on a 1000x1000x50 cube -----------------------
    rng = np.random.default_rng(3)
    ys = np.linspace(30, 45, 1000); xs = np.linspace(-125, -100, 1000); zs = np.round(-100 + 26100 * np.linspace(0, 1, 50) ** 2, 3)
    e = [(rng.standard_normal((50, 1000, 1000)).astype(np.float32)) for _ in range(4)]
    a = R.Cube(ys, xs, zs, e[0], e[1], order='zyx'); b = R.Cube(ys, xs, zs, e[2], e[3], order='zyx')
    dtb = timeit(lambda: a.blend(0.25, b, 0.75), reps=3)
    m = a.blend(0.25, b, 0.75)
    n = 5_000_000
    pts = torch.from_numpy(np.stack([rng.uniform(30.5, 44.5, n), rng.uniform(-124, -101, n), rng.uniform(0, 4000, n)], -1)).to(dev)
    dti = timeit(lambda: m.interp(pts), reps=5)
    print(json.dumps({'what': 'config 5-like: blend of two 1000x1000x50 f32 epochs (800 MB in, 400 MB out)', 'ms': dtb * 1e3,
                      'GBps': 1.2e9 / dtb / 1e9}))
    print(json.dumps({'what': 'config 5-like: 5 M random station points, trilinear gather of both fields (device resident)',
                      'points_per_s': n / dti, 'ms': dti * 1e3}))




def producer_bench():
    """cube producer: ERA5-like 300x300 columns x 137 model levels -> 145 output levels"""
    from raider_amd.weather import cubes_from_model_levels, MODEL_LEVEL_HEIGHTS
    rng = np.random.default_rng(0)
    A = B = 300; nl = 137
    base = np.linspace(0, 1, nl)[None, None, :] ** 1.8
    zs = -100.0 + 200.0 * rng.uniform(0, 1, (A, B, 1)) + 80000.0 * base
    t = np.maximum(288.0 - 0.0065 * zs, 200.0); p = 101325.0 * np.exp(-zs / 7600.0); q = 0.012 * np.exp(-zs / 2400.0)
    dev = torch.device('cuda')
    arrs = [torch.from_numpy(a).to(dev) for a in (zs, p, t, q)]
    xs = np.linspace(-120, -110, B); ys = np.linspace(30, 40, A)
    newz = np.concatenate([MODEL_LEVEL_HEIGHTS, np.linspace(42000, 80000, 65)])
    fn = lambda: cubes_from_model_levels(xs, ys, *arrs, 'q', newz)
    dt = timeit(fn, reps=5)
    cols = A * B
    print(json.dumps({'what': f'cube producer {A}x{B} columns, {nl} model levels -> {newz.size} levels (device-resident inputs)', 'ms': dt * 1e3,
                      'columns_per_s': cols / dt, 'GBps_in_plus_out': cols * (32 * nl + 24 * newz.size) / dt / 1e9}))


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'producer':
        producer_bench()
    else:
        main()
        producer_bench()
