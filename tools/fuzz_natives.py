#!/usr/bin/env python3
"""Randomised parity sweep of the remaining entry points against the NumPy oracle: the zenith cube (`_build_cube`, scipy-RGI
semantics incl. outside / last-node / NaN queries), station queries (`Cube.interp`), the two native-extension mirrors
(`interpolate` 1-3 D with / without fill value, `interpolate_along_axis` on every axis) and `makePoints0D..3D`.
usage: fuzz_natives.py [ntrials=100] [seed=0]"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R                                                        # noqa: E402
from oracle import raider_oracle as O                                         # noqa: E402
from raider_amd.interpolate import interpolate, interpolate_along_axis        # noqa: E402
from raider_amd import makePoints as MP                                       # noqa: E402

# the reference's OWN two native extensions, compiled from their sources by oracle/build_ref.sh (checker only): when the binaries
# are there (they travel to the GPU box with the repo), every native-extension trial is also compared with them directly
REFN = None
_so = Path(__file__).resolve().parent.parent / 'oracle' / '_ref' / 'RAiDER'
if any(_so.glob('interpolate*.so')) and 'RAiDER' not in sys.modules:
    import types
    _pkg = types.ModuleType('RAiDER'); _pkg.__path__ = [str(_so)]
    sys.modules['RAiDER'] = _pkg
    try:
        import RAiDER.interpolate as _ri      # noqa: E402
        import RAiDER.makePoints as _rm       # noqa: E402
        REFN = (_ri, _rm)
    except ImportError:
        REFN = None

ntrials = int(sys.argv[1]) if len(sys.argv) > 1 else 100
rng = np.random.default_rng(int(sys.argv[2]) if len(sys.argv) > 2 else 0)
worst = {}
bad = []


def note(name, got, want, tol, tag):
    got, want = np.asarray(got), np.asarray(want)
    if got.shape != want.shape or not np.array_equal(np.isnan(got), np.isnan(want)):
        bad.append(dict(tag, what=name, kind='shape / NaN pattern', got_nan=int(np.isnan(got).sum()), want_nan=int(np.isnan(want).sum())))
        return
    fin = np.isfinite(want)
    d = float(np.max(np.abs(got[fin] - want[fin]) / np.maximum(1.0, np.abs(want[fin])))) if fin.any() else 0.0
    worst[name] = max(worst.get(name, 0.0), d)
    if d > tol:
        bad.append(dict(tag, what=name, kind='value', rel=d))


def axis(n, lo, hi, kind):
    g = np.linspace(lo, hi, n)
    if kind == 'jitter':
        g = g + (hi - lo) / n * 1e-6 * rng.uniform(-1, 1, n)
    elif kind == 'irregular':
        g = np.sort(rng.uniform(lo, hi, n)); g[0], g[-1] = lo, hi
        g = g + np.arange(n) * 1e-9
    return g


for trial in range(ntrials):
    tag = dict(trial=trial)
    # ---- zenith cube and station queries on a random cube -------------------------------------------------------------------
    ny, nx, nz = int(rng.integers(3, 40)), int(rng.integers(3, 40)), int(rng.integers(3, 40))
    kinds = [str(rng.choice(['exact', 'jitter', 'irregular'])) for _ in range(3)]
    ys, xs = axis(ny, 30.0, 36.0, kinds[0]), axis(nx, -121.0, -113.0, kinds[1])
    zs = np.sort(np.round(-100 + 41000 * np.linspace(0, 1, nz) ** 2, 3)) if kinds[2] != 'irregular' else axis(nz, -100.0, 40000.0, 'irregular')
    if rng.random() < 0.25:
        ys = ys[::-1].copy()
    wet = rng.uniform(0, 0.4, (nz, ny, nx)); hyd = rng.uniform(1.0, 2.5, (nz, ny, nx))
    if rng.random() < 0.2:
        wet[rng.random(wet.shape) < 0.02] = np.nan
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx')
    ip = list(O.getInterpolators(xs, ys, zs, wet, hyd))
    tag.update(cube=[ny, nx, nz], axes=kinds)
    xp = np.linspace(-121.5, -112.5, int(rng.integers(2, 30))); yp = np.linspace(36.4, 29.6, int(rng.integers(2, 30)))
    zp = np.concatenate([rng.uniform(-300, 42000, 4), [zs[0], zs[-1], zs[nz // 2]]])
    if rng.random() < 0.3:
        xp[0] = xs[-1]; yp[-1] = ys[0]                                    # exactly on the last / first node
    res = cube.build_cube(xp, yp, zp)
    ow, oh = O.build_cube(xp, yp, zp, ip)
    note('build_cube wet', res[0], ow, 1e-13, tag); note('build_cube hydro', res[1], oh, 1e-13, tag)
    n = int(rng.integers(1, 4000))
    pts = np.stack([rng.uniform(29.5, 36.5, n), rng.uniform(-121.5, -112.5, n), rng.uniform(-300, 42000, n)], -1)
    pts[rng.random(n) < 0.02] = np.nan
    gw, gh = cube.interp(pts)
    note('interp wet', gw, ip[0](pts), 1e-13, tag); note('interp hydro', gh, ip[1](pts), 1e-13, tag)
    # ---- the two-epoch station query (cli/raider.py:817-819 + delay.py:116-121), three routes: the blended cube + gather, the blend at the
    # corners (rdr_interp3_blend), the blend made for the gather with paired x columns (rdr_interp3_blend_cube) - bit-identical to each other,
    # the first against the oracle's f32 / f64 blend arithmetic
    if trial % 3 == 0:
        dt_ = np.float32 if rng.random() < 0.6 else np.float64
        ea = (wet.astype(dt_), hyd.astype(dt_)); eb = (rng.uniform(0, 0.4, wet.shape).astype(dt_), rng.uniform(1.0, 2.5, hyd.shape).astype(dt_))
        w1 = float(rng.choice([0.25, 0.5, 0.6041666666666667])); w2 = 1.0 - w1
        ca = R.Cube(ys, xs, zs, ea[0], ea[1], order='zyx'); cb = R.Cube(ys, xs, zs, eb[0], eb[1], order='zyx')
        r0 = ca.blend(w1, cb, w2).interp(pts)
        ipb = list(O.getInterpolators(xs, ys, zs, O.blend_cubes(w1, ea[0], w2, eb[0]), O.blend_cubes(w1, ea[1], w2, eb[1])))
        note('blend + interp wet', r0[0], ipb[0](pts), 1e-13, tag); note('blend + interp hydro', r0[1], ipb[1](pts), 1e-13, tag)
        for route, via in (('corners', False), ('paired cube', True)):
            r1 = ca.interp_blend(w1, cb, w2, pts, via_cube=via)
            if not (np.array_equal(r0[0], r1[0], equal_nan=True) and np.array_equal(r0[1], r1[1], equal_nan=True)):
                bad.append(dict(tag, what=f'two-epoch query, {route}', kind='not the bits of blend + interp', dtype=str(np.dtype(dt_))))
        worst['two-epoch routes (bit identity)'] = 0.0
    # ---- native `interpolate` ------------------------------------------------------------------------------------------------
    nd = int(rng.integers(1, 6))                       # 1-D ... 5-D (test_interpolator.py goes to 4-D)
    shape = tuple(int(rng.integers(2, 24 if nd < 4 else 9)) for _ in range(nd))
    grids = tuple(axis(s, -1.0, 2.0, str(rng.choice(['exact', 'irregular']))) for s in shape)
    vals = rng.standard_normal(shape)
    q = rng.uniform(-1.4, 2.4, (int(rng.integers(1, 3000)), nd))
    if rng.random() < 0.5:
        q[0] = [g[-1] for g in grids]; q[-1] = [g[0] for g in grids]
    fill = None if rng.random() < 0.5 else float(rng.choice([np.nan, 0.0, -7.5]))
    got_i = interpolate(grids, vals, q, fill_value=fill)
    note(f'interpolate {nd}D', got_i, O.native_interpolate(grids, vals, q, fill_value=fill), 1e-12, dict(tag, nd=nd, fill=str(fill)))
    if REFN:
        note(f'interpolate {nd}D vs compiled reference', got_i, REFN[0].interpolate(grids, vals, q, fill_value=fill, assume_sorted=False, max_threads=2), 0.0, dict(tag, nd=nd, fill=str(fill)))
    # ---- native `interpolate_along_axis` ---------------------------------------------------------------------------------------
    nd = int(rng.integers(1, 4)); ax = int(rng.integers(0, nd))
    shape = [int(rng.integers(2, 12)) for _ in range(nd)]
    shape_q = list(shape); shape_q[ax] = int(rng.integers(1, 9))
    base = np.sort(rng.uniform(0, 10, shape), axis=ax) + np.arange(shape[ax]).reshape([-1 if i == ax else 1 for i in range(nd)]) * 1e-6
    vals = rng.standard_normal(shape)
    qq = rng.uniform(-1, 11, shape_q)
    fill = None if rng.random() < 0.5 else float(rng.choice([np.nan, 0.0]))
    try:
        got = interpolate_along_axis(base, vals, qq, axis=ax, fill_value=fill, max_threads=1)
        note('interpolate_along_axis', got, O.native_interpolate_along_axis(base, vals, qq, axis=ax, fill_value=fill), 1e-12, dict(tag, nd=nd, axis=ax, fill=str(fill)))
        if REFN:
            note('interpolate_along_axis vs compiled reference', got, REFN[0].interpolate_along_axis(base, vals, qq, axis=ax, fill_value=fill, assume_sorted=False, max_threads=1),
                 0.0, dict(tag, nd=nd, axis=ax, fill=str(fill)))
    except Exception as e:
        bad.append(dict(tag, what='interpolate_along_axis', kind=type(e).__name__, msg=str(e)[:200], nd=nd, axis=ax))
    # ---- makePoints ----------------------------------------------------------------------------------------------------------------
    k = int(rng.integers(0, 4))
    lead = tuple(int(rng.integers(1, 7)) for _ in range(k))
    sp = rng.uniform(-6.4e6, 6.4e6, lead + (3,)); slv = rng.standard_normal(lead + (3,))
    max_len = float(rng.uniform(50, 20000)); step = float(rng.choice([15.0, 100.0, 333.3, 1000.0]))
    got = getattr(MP, f'makePoints{k}D')(max_len, sp, slv, step)
    want = O.makePoints(max_len, sp, slv, step)
    if got.shape != want.shape or not np.array_equal(got, want):
        bad.append(dict(tag, what=f'makePoints{k}D', kind='not bit-exact', shape=list(got.shape)))
    if REFN:
        ref = np.asarray(getattr(REFN[1], f'makePoints{k}D')(max_len, sp, slv, step))
        if got.shape != ref.shape or not np.array_equal(got, ref):
            bad.append(dict(tag, what=f'makePoints{k}D vs compiled reference', kind='not bit-exact', shape=list(got.shape), ref_shape=list(ref.shape)))
        worst['makePoints vs compiled reference'] = 0.0
print(json.dumps(dict(trials=ntrials, compiled_reference_natives=bool(REFN), worst_rel=worst, n_bad=len(bad))))
for b in bad[:40]:
    print(json.dumps(b))
