"""GPU: ray batches whose rays start at their OWN heights (a SAR scene on a DEM; SURVEY 8(d) "c3b").  The reference integrates one
slice at one height (delay.py:256-323); the rule for per-ray heights (include/raider_hip.h rdr_rays.hts) is its slice
algorithm ray by ray, with the slice-level reductions (nParts, the all-pixels z-clamp) kept batch-level.  Pinned two ways:
  * equal heights reproduce the slice kernels' result BIT FOR BIT (every input form, conic cubes, generic rays, f64 cubes);
  * mixed heights agree with the oracle's restatement (oracle_c.build_cube_ray_per_pixel, itself reduced to the reference goldens
    in tests/test_oracle_c.py) to 1e-9 m with the same partition."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import raider_oracle as O
from oracle import oracle_c as OC

TIGHT = 1e-9


@pytest.fixture(scope='module')
def R():
    import raider_amd
    return raider_amd


@pytest.fixture(scope='module')
def c1():
    return O.synthetic_cube(50, 50, 40, seed=0)


def _table_zz(R, cube, ht, zref):
    """model interval of every entry of the level table built for height ht"""
    return cube.ray_levels(ht, zref)[2]


def _scene(ny=21, nx=27):
    xp = np.linspace(-119.0, -115.5, nx); yp = np.linspace(34.6, 31.4, ny)
    xx, yy = np.meshgrid(xp, yp)
    inc = 30.0 + 16.0 * (np.arange(nx) / nx)[None, :] + 0.0 * yy
    return xp, yp, xx, yy, inc


@pytest.mark.parametrize('ht', [0.0, 437.5, -60.0, 2500.0])
def test_equal_heights_are_the_slice_bit_for_bit(R, c1, ht):
    cube = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet'], c1['hydro'], order='zyx')
    zref = float(c1['zs'].max() - 1)
    xp, yp, xx, yy, inc = _scene()
    hts = np.full(yy.shape, ht)
    los = O.look_vectors_from_inc_hd(inc, np.full(yy.shape, -167.9), yy, xx, ht)
    ws, hs, nps, fs = cube.raytrace(R.Rays.grid(xp, yp, los=los), ht, zref)
    # GRID + look vectors
    w, h, npp, fp = cube.raytrace(R.Rays.grid(xp, yp, los=los, hts=hts), None, zref)
    assert np.array_equal(npp, nps) and fp == fs
    assert np.array_equal(w, ws) and np.array_equal(h, hs) and np.isfinite(h).all()
    # GRID + incidence / heading rasters, and scalars
    ws2, hs2, _, _ = cube.raytrace(R.Rays.grid(xp, yp, inc=inc, hd=-167.9), ht, zref)
    w2, h2, _, _ = cube.raytrace(R.Rays.grid(xp, yp, inc=inc, hd=-167.9, hts=hts), None, zref)
    assert np.array_equal(w2, ws2) and np.array_equal(h2, hs2)
    ws3, hs3, _, _ = cube.raytrace(R.Rays.grid(xp, yp, inc=39.0, hd=-167.9), ht, zref)
    w3, h3, _, _ = cube.raytrace(R.Rays.grid(xp, yp, inc=39.0, hd=-167.9, hts=hts), None, zref)
    assert np.array_equal(w3, ws3) and np.array_equal(h3, hs3)
    # point list (LLH) and ECEF origins
    ws4, hs4, _, _ = cube.raytrace(R.Rays.points(lat=yy.ravel(), lon=xx.ravel(), los=los.reshape(-1, 3)), ht, zref)
    w4, h4, _, _ = cube.raytrace(R.Rays.points(lat=yy.ravel(), lon=xx.ravel(), los=los.reshape(-1, 3), hts=hts.ravel()), None, zref)
    assert np.array_equal(w4, ws4) and np.array_equal(h4, hs4)
    np.testing.assert_allclose(w4.reshape(yy.shape), ws, rtol=0, atol=1e-12)      # (GRID tiles share their sines / cosines: not the same bits)
    xyz = np.stack(O.lla2ecef(yy.ravel(), xx.ravel(), hts.ravel()), -1)
    w5, h5, _, _ = cube.raytrace(R.Rays.points(xyz=xyz, los=los.reshape(-1, 3), hts=hts.ravel()), None, zref)
    ws5, hs5, _, _ = cube.raytrace(R.Rays.points(xyz=xyz, los=los.reshape(-1, 3)), ht, zref)
    assert np.array_equal(w5, ws5) and np.array_equal(h5, hs5)
    # device arrays, asynchronous; and the split prepass / march pair with a device partition (what multi-GPU slabs use)
    import torch
    dev = torch.device('cuda:0')
    t = lambda a: torch.from_numpy(np.ascontiguousarray(a)).to(dev)
    rays_d = R.Rays.grid(t(xp), t(yp), los=t(los), hts=t(hts))
    assert rays_d.ht_min == ht
    wd, hd_, _, _ = cube.raytrace(rays_d, None, zref, want_nparts=False)
    part = torch.zeros(len(nps) + 4, dtype=torch.float64, device=dev)
    cube.ray_prepass_device(rays_d, None, zref, part)
    wd2, hd2 = cube.ray_march_device(rays_d, None, zref, part)
    torch.cuda.synchronize()
    assert np.array_equal(wd.cpu().numpy(), ws) and np.array_equal(hd_.cpu().numpy(), hs)
    assert np.array_equal(wd2.cpu().numpy(), ws) and np.array_equal(hd2.cpu().numpy(), hs)
    ml, fl = cube.ray_prepass(R.Rays.grid(xp, yp, los=los, hts=hts), None, zref)
    w6, h6 = cube.ray_march(R.Rays.grid(xp, yp, los=los, hts=hts), None, zref, R.nparts_from_maxlen(ml), fl)
    assert np.array_equal(w6, ws) and np.array_equal(h6, hs)


def test_equal_heights_conic_cube_generic_rays_and_f64_cube(R):
    # Lambert-conformal-conic cube (HRRR-like sphere)
    rng = np.random.default_rng(4)
    lcc = dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5)
    xs = -4.0e5 + 9000.0 * np.arange(60); ys = -3.0e5 + 9000.0 * np.arange(50); zs = np.round(-100 + 26100 * np.linspace(0, 1, 30) ** 2, 3)
    wet = (60 * np.exp(-zs / 2000.0)[:, None, None] * (1 + 0.1 * rng.standard_normal((1, 50, 60)))).astype(np.float32)
    hyd = (270 * np.exp(-zs / 8000.0)[:, None, None] * (1 + 0.01 * rng.standard_normal((1, 50, 60)))).astype(np.float32)
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx').set_projection_lcc(**lcc)
    zref = float(zs.max() - 1)
    yp = np.linspace(39.5, 37.0, 19); xp = np.linspace(-100.5, -96.5, 23)
    for ht in (120.0, 1500.0):
        s = cube.raytrace(R.Rays.grid(xp, yp, inc=35.0, hd=-12.1), ht, zref)
        p = cube.raytrace(R.Rays.grid(xp, yp, inc=35.0, hd=-12.1, hts=np.full((19, 23), ht)), None, zref)
        assert np.array_equal(s[2], p[2]) and np.array_equal(s[0], p[0]) and np.array_equal(s[1], p[1]) and np.isfinite(s[1]).mean() > 0.9
    # generic (polar) rays + an f64 cube
    c = O.synthetic_cube(40, 60, 30, seed=2, ztop=30000.0, y0=80.0, y1=89.8, x0=-60.0, x1=60.0)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'].astype(np.float64), c['hydro'].astype(np.float64), order='zyx')
    zref = float(c['zs'].max() - 1)
    yp = np.linspace(89.0, 86.5, 12); xp = np.linspace(-20.0, 20.0, 14)
    ctx = R.Context.default()
    for ht in (0.0, 900.0):
        s = cube.raytrace(R.Rays.grid(xp, yp, inc=33.0, hd=10.0), ht, zref)
        n_generic = ctx.generic_ray_count()
        p = cube.raytrace(R.Rays.grid(xp, yp, inc=33.0, hd=10.0, hts=np.full((12, 14), ht)), None, zref)
        assert n_generic > 50 and ctx.generic_ray_count() == n_generic
        assert np.array_equal(s[2], p[2]) and np.array_equal(s[0], p[0], equal_nan=True) and np.array_equal(s[1], p[1], equal_nan=True)
        assert np.isfinite(s[1]).mean() > 0.5


def _oracle_vs_gpu(R, c, lat, lon, hts, los, zref, max_seg=1000.0, tol=TIGHT, shape=None):
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    ow, oh, onp = OC.build_cube_ray_per_pixel(c, lat, lon, hts, los, zref, max_seg=max_seg)
    rays = R.Rays.points(lat=lat.ravel(), lon=lon.ravel(), los=los.reshape(-1, 3), hts=hts.ravel())
    w, h, npp, fl = cube.raytrace(rays, None, zref, max_seg=max_seg)
    kz = _table_zz(R, cube, rays.ht_min, zref)
    assert np.array_equal(npp, onp[kz]), (npp, onp[kz])
    unused = np.setdiff1d(np.arange(onp.size), kz)
    assert not onp[unused].any()
    assert np.array_equal(np.isnan(w), np.isnan(ow.ravel()))
    np.testing.assert_allclose(w, ow.ravel(), rtol=0, atol=tol, equal_nan=True)
    np.testing.assert_allclose(h, oh.ravel(), rtol=0, atol=tol, equal_nan=True)
    return w, h, npp


def test_mixed_heights_against_the_oracle(R, c1):
    rng = np.random.default_rng(12)
    n = 4000
    lat = rng.uniform(31.0, 35.0, n); lon = rng.uniform(-120.0, -114.5, n)
    zs = c1['zs']
    hts = rng.uniform(-95.0, 3000.0, n)
    hts[:8] = [zs[3], zs[4] - 0.4, zs[4] + 0.3, -99.5, zs[0] + 0.5, zs[7] - 1.0, zs[7] - 0.999, 36000.0]     # nodes, the 1 m rule, above the top
    inc = rng.uniform(15.0, 55.0, n)
    los = O.look_vectors_from_inc_hd(inc, rng.uniform(-180, 180, n), lat, lon, hts)
    zref = 30000.0
    w, h, npp = _oracle_vs_gpu(R, c1, lat, lon, hts, los, zref)
    assert w[7] == 0.0 and h[7] == 0.0 and np.isfinite(h).all()
    # the partition is NOT the lowest slice's: the maxima of the upper levels come from the steepest rays wherever they start, but the
    # lowest levels only see the rays that start below them
    cube = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet'], c1['hydro'], order='zyx')
    lowest = cube.raytrace(R.Rays.points(lat=lat, lon=lon, los=los), float(hts.min()), zref)[2]
    assert npp.shape == lowest.shape and np.array_equal(npp[-5:], lowest[-5:])
    # other segment length / integration top, a scene partly outside the cube (NaN mask), heights on a grid with look vectors
    _oracle_vs_gpu(R, c1, lat[:500], lon[:500], hts[:500], los[:500], 12000.0, max_seg=400.0)
    lat2 = rng.uniform(29.8, 36.2, 600); lon2 = rng.uniform(-121.3, -112.8, 600); h2 = rng.uniform(0, 2500, 600)
    los2 = O.look_vectors_from_inc_hd(rng.uniform(20, 50, 600), np.full(600, -167.9), lat2, lon2, h2)
    w2, _, _ = _oracle_vs_gpu(R, c1, lat2, lon2, h2, los2, zref)
    assert 0.05 < np.isnan(w2).mean() < 0.9


def test_dem_scene_on_the_bench_cube(R):
    """The c3b recipe at test size: per-pixel heights rng(2).uniform(0, 3000) on the scene grid, per-pixel look vectors, the
    300x300x80 cube - through GRID origins (the 16x16-pixel tiles of the light kernels) against the oracle."""
    from raider_amd.synthetic import synthetic_cube, scene_grid
    c = synthetic_cube(300, 300, 80, seed=0)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    zref = float(c['zs'].max() - 1)
    rows, cols = 150, 170
    xp, yp, inc_cols, hd = scene_grid(rows, cols)
    xx, yy = np.meshgrid(xp, yp)
    hts = np.random.default_rng(2).uniform(0.0, 3000.0, (rows, cols))
    los = O.look_vectors_from_inc_hd(np.broadcast_to(inc_cols, (rows, cols)), np.full((rows, cols), hd), yy, xx, hts)
    ow, oh, onp = OC.build_cube_ray_per_pixel(c, yy, xx, hts, los, zref)
    rays = R.Rays.grid(xp, yp, los=los, hts=hts)
    w, h, npp, fl = cube.raytrace(rays, None, zref)
    assert np.array_equal(npp, onp[_table_zz(R, cube, rays.ht_min, zref)])
    np.testing.assert_allclose(w, ow, rtol=0, atol=TIGHT); np.testing.assert_allclose(h, oh, rtol=0, atol=TIGHT)
    assert R.Context.default().generic_ray_count() == 0
    # a ray sees less atmosphere from higher up: against the slice at the mean height the per-pixel delays correlate with -height
    hs = cube.raytrace(R.Rays.grid(xp, yp, los=los), 1500.0, zref)[1]
    assert np.corrcoef((h - hs).ravel(), hts.ravel())[0, 1] < -0.99
    # workspace smaller than the batch: chunked pass 2, same bits
    ctx = R.Context.default()
    ctx.set_workspace_limit(3 << 20)
    try:
        w2, h2, np2, _ = cube.raytrace(rays, None, zref)
    finally:
        ctx.set_workspace_limit(48 << 30)
    assert np.array_equal(w2, w) and np.array_equal(h2, h) and np.array_equal(np2, npp)


def test_per_ray_height_errors(R, c1):
    cube = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet'], c1['hydro'], order='zyx')
    zref = float(c1['zs'].max() - 1)
    xp, yp, xx, yy, inc = _scene(6, 7)
    hts = np.random.default_rng(0).uniform(0, 1000, yy.shape)
    rays = R.Rays.grid(xp, yp, inc=35.0, hd=-167.9, hts=hts)
    with pytest.raises(ValueError, match='above the lowest'):
        cube.raytrace(rays, float(hts.min()) + 1.0, zref)
    with pytest.raises(ValueError, match='slice height'):
        cube.raytrace(R.Rays.grid(xp, yp, inc=35.0, hd=-167.9), None, zref)
    with pytest.raises(ValueError, match='shape'):
        R.Rays.grid(xp, yp, inc=35.0, hd=-167.9, hts=hts[:-1])
    with pytest.raises(ValueError, match='ONE slice'):
        cube.raytrace_slices(rays, [0.0, 100.0], zref)
    # an explicit table height BELOW the lowest ray is allowed: same rays, a taller (or equal) level table, same delays
    a = cube.raytrace(rays, None, zref)
    b = cube.raytrace(rays, float(hts.min()) - 50.0, zref)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    # the C contract (ht <= min(hts)) violated behind the wrapper's back: refused by name, NaN on the asynchronous path
    rays.ht_min = 900.0
    with pytest.raises(ValueError, match='per-ray heights'):
        cube.raytrace(rays, None, zref)
    import torch
    dev = torch.device('cuda:0')
    rd = R.Rays.grid(torch.from_numpy(xp).to(dev), torch.from_numpy(yp).to(dev), inc=35.0, hd=-167.9, hts=torch.from_numpy(hts).to(dev))
    rd.ht_min = 900.0
    w, h, _, _ = cube.raytrace(rd, None, zref, want_nparts=False)
    torch.cuda.synchronize()
    assert torch.isnan(w).all() and torch.isnan(h).all()
