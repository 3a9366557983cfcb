"""GPU: BASELINE-sized workloads through size-independent properties + spot checks against the oracle, and the edge
cases of the ray tracer (ragged tiles, tiny / non-uniform / descending axes, f64 cubes, origin & LOS modes, empty
batches, NaN rays, workspace chunking)."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import raider_oracle as O

TIGHT = 1e-9


@pytest.fixture(scope='module')
def R():
    import raider_amd
    return raider_amd


@pytest.fixture(scope='module')
def era5():
    return O.synthetic_cube(300, 300, 80, seed=0)


def _scene(rows, cols):
    xpts = np.linspace(-119.5, -115.5, cols)
    ypts = np.linspace(34.5, 31.5, rows)
    inc_cols = 30.0 + 16.0 * (np.arange(cols) / float(cols))
    return xpts, ypts, inc_cols


def _oracle_block(cube, xp, yp, inc_block, hd, zref, nparts, ht=0.0):
    ip = list(O.getInterpolators(cube['xs'], cube['ys'], cube['zs'], cube['wet'], cube['hydro']))
    look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(inc_block, np.full(yy.shape, hd), llh[1], llh[0], llh[2])
    w, h = O.build_cube_ray(xp, yp, np.array([ht]), look, ip, MAX_TROPO_HEIGHT=zref, nParts_override=[nparts])
    return w[0], h[0]


def _c_oracle_blocks(cube, xpts, ypts, inc_cols, hd, zref, nparts, wn, hn, origins, size):
    """size x size blocks of the scene against the multi-core C restatement (oracle/oracle_c.c; itself pinned on the goldens and
    on the NumPy oracle in tests/test_oracle_c.py) driven with the whole-slice partition: the look vectors are the oracle's own."""
    from oracle import oracle_c as OC
    worst = 0.0
    for r0, c0 in origins:
        xp, yp = xpts[c0:c0 + size], ypts[r0:r0 + size]
        xx, yy = np.meshgrid(xp, yp)
        los = O.look_vectors_from_inc_hd(np.broadcast_to(inc_cols[c0:c0 + size], yy.shape), np.full(yy.shape, hd), yy, xx, 0.0)
        ow, oh, _ = OC.build_cube_ray_slice(cube, xp, yp, 0.0, los, zref, nparts=nparts)
        gw, gh = wn[r0:r0 + size, c0:c0 + size], hn[r0:r0 + size, c0:c0 + size]
        gw, gh = (g.cpu().numpy() if hasattr(g, 'cpu') else g for g in (gw, gh))
        assert np.isfinite(ow).all() and np.isfinite(oh).all()
        np.testing.assert_allclose(gw, ow, rtol=0, atol=TIGHT)
        np.testing.assert_allclose(gh, oh, rtol=0, atol=TIGHT)
        worst = max(worst, float(np.abs(gw - ow).max()), float(np.abs(gh - oh).max()))
    return worst


def test_config3_full_size_properties(R, era5):
    """configs[2]: 4000x4000 rays through the 300x300x80 cube.  (a) random blocks against the oracle driven with the
    whole-slice partition; (b) chunked workspace == single-chunk workspace bit for bit; (c) exact linearity in the cube
    (doubling f32 refractivities doubles every delay bit for bit); (d) two half scenes driven with the all-reduced
    partition == the whole scene bit for bit."""
    import torch
    dev = torch.device('cuda')
    ctx = R.Context.default()
    rows = cols = 4000
    xpts, ypts, inc_cols = _scene(rows, cols)
    hd = -167.9
    zref = float(era5['zs'].max() - 1)
    cube = R.Cube(era5['ys'], era5['xs'], era5['zs'], era5['wet'], era5['hydro'], order='zyx')
    xt, yt = torch.from_numpy(xpts).to(dev), torch.from_numpy(ypts).to(dev)
    inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
    hdt = torch.full((rows, cols), hd, dtype=torch.float64, device=dev)
    rays = R.Rays.grid(xt, yt, inc=inc, hd=hdt)
    wet, hyd, nparts, flags = cube.raytrace(rays, 0.0, zref)
    torch.cuda.synchronize()
    assert int(nparts.sum()) == 178 and len(nparts) == 76
    wn, hn = wet.cpu().numpy(), hyd.cpu().numpy()
    assert np.isfinite(wn).all() and np.isfinite(hn).all()
    # (a) four 20x20 blocks (corners + interior)
    for r0, c0 in ((0, 0), (rows - 20, cols - 20), (1234, 2777), (3000, 16)):
        ow, oh = _oracle_block(era5, xpts[c0:c0 + 20], ypts[r0:r0 + 20], np.broadcast_to(inc_cols[c0:c0 + 20], (20, 20)), hd, zref, nparts)
        np.testing.assert_allclose(wn[r0:r0 + 20, c0:c0 + 20], ow, rtol=0, atol=TIGHT)
        np.testing.assert_allclose(hn[r0:r0 + 20, c0:c0 + 20], oh, rtol=0, atol=TIGHT)
    # (a') nine 256 x 256 blocks (590 k rays: corners, edges, interior, ragged offsets) against the C oracle
    b = 256
    worst = _c_oracle_blocks(era5, xpts, ypts, inc_cols, hd, zref, nparts, wn, hn,
                             ((0, 0), (0, cols - b), (rows - b, 0), (rows - b, cols - b), (1234, 2777), (3000, 16), (1871, 0), (7, 3700), (2001, 1999)), b)
    assert worst < 1e-9
    # (b) chunked integration (1 GiB workspace -> 11 chunks)
    ctx.set_workspace_limit(1 << 30)
    try:
        w2, h2, np2, _ = cube.raytrace(rays, 0.0, zref)
        assert np.array_equal(np2, nparts)
        assert torch.equal(w2, wet) and torch.equal(h2, hyd)
    finally:
        ctx.set_workspace_limit(48 << 30)
    # (c) linearity
    cube2 = R.Cube(era5['ys'], era5['xs'], era5['zs'], 2 * era5['wet'], 2 * era5['hydro'], order='zyx')
    w3, h3, _, _ = cube2.raytrace(rays, 0.0, zref)
    assert torch.equal(w3, 2 * wet) and torch.equal(h3, 2 * hyd)
    # (d) halves with the all-reduced partition (what raider_amd.distributed does across ranks)
    tops, parts = [], []
    for sl in (slice(0, 2000), slice(2000, 4000)):
        rr = R.Rays.grid(xt, yt[sl].contiguous(), inc=inc[sl].contiguous(), hd=hdt[sl].contiguous())
        ml, fl = cube.ray_prepass(rr, 0.0, zref)
        tops.append((rr, ml, fl))
    gmax = np.maximum(tops[0][1], tops[1][1]); gfl = tops[0][2] | tops[1][2]
    gnp = R.nparts_from_maxlen(gmax)
    assert np.array_equal(gnp, nparts)
    for (rr, _, _), sl in zip(tops, (slice(0, 2000), slice(2000, 4000))):
        wh, hh = cube.ray_march(rr, 0.0, zref, gnp, gfl)
        assert torch.equal(wh, wet[sl]) and torch.equal(hh, hyd[sl])


def test_zenith_constant_refractivity_is_path_length(R, era5):
    """N == 1, zenith look vectors: delay*1e6 == zref - ht (the ray runs along the ellipsoid normal)."""
    ones = np.ones_like(era5['wet'])
    cube = R.Cube(era5['ys'], era5['xs'], era5['zs'], ones, ones, order='zyx')
    xpts, ypts, _ = _scene(1000, 1000)
    zref = 30000.0
    for ht in (0.0, 1234.5):
        wet, hyd, _, _ = cube.raytrace(R.Rays.grid(xpts, ypts, zenith=True), ht, zref)
        np.testing.assert_allclose(wet * 1e6, zref - ht, rtol=0, atol=2e-4)
        assert np.array_equal(wet, hyd)


def test_config2_conventional_1000(R, era5):
    """configs[1]: 1000x1000 ZTD gather on the f64 totals cube, then /cos(inc); sample rows against the oracle."""
    tot = R.Cube(era5['ys'], era5['xs'], era5['zs'], era5['wet_total'], era5['hydro_total'], order='zyx')
    xpts, ypts, _ = _scene(1000, 1000)
    zpts = np.array([0.0, 150.0, 2999.0])
    wet, hyd = tot.build_cube(xpts, ypts, zpts)
    it = list(O.getInterpolators(era5['xs'], era5['ys'], era5['zs'], era5['wet_total'], era5['hydro_total']))
    for r in (0, 499, 999):
        ow, oh = O.build_cube(xpts, ypts[r:r + 1], zpts, it)
        np.testing.assert_allclose(wet[:, r:r + 1], ow, rtol=0, atol=1e-14)
        np.testing.assert_allclose(hyd[:, r:r + 1], oh, rtol=0, atol=1e-14)
    from raider_amd.losreader import Conventional
    conv = Conventional(inc=np.full((1000, 1000), 39.0), heading=np.full((1000, 1000), -167.9))
    conv.setPoints(np.zeros((1000, 1000)), np.zeros((1000, 1000)), np.zeros((1000, 1000)))
    np.testing.assert_allclose(conv(hyd[0]), hyd[0] / np.cos(np.radians(39.0)), rtol=1e-15)


def test_config4_full_size_one_gpu(R, era5):
    """configs[3]: the 10000x10000 scene (1e8 rays) BASELINE shards over 8 GPUs, here on ONE (23 GB of ray records fit the
    default workspace): finite everywhere, the reference's nParts, four blocks against the oracle driven with the whole-slice
    partition, and an 8 GiB workspace (the slice integrated in 3 chunks) == the unchunked result bit for bit."""
    import torch
    dev = torch.device('cuda')
    ctx = R.Context.default()
    rows = cols = 10000
    xpts, ypts, inc_cols = _scene(rows, cols)
    hd = -167.9
    zref = float(era5['zs'].max() - 1)
    cube = R.Cube(era5['ys'], era5['xs'], era5['zs'], era5['wet'], era5['hydro'], order='zyx')
    xt, yt = torch.from_numpy(xpts).to(dev), torch.from_numpy(ypts).to(dev)
    inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
    rays = R.Rays.grid(xt, yt, inc=inc, hd=hd)
    wet = torch.empty((rows, cols), dtype=torch.float64, device=dev); hyd = torch.empty_like(wet)
    _, _, nparts, flags = cube.raytrace(rays, 0.0, zref, out=(wet, hyd))
    assert int(nparts.sum()) == 178 and len(nparts) == 76
    assert bool(torch.isfinite(wet).all()) and bool(torch.isfinite(hyd).all())
    for r0, c0 in ((0, 0), (rows - 20, cols - 20), (6321, 2777), (9000, 16)):
        ow, oh = _oracle_block(era5, xpts[c0:c0 + 20], ypts[r0:r0 + 20], np.broadcast_to(inc_cols[c0:c0 + 20], (20, 20)), hd, zref, nparts)
        np.testing.assert_allclose(wet[r0:r0 + 20, c0:c0 + 20].cpu().numpy(), ow, rtol=0, atol=TIGHT)
        np.testing.assert_allclose(hyd[r0:r0 + 20, c0:c0 + 20].cpu().numpy(), oh, rtol=0, atol=TIGHT)
    b = 256                                                                # + nine 256 x 256 blocks against the C oracle
    _c_oracle_blocks(era5, xpts, ypts, inc_cols, hd, zref, nparts, wet, hyd,
                     ((0, 0), (0, cols - b), (rows - b, 0), (rows - b, cols - b), (6321, 2777), (9000, 16), (4999, 5001), (13, 9700), (7777, 0)), b)
    w2 = torch.empty_like(wet); h2 = torch.empty_like(wet)
    ctx.set_workspace_limit(8 << 30)
    try:
        _, _, np2, _ = cube.raytrace(rays, 0.0, zref, out=(w2, h2))
    finally:
        ctx.set_workspace_limit(48 << 30)
    assert np.array_equal(np2, nparts) and torch.equal(w2, wet) and torch.equal(h2, hyd)


HRRR = dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5 - 360.0, a=6371229.0, es=0.0)       # models/hrrr.py:248-259


def test_config5_lcc_blend_stations_and_rays(R):
    """configs[4] as SURVEY 8(d) specifies it: HRRR-like 1000x1000x50 f32 cubes on the 3-km Lambert-conformal-conic grid
    (x = -1.5e6 + 3000 i, y = -1.5e6 + 3000 j, 50 quadratic levels to 26 km), two epochs blended (0.25, 0.75), 5 M station
    points given in lon/lat (projected on the device), and a ray-traced 2000x2000 scene through the same blended cube.
    gather(blend) == the oracle RGI on the f32-blended cube at the oracle's own LCC coordinates; ray blocks == the oracle with
    the cube's projection (delay.py:253,295) driven with the whole-slice partition."""
    rng = np.random.default_rng(3)
    xs = -1.5e6 + 3000.0 * np.arange(1000); ys = -1.5e6 + 3000.0 * np.arange(1000)
    zs = np.round(-100 + 26100 * np.linspace(0, 1, 50) ** 2, 3)
    hyd0 = 270 * np.exp(-zs / 8000)[:, None, None]; wet0 = 60 * np.exp(-zs / 2000)[:, None, None]
    e = [(hyd0 * (1 + 0.01 * rng.standard_normal((1, 1000, 1000)))).astype(np.float32) if k % 2 else
         (wet0 * (1 + 0.1 * rng.standard_normal((1, 1000, 1000)))).astype(np.float32) for k in range(4)]      # wet_a, hydro_a, wet_b, hydro_b
    a = R.Cube(ys, xs, zs, e[0], e[1], order='zyx').set_projection_lcc(**HRRR)
    b = R.Cube(ys, xs, zs, e[2], e[3], order='zyx').set_projection_lcc(**HRRR)
    w1, w2 = 0.25, 0.75
    m = a.blend(w1, b, w2)
    assert m.projection == a.projection
    bw = O.blend_cubes(w1, e[0], w2, e[2]); bh = O.blend_cubes(w1, e[1], w2, e[3])
    # ---- 5 M stations in lon/lat, inside the grid
    n = 5_000_000
    lat = rng.uniform(28.0, 49.0, n); lon = rng.uniform(-110.0, -85.0, n); hgt = rng.uniform(0, 4000, n)
    py, px = m.project(lat, lon)
    assert px.min() > xs[0] and px.max() < xs[-1] and py.min() > ys[0] and py.max() < ys[-1]
    gw, gh = m.interp(np.stack([py, px, hgt], -1))
    assert np.isfinite(gw).all() and np.isfinite(gh).all()
    idx = rng.choice(n, 20000, replace=False)
    ox, oy = O.lcc_forward(lat[idx], lon[idx], **HRRR)
    np.testing.assert_allclose(px[idx], ox, rtol=0, atol=2e-6); np.testing.assert_allclose(py[idx], oy, rtol=0, atol=2e-6)
    iw, ih = O.getInterpolators(xs, ys, zs, bw, bh)
    q = np.stack([oy, ox, hgt[idx]], -1)
    np.testing.assert_allclose(gw[idx], iw(q), rtol=0, atol=1e-8)          # (2e-6 m of projection difference x the field's gradient)
    np.testing.assert_allclose(gh[idx], ih(q), rtol=0, atol=1e-8)
    # ---- 2000 x 2000 rays through the blended LCC cube
    rows = cols = 2000
    lonp = np.linspace(-104.0, -92.0, cols); latp = np.linspace(44.0, 33.0, rows)
    inc_cols = 30.0 + 16.0 * (np.arange(cols) / float(cols)); hd = -167.9
    zref = float(zs.max() - 1)
    inc = np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))
    wet, hyd, nparts, _ = m.raytrace(R.Rays.grid(lonp, latp, inc=inc, hd=hd), 0.0, zref)
    assert np.isfinite(wet).all() and np.isfinite(hyd).all()
    ip = list(O.getInterpolators(xs, ys, zs, bw, bh))
    for r0, c0 in ((0, 0), (rows - 12, cols - 12), (777, 1234)):
        incb = np.broadcast_to(inc_cols[c0:c0 + 12], (12, 12))
        look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(incb, np.full(yy.shape, hd), llh[1], llh[0], llh[2])
        ow, oh = O.build_cube_ray(lonp[c0:c0 + 12], latp[r0:r0 + 12], np.array([0.0]), look, ip, MAX_TROPO_HEIGHT=zref,
                                  nParts_override=[nparts], model_proj=HRRR)
        np.testing.assert_allclose(wet[r0:r0 + 12, c0:c0 + 12], ow[0], rtol=0, atol=TIGHT)
        np.testing.assert_allclose(hyd[r0:r0 + 12, c0:c0 + 12], oh[0], rtol=0, atol=TIGHT)
    # ... and five 200 x 200 blocks (200 k rays) against the C oracle with the same projection
    from oracle import oracle_c as OC
    blended = dict(ys=ys, xs=xs, zs=zs, wet=bw, hydro=bh)
    b = 200
    for r0, c0 in ((0, 0), (0, cols - b), (rows - b, 0), (rows - b, cols - b), (901, 777)):
        xx, yy = np.meshgrid(lonp[c0:c0 + b], latp[r0:r0 + b])
        los = O.look_vectors_from_inc_hd(np.broadcast_to(inc_cols[c0:c0 + b], yy.shape), np.full(yy.shape, hd), yy, xx, 0.0)
        ow, oh, _ = OC.build_cube_ray_slice(blended, lonp[c0:c0 + b], latp[r0:r0 + b], 0.0, los, zref, nparts=nparts, model_proj=HRRR)
        assert np.isfinite(ow).all()
        np.testing.assert_allclose(wet[r0:r0 + b, c0:c0 + b], ow, rtol=0, atol=TIGHT)
        np.testing.assert_allclose(hyd[r0:r0 + b, c0:c0 + b], oh, rtol=0, atol=TIGHT)


# ---- edge cases -----------------------------------------------------------------------------------------------
def _small_cube(ny=9, nx=11, nz=12, seed=2, **kw):
    return O.synthetic_cube(ny, nx, nz, seed=seed, y0=31.0, y1=35.0, x0=-120.0, x1=-115.0, **kw)


def _check_against_oracle(R, c, xpts, ypts, inc, hd, ht, zref, cube=None, max_seg=1000.0):
    cube = cube or R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro']))
    incb = np.broadcast_to(np.asarray(inc, float), (len(ypts), len(xpts)))
    look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(incb, np.full(yy.shape, hd), llh[1], llh[0], llh[2])
    (ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([ht]), look, ip, MAX_SEGMENT_LENGTH=max_seg, MAX_TROPO_HEIGHT=zref, return_nparts=True)
    wet, hyd, nparts, _ = cube.raytrace(R.Rays.grid(xpts, ypts, inc=incb.copy(), hd=hd), ht, zref, max_seg=max_seg)
    assert np.array_equal(nparts, onp[0])
    np.testing.assert_allclose(wet, ow[0], rtol=0, atol=TIGHT, equal_nan=True)
    np.testing.assert_allclose(hyd, oh[0], rtol=0, atol=TIGHT, equal_nan=True)
    return wet, hyd


@pytest.mark.parametrize('shape', [(1, 1), (1, 37), (17, 1), (16, 16), (33, 47)])
def test_ragged_scene_shapes(R, shape):
    c = _small_cube()
    ny, nx = shape
    _check_against_oracle(R, c, np.linspace(-118.5, -116.5, nx), np.linspace(33.9, 32.1, ny), 35.0, -167.9, 10.0, c['zs'].max() - 1)


def test_tiny_and_nonuniform_axes(R):
    """3-node axes (window search falls back to bisection), non-uniform x/y axes (bisection path), descending y."""
    c = _small_cube(ny=3, nx=3, nz=3)
    _check_against_oracle(R, c, np.linspace(-118.5, -116.5, 5), np.linspace(33.9, 32.1, 4), 30.0, -12.1, 0.0, c['zs'].max() - 1)
    c = _small_cube()
    rng = np.random.default_rng(0)
    c['xs'] = np.sort(-120 + 5 * rng.uniform(0, 1, 11)); c['xs'][0] = -120; c['xs'][-1] = -115
    c['ys'] = np.sort(31 + 4 * rng.uniform(0, 1, 9)); c['ys'][0] = 31; c['ys'][-1] = 35
    _check_against_oracle(R, c, np.linspace(-118.5, -116.5, 9), np.linspace(33.9, 32.1, 8), 40.0, -167.9, 0.0, c['zs'].max() - 1)
    # descending y axis in the file (scipy flips; so does the device pack kernel)
    d = dict(c); d['ys'] = c['ys'][::-1].copy(); d['wet'] = c['wet'][:, ::-1].copy(); d['hydro'] = c['hydro'][:, ::-1].copy()
    w1, h1 = _check_against_oracle(R, d, np.linspace(-118.5, -116.5, 9), np.linspace(33.9, 32.1, 8), 40.0, -167.9, 0.0, c['zs'].max() - 1)
    w0, h0 = _check_against_oracle(R, c, np.linspace(-118.5, -116.5, 9), np.linspace(33.9, 32.1, 8), 40.0, -167.9, 0.0, c['zs'].max() - 1)
    assert np.array_equal(w0, w1) and np.array_equal(h0, h1)


def test_long_nonuniform_axes_use_large_lds_tables(R):
    """A wide cube whose lat / lon axes are only NEARLY uniform (f32-rounded 0.1-degree nodes, as global analyses are distributed):
    their (node, 1/spacing) tables are 16 B per node of LDS, here 83 KB - past the 64 KB default allocation, inside the 160 KB a
    gfx950 workgroup may have.  Still exact against the oracle; axes that cannot fit at all are refused with a clear message."""
    n = 2600
    rng = np.random.default_rng(5)
    ys = np.linspace(-60.0, 60.0, n).astype(np.float32).astype(np.float64)
    xs = np.linspace(-130.0, 130.0, n).astype(np.float32).astype(np.float64)
    zs = np.array([-100.0, 900.0, 2500.0, 5500.0, 11000.0, 21000.0])
    prof_h = 270.0 * np.exp(-zs / 8000.0); prof_w = 60.0 * np.exp(-zs / 2000.0)
    g = (1.0 + 0.02 * rng.standard_normal((n, n))).astype(np.float32)
    c = dict(ys=ys, xs=xs, zs=zs, wet=(prof_w[:, None, None] * g[None]).astype(np.float32), hydro=(prof_h[:, None, None] * g[None]).astype(np.float32))
    assert np.ptp(np.diff(ys)) > 1e-7                                     # not exactly uniform: the table path
    _check_against_oracle(R, c, np.linspace(-118.5, -116.5, 9), np.linspace(33.9, 32.1, 8), 38.0, -167.9, 0.0, zs.max() - 1)
    del c, g
    # 500 model levels (the library's limit is 512): the per-level tables alone are 96 KB
    c = O.synthetic_cube(12, 14, 500, seed=3, y0=31.0, y1=35.0, x0=-120.0, x1=-115.0)
    c['zs'] = np.linspace(-100.0, 40000.0, 500)
    _check_against_oracle(R, c, np.linspace(-118.5, -116.5, 7), np.linspace(33.9, 32.1, 6), 33.0, -12.1, 20.0, c['zs'].max() - 1)
    n = 6000                                                              # 192 KB of tables: more LDS than a workgroup can have
    ys = np.linspace(-60.0, 60.0, n).astype(np.float32).astype(np.float64)
    xs = np.linspace(-130.0, 130.0, n).astype(np.float32).astype(np.float64)
    import torch
    v = torch.ones((2, n, n), dtype=torch.float32, device='cuda:0')
    big = R.Cube(ys, xs, np.array([0.0, 30000.0]), v, v, order='zyx')
    with pytest.raises(Exception, match='LDS'):
        big.raytrace(R.Rays.grid(np.linspace(-118.5, -116.5, 9), np.linspace(33.9, 32.1, 8), inc=30.0, hd=-167.9), 0.0, 29000.0)


def test_cube_beyond_4_gib_uses_64_bit_offsets(R):
    """A 1700 x 1700 x 240 f32 cube is 5.5 GB of interleaved (wet, hydro) pairs - a global 0.1-degree analysis is of that order - so
    the 32-bit gather offsets of the usual instantiation do not reach its far end: the kernels must take their 64-bit form.  Built
    on the device; rays, zenith nodes and station points sit in the last rows (byte offsets > 2^32) and are checked against the
    oracle on the sub-cube around them."""
    import torch
    dev = torch.device('cuda:0')
    ny = nx = 1700; nz = 240
    ys = np.linspace(20.0, 54.0, ny); xs = np.linspace(-130.0, -96.0, nx); zs = np.linspace(-100.0, 41000.0, nz)
    iy = torch.arange(ny, device=dev, dtype=torch.float32); ix = torch.arange(nx, device=dev, dtype=torch.float32)
    g = 1.0 + 0.05 * torch.sin(0.37 * iy)[:, None] * torch.cos(0.21 * ix)[None, :]
    zt = torch.from_numpy(zs).to(dev)
    pw = (60.0 * torch.exp(-zt / 2000.0)).float(); ph = (270.0 * torch.exp(-zt / 8000.0)).float()
    wet = (pw[:, None, None] * g[None]).contiguous(); hyd = (ph[:, None, None] * g[None]).contiguous()
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx')
    j0, j1, i0, i1 = 1560, 1700, 1500, 1700                                           # rows whose offsets exceed 4 GiB
    assert (j0 * nx * nz) * 8 > 2 ** 32
    sub = dict(ys=ys[j0:j1], xs=xs[i0:i1], zs=zs, wet=wet[:, j0:j1, i0:i1].cpu().numpy(), hydro=hyd[:, j0:j1, i0:i1].cpu().numpy())
    del wet, hyd, g
    ypts = np.linspace(ys[j0 + 60], ys[j0 + 30], 9); xpts = np.linspace(xs[i0 + 60], xs[i0 + 110], 11)
    _check_against_oracle(R, sub, xpts, ypts, 37.0, -167.9, 150.0, zs.max() - 1, cube=cube)
    ip = list(O.getInterpolators(sub['xs'], sub['ys'], sub['zs'], sub['wet'], sub['hydro']))
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(ys[j0 + 5], ys[-1], 3000), rng.uniform(xs[i0 + 5], xs[-1], 3000), rng.uniform(-50, 30000, 3000)], -1)
    pts[:20, 0] = ys[-1]; pts[20:40, 1] = xs[-1]                                       # the very last node of both axes
    w, h = cube.interp(pts)
    np.testing.assert_allclose(w, ip[0](pts), rtol=0, atol=1e-12); np.testing.assert_allclose(h, ip[1](pts), rtol=0, atol=1e-12)
    assert np.isfinite(w).all()
    zw, zh = cube.build_cube(xpts, ypts, np.array([0.0, 2500.0]))
    ow, oh = O.build_cube(xpts, ypts, np.array([0.0, 2500.0]), ip)
    np.testing.assert_allclose(zw, ow, rtol=0, atol=1e-12); np.testing.assert_allclose(zh, oh, rtol=0, atol=1e-12)


def test_f64_cube_and_other_maxseg(R):
    c = _small_cube(nz=30)
    c64 = dict(c); c64['wet'] = c['wet'].astype(np.float64) * 1.000000123; c64['hydro'] = c['hydro'].astype(np.float64) * 0.999999877
    _check_against_oracle(R, c64, np.linspace(-118.5, -116.5, 13), np.linspace(33.9, 32.1, 10), 25.0, 12.0, 300.0, 26000.0, max_seg=250.0)
    _check_against_oracle(R, c64, np.linspace(-118.5, -116.5, 13), np.linspace(33.9, 32.1, 10), 50.0, 170.0, -50.0, 9000.0, max_seg=3000.0)


def test_origin_and_los_modes_agree(R):
    """GRID / LLH / XYZ origins and vector / per-ray inc-heading / scalar inc-heading look vectors: same rays, same bits."""
    c = _small_cube(nz=20)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    xp = np.linspace(-118.5, -116.5, 19); yp = np.linspace(33.9, 32.1, 14)
    xx, yy = np.meshgrid(xp, yp)
    ht, zref = 120.0, c['zs'].max() - 1
    ref = cube.raytrace(R.Rays.grid(xp, yp, inc=37.0, hd=-167.9), ht, zref)
    los = R.Rays.grid(xp, yp, inc=37.0, hd=-167.9).look_vectors()
    np.testing.assert_allclose(los, O.look_vectors_from_inc_hd(np.full(yy.shape, 37.0), np.full(yy.shape, -167.9), yy, xx, ht), rtol=0, atol=1e-15)
    xyz = np.stack(O.lla2ecef(yy, xx, np.full(yy.shape, ht)), -1)
    variants = [
        R.Rays.grid(xp, yp, inc=np.full(yy.shape, 37.0), hd=np.full(yy.shape, -167.9)),
        R.Rays.grid(xp, yp, los=los),
        R.Rays.points(lat=yy.copy(), lon=xx.copy(), inc=37.0, hd=-167.9),
        R.Rays.points(lat=yy.copy(), lon=xx.copy(), los=los),
    ]
    for rays in variants:
        w, h, npx, _ = cube.raytrace(rays, ht, zref)
        assert np.array_equal(npx, ref[2])
        np.testing.assert_allclose(w, ref[0], rtol=0, atol=1e-12)
        np.testing.assert_allclose(h, ref[1], rtol=0, atol=1e-12)
    # ECEF origins (1 ulp away from the device's own lla2ecef) with explicit look vectors
    w, h, npx, _ = cube.raytrace(R.Rays.points(xyz=xyz, los=los), ht, zref)
    np.testing.assert_allclose(w, ref[0], rtol=0, atol=1e-11)
    np.testing.assert_allclose(h, ref[1], rtol=0, atol=1e-11)


def test_empty_and_nan_batches(R):
    c = _small_cube()
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    zref = c['zs'].max() - 1
    w, h, _, _ = cube.raytrace(R.Rays.points(lat=np.zeros(0), lon=np.zeros(0), inc=30.0, hd=0.0), 0.0, zref)
    assert w.shape == (0,) and h.shape == (0,)
    gw, gh = cube.interp(np.zeros((0, 3)))
    assert gw.shape == (0,)
    xp = np.linspace(-118.5, -116.5, 8); yp = np.linspace(33.9, 32.1, 6)
    los = R.Rays.grid(xp, yp, inc=30.0, hd=0.0).look_vectors()
    los[2, 3] = np.nan                                     # ONE failed geo2rdr pixel: the reference's nParts is undefined
    with pytest.raises(ValueError):
        cube.raytrace(R.Rays.grid(xp, yp, los=los), 0.0, zref)
    # ...but the explicit-partition path still integrates every other ray and leaves NaN in that pixel
    good = R.Rays.grid(xp, yp, inc=30.0, hd=0.0)
    ml, fl = cube.ray_prepass(good, 0.0, zref)
    w, h = cube.ray_march(R.Rays.grid(xp, yp, los=los), 0.0, zref, R.nparts_from_maxlen(ml), fl)
    assert np.isnan(w[2, 3]) and np.isnan(h[2, 3]) and np.isfinite(np.delete(w.ravel(), 2 * 8 + 3)).all()
    ref = cube.raytrace(good, 0.0, zref)
    mask = np.ones(w.shape, bool); mask[2, 3] = False
    np.testing.assert_allclose(w[mask], ref[0][mask], rtol=0, atol=1e-12)


def test_bottom_clamp_quirk(R):
    """ht == min(model_zs): the first sample sits on the cube floor +- 1e-9 m; when EVERY pixel's sample is below it the
    reference clamps them to zmin (delay.py:306-307), otherwise the pixels below get NaN.  Which one happens is decided
    by round-off in the reference too; here: the kernel's own flag decides, and results are finite or NaN accordingly."""
    c = _small_cube()
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    xp = np.linspace(-118.5, -116.5, 8); yp = np.linspace(33.9, 32.1, 6)
    rays = R.Rays.grid(xp, yp, inc=30.0, hd=0.0)
    zmin = float(c['zs'].min())
    ml, fl = cube.ray_prepass(rays, zmin, c['zs'].max() - 1)
    w, h = cube.ray_march(rays, zmin, c['zs'].max() - 1, R.nparts_from_maxlen(ml), fl & ~4)      # force "all below" -> clamp
    assert np.isfinite(w).all()
    ref = cube.raytrace(R.Rays.grid(xp, yp, inc=30.0, hd=0.0), zmin + 1e-3, c['zs'].max() - 1)
    np.testing.assert_allclose(h, ref[1], rtol=0, atol=1e-6)


def test_hrrr_lambert_cube(R):
    """Projected model CRS (HRRR spherical LCC, models/hrrr.py:248-259): lon/lat query nodes and ray samples are
    projected to the model's x/y metres on the device (pyproj's step in delay.py:207-209,253,295).  Checked against the
    oracle's restatement of the PROJ formulas (parity with PROJ itself is unpinned) and against closed-form properties."""
    from raider_amd.delay import _build_cube, _build_cube_ray
    from raider_amd.delayFcns import interpolators_from_cube
    from raider_amd.losreader import Raytracing
    hrrr = dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0)
    # origin of the projection maps to (0,0); scale is true on the standard parallel
    x0, y0 = O.lcc_forward(38.5, 262.5, **hrrr)
    assert abs(x0) < 1e-6 and abs(y0) < 1e-6
    xe, _ = O.lcc_forward(38.5, 262.5 + 1e-4, **hrrr)
    assert abs(xe - 6371229.0 * np.cos(np.radians(38.5)) * np.radians(1e-4)) < 1e-6
    # 3-km HRRR-like grid
    ny, nx, nz = 120, 140, 30
    ys = -180e3 + 3000.0 * np.arange(ny); xs = -900e3 + 3000.0 * np.arange(nx)
    zs = np.round(-100 + 26100 * np.linspace(0, 1, nz) ** 2, 3)
    rng = np.random.default_rng(9)
    z3 = zs[:, None, None]
    wet = (60 * np.exp(-z3 / 2000) * (1 + 0.1 * rng.standard_normal((ny, nx))[None])).astype(np.float32)
    hyd = (270 * np.exp(-z3 / 8000) * (1 + 0.01 * rng.standard_normal((ny, nx))[None])).astype(np.float32)
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx')
    cube.set_projection_lcc(**hrrr)
    lat = np.linspace(38.4, 37.6, 21); lon = np.linspace(-106.5, -104.9, 25)      # negative longitudes vs lon_0 = 262.5
    py, px = cube.project(*np.meshgrid(lat, lon, indexing='ij'))
    ox, oy = O.lcc_forward(*np.meshgrid(lat, lon, indexing='ij'), **hrrr)
    np.testing.assert_allclose(px, ox, rtol=0, atol=1e-6); np.testing.assert_allclose(py, oy, rtol=0, atol=1e-6)
    assert xs[0] < px.min() and px.max() < xs[-1] and ys[0] < py.min() and py.max() < ys[-1]
    ifw, ifh = interpolators_from_cube(cube)
    ip = list(O.getInterpolators(xs, ys, zs, wet, hyd))
    proj_str = '+proj=lcc +lat_1=38.5 +lat_2=38.5 +lat_0=38.5 +lon_0=262.5 +x_0=0 +y_0=0 +a=6371229 +b=6371229 +units=m +no_defs'
    zpts = np.array([0.0, 1500.0])
    gw, gh = _build_cube(lon, lat, zpts, proj_str, 4326, [ifw, ifh])
    ow, oh = O.build_cube(lon, lat, zpts, ip, model_proj=hrrr)
    np.testing.assert_allclose(gw, ow, rtol=0, atol=1e-9); np.testing.assert_allclose(gh, oh, rtol=0, atol=1e-9)
    look = lambda ht, llh, xyz, yy: O.look_vectors_from_inc_hd(np.full(yy.shape, 36.0), np.full(yy.shape, -167.9), llh[1], llh[0], llh[2])
    zref = float(zs.max() - 1)
    (rw, rh), onp = O.build_cube_ray(lon, lat, zpts, look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True, model_proj=hrrr)
    w, h = _build_cube_ray(lon, lat, zpts, Raytracing(inc=36.0, heading=-167.9), proj_str, 4326, [ifw, ifh], MAX_TROPO_HEIGHT=zref)
    assert np.isfinite(rw).all()
    np.testing.assert_allclose(w, rw, rtol=0, atol=TIGHT); np.testing.assert_allclose(h, rh, rtol=0, atol=TIGHT)


def test_ellipsoidal_lambert_cube(R):
    """An LCC cube on an ELLIPSOIDAL cone (two standard parallels, WGS84): the polynomial kernels only know the spherical cone
    (HRRR), so every ray is handed to the generic kernels, which evaluate the full PROJ formulas - same parity bar."""
    from raider_amd.delay import _build_cube_ray
    from raider_amd.delayFcns import interpolators_from_cube
    from raider_amd.losreader import Raytracing
    ell = dict(lat_1=33.0, lat_2=45.0, lat_0=38.5, lon_0=262.5, x_0=1000.0, y_0=-2000.0, a=6378137.0, es=0.0066943799901413165)
    lat = np.linspace(38.4, 37.6, 13); lon = np.linspace(-106.5, -104.9, 15)
    cx, cy = O.lcc_forward(*np.meshgrid(lat, lon, indexing='ij'), **ell)
    ny, nx, nz = 90, 110, 30
    ys = cy.min() - 60e3 + 3000.0 * np.arange(ny); xs = cx.min() - 60e3 + 3000.0 * np.arange(nx)
    assert ys[-1] > cy.max() + 30e3 and xs[-1] > cx.max() + 30e3
    zs = np.round(-100 + 26100 * np.linspace(0, 1, nz) ** 2, 3)
    rng = np.random.default_rng(10)
    z3 = zs[:, None, None]
    wet = (60 * np.exp(-z3 / 2000) * (1 + 0.1 * rng.standard_normal((ny, nx))[None])).astype(np.float32)
    hyd = (270 * np.exp(-z3 / 8000) * (1 + 0.01 * rng.standard_normal((ny, nx))[None])).astype(np.float32)
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx')
    cube.set_projection_lcc(**ell)
    py, px = cube.project(*np.meshgrid(lat, lon, indexing='ij'))
    np.testing.assert_allclose(px, cx, rtol=0, atol=1e-6); np.testing.assert_allclose(py, cy, rtol=0, atol=1e-6)
    ip = list(O.getInterpolators(xs, ys, zs, wet, hyd))
    look = lambda ht, llh, xyz, yy: O.look_vectors_from_inc_hd(np.full(yy.shape, 36.0), np.full(yy.shape, -167.9), llh[1], llh[0], llh[2])
    zref = float(zs.max() - 1)
    zpts = np.array([0.0, 1500.0])
    (rw, rh), _ = O.build_cube_ray(lon, lat, zpts, look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True, model_proj=ell)
    w, h = _build_cube_ray(lon, lat, zpts, Raytracing(inc=36.0, heading=-167.9), dict(proj='lcc', **ell), 4326, list(interpolators_from_cube(cube)),
                           MAX_TROPO_HEIGHT=zref)
    assert np.isfinite(rw).all()
    np.testing.assert_allclose(w, rw, rtol=0, atol=TIGHT); np.testing.assert_allclose(h, rh, rtol=0, atol=TIGHT)


def test_device_resident_partition_exchange(R):
    """rdr_ray_prepass_device / rdr_ray_march_device: two half-slabs driven through a device-resident partition that is
    MAX-combined on the device (what the RCCL all-reduce does across ranks) reproduce the whole-slice result bit for bit,
    and the partition holds exactly the host API's per-level maxima and flags."""
    import torch
    dev = torch.device('cuda:0')
    c = O.synthetic_cube(50, 50, 40, seed=0)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet']).to(dev), torch.from_numpy(c['hydro']).to(dev), order='zyx')
    xp = torch.from_numpy(np.linspace(-119.5, -115.5, 64)).to(dev); yp = np.linspace(34.5, 31.5, 64)
    inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(np.linspace(30, 46, 64)[:, None], (64, 64)))).to(dev)   # steeper towards the last rows
    hd = torch.full((64, 64), -167.9, dtype=torch.float64, device=dev)
    zref = float(c['zs'].max() - 1)
    whole = R.Rays.grid(xp, torch.from_numpy(yp).to(dev), inc=inc, hd=hd)
    ww, wh, nparts, flags = cube.raytrace(whole, 0.0, zref)
    K = len(nparts)
    maxlen, fl = cube.ray_prepass(whole, 0.0, zref)
    halves = [R.Rays.grid(xp, torch.from_numpy(yp[a:b].copy()).to(dev), inc=inc[a:b].contiguous(), hd=hd[a:b].contiguous()) for a, b in ((0, 32), (32, 64))]
    parts = [torch.zeros(K + 4, dtype=torch.float64, device=dev) for _ in halves]
    for r, p in zip(halves, parts):
        cube.ray_prepass_device(r, 0.0, zref, p)
    glob = torch.maximum(parts[0], parts[1])
    assert np.array_equal(glob[:K].cpu().numpy(), maxlen)
    assert [int(v) for v in glob[K:].cpu().numpy()] == [(fl >> b) & 1 for b in range(4)]
    outs = [cube.ray_march_device(r, 0.0, zref, glob) for r in halves]
    torch.cuda.synchronize()
    assert torch.equal(torch.cat([outs[0][0], outs[1][0]]), ww) and torch.equal(torch.cat([outs[0][1], outs[1][1]]), wh)
    # a shard-local partition is NOT the same thing
    local = cube.ray_march_device(halves[0], 0.0, zref, parts[0])
    torch.cuda.synchronize()
    assert not torch.equal(local[1], wh[:32])


def test_continental_grid_axes(R):
    """A CONUS-sized model grid (1059 x 1799 nodes - HRRR's lattice - x 12 levels): exactly-uniform axes need no LDS table in the
    ray kernels (69 KB per workgroup otherwise) and exceed what the zenith kernels stage in LDS (they read them from global
    memory then).  Ray tracing and the zenith gather against the oracle on a patch of it."""
    ny, nx, nz = 1059, 1799, 12
    ys = 21.0 + 0.03 * np.arange(ny); xs = -135.0 + 0.03 * np.arange(nx)
    zs = np.round(-100 + 30000 * np.linspace(0, 1, nz) ** 2, 3)
    rng = np.random.default_rng(2)
    base = np.exp(-zs / 7000.0)[:, None, None]
    bump = (1 + 0.02 * np.sin(ys / 3.0)[None, :, None] * np.cos(xs / 2.0)[None, None, :]).astype(np.float32)
    hyd = (270.0 * base * bump).astype(np.float32); wet = (50.0 * np.exp(-zs / 2500.0)[:, None, None] * bump).astype(np.float32)
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx')
    zref = float(zs.max() - 1)
    xp = np.linspace(-101.0, -100.2, 13); yp = np.linspace(40.4, 39.9, 11)
    w, h, nparts, _ = cube.raytrace(R.Rays.grid(xp, yp, inc=37.0, hd=-167.9), 300.0, zref)
    sl_y = slice(600, 660); sl_x = slice(1100, 1180)                   # the patch the rays stay in
    ip = list(O.getInterpolators(xs[sl_x], ys[sl_y], zs, wet[:, sl_y, sl_x], hyd[:, sl_y, sl_x]))
    look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(np.full(yy.shape, 37.0), np.full(yy.shape, -167.9), llh[1], llh[0], llh[2])
    (ow, oh), onp = O.build_cube_ray(xp, yp, np.array([300.0]), look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True)
    assert np.array_equal(nparts, onp[0]) and np.isfinite(ow).all()
    np.testing.assert_allclose(w, ow[0], rtol=0, atol=5e-9); np.testing.assert_allclose(h, oh[0], rtol=0, atol=5e-9)
    gw, gh = cube.build_cube(xp, yp, np.array([0.0, 2500.0]))
    cw, ch = O.build_cube(xp, yp, np.array([0.0, 2500.0]), ip)
    np.testing.assert_allclose(gw, cw, rtol=0, atol=1e-12); np.testing.assert_allclose(gh, ch, rtol=0, atol=1e-12)
    # points outside the grid are NaN on this path too
    wo, ho, _, _ = cube.raytrace(R.Rays.grid(np.array([-136.0, -120.0]), np.array([30.0]), inc=20.0, hd=0.0), 0.0, zref)
    assert np.isnan(wo[0, 0]) and np.isfinite(wo[0, 1])


def test_host_buffer_pipeline_matches_device_path(R):
    """>= 2 M rays with per-ray look vectors handed over as NumPy arrays take the pipelined path (look vectors up / outputs down
    in row chunks overlapped with the two passes): bit-identical to the device-resident path, ragged scene edges included,
    for GRID and for point-list origins."""
    import torch
    dev = torch.device('cuda:0')
    c = O.synthetic_cube(60, 60, 30, seed=1)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    ny, nx = 1453, 1501                                  # 2.18 M rays, neither dimension a multiple of the 16-pixel tile
    xp = np.linspace(-119.5, -115.5, nx); yp = np.linspace(34.5, 31.5, ny)
    inc = np.broadcast_to(np.linspace(25, 45, nx)[None, :], (ny, nx)).copy()
    los = R.Rays.grid(xp, yp, inc=inc, hd=np.full((ny, nx), -167.9)).look_vectors()          # NumPy (ny, nx, 3)
    los[7, 11] = np.nan                                                                        # one NaN ray
    zref = float(c['zs'].max() - 1)
    with pytest.raises(ValueError, match='NaN'):                                              # the NaN poisons the slice maximum (delay.py:283)
        cube.raytrace(R.Rays.grid(xp, yp, los=los), 120.0, zref)
    los[7, 11] = los[7, 12]
    hw, hh, hn, hf = cube.raytrace(R.Rays.grid(xp, yp, los=los), 120.0, zref)                  # host buffers -> pipelined
    dw, dh, dn, df = cube.raytrace(R.Rays.grid(torch.from_numpy(xp).to(dev), torch.from_numpy(yp).to(dev), los=torch.from_numpy(los).to(dev)), 120.0, zref)
    assert np.array_equal(hn, dn) and hf == df
    assert np.array_equal(hw, dw.cpu().numpy()) and np.array_equal(hh, dh.cpu().numpy())
    assert np.isfinite(hw).mean() > 0.9
    # point-list origins (LLH) through the same path
    xx, yy = np.meshgrid(xp, yp)
    pw, ph, pn, pf = cube.raytrace(R.Rays.points(lat=yy.ravel(), lon=xx.ravel(), los=los.reshape(-1, 3)), 120.0, zref)
    assert np.array_equal(pn, dn)
    # (same arithmetic, but a different instantiation of pass 1 - crossings_kernel is specialised by input form, and the compiler
    # contracts a*b+c differently in each: a few ulp in the ray origin, 1e-13 m in the delay)
    np.testing.assert_allclose(pw.reshape(ny, nx), hw, rtol=0, atol=1e-12); np.testing.assert_allclose(ph.reshape(ny, nx), hh, rtol=0, atol=1e-12)


def test_two_contexts_in_two_threads(R):
    """SURVEY 8(b) threading contract: a context is not thread-safe, DISTINCT contexts are.  Two threads, each with its own
    context / cube / stream, trace different scenes concurrently (ctypes releases the GIL); results equal the serial ones."""
    import threading
    c = O.synthetic_cube(50, 50, 40, seed=0)
    zref = float(c['zs'].max() - 1)
    scenes = [(np.linspace(-119.5, -115.5, 300), np.linspace(34.5, 31.5, 280), 33.0), (np.linspace(-118.0, -116.0, 310), np.linspace(33.0, 32.0, 290), 41.0)]

    def run(ctx, scene, out, reps):
        cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx', ctx=ctx)
        xp, yp, inc = scene
        for _ in range(reps):
            w, h, n, _f = cube.raytrace(R.Rays.grid(xp, yp, inc=inc, hd=-167.9), 0.0, zref)
        out.append((w, h, n))

    serial = []
    for s in scenes:
        run(R.Context(0), s, serial, 1)
    outs = [[], []]
    threads = [threading.Thread(target=run, args=(R.Context(0), scenes[i], outs[i], 6)) for i in range(2)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(2):
        assert len(outs[i]) == 1
        assert np.array_equal(outs[i][0][0], serial[i][0]) and np.array_equal(outs[i][0][1], serial[i][1]) and np.array_equal(outs[i][0][2], serial[i][2])


def test_one_context_shared_by_four_threads(R):
    """The Python layer serialises calls into ONE context (a lock per context; the C context itself is not thread-safe and ctypes
    drops the GIL): four threads tracing different scenes, interpolating stations and building zenith cubes through the SAME
    context and cube get exactly the serial results, and an argument error raised in one thread carries that thread's message."""
    import threading
    c = O.synthetic_cube(50, 50, 40, seed=0)
    zref = float(c['zs'].max() - 1)
    ctx = R.Context(0)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx', ctx=ctx)
    rng = np.random.default_rng(0)
    jobs = []
    for t in range(4):
        nx, ny = 150 + 37 * t, 140 + 29 * t
        jobs.append((np.linspace(-119.5 + 0.1 * t, -115.5, nx), np.linspace(34.5, 31.5 + 0.1 * t, ny), 30.0 + 4 * t,
                     np.stack([rng.uniform(31, 35, 5000), rng.uniform(-120, -115, 5000), rng.uniform(0, 9000, 5000)], -1)))

    def work(job, out, reps):
        xp, yp, inc, pts = job
        for _ in range(reps):
            w, h, n, _f = cube.raytrace(R.Rays.grid(xp, yp, inc=inc, hd=-167.9), 0.0, zref)
            pw, ph = cube.interp(pts)
            zw, zh = cube.build_cube(xp[:40], yp[:30], np.array([0.0, 1000.0]))
            try:
                cube.raytrace(R.Rays.grid(xp, yp, inc=inc, hd=-167.9), 0.0, zref, max_seg=-1.0 - inc)
                msg = None
            except ValueError as e:
                msg = str(e)
        out.append((w, h, n, pw, ph, zw, zh, msg))

    serial = []
    for j in jobs:
        work(j, serial, 1)
    outs = [[] for _ in jobs]
    threads = [threading.Thread(target=work, args=(jobs[i], outs[i], 5)) for i in range(4)]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    for i in range(4):
        assert len(outs[i]) == 1
        for a, b in zip(outs[i][0][:7], serial[i][:7]):
            assert np.array_equal(a, b, equal_nan=True)
        assert outs[i][0][7] is not None and 'MAX_SEGMENT_LENGTH' in outs[i][0][7]


def test_lcc_projection_against_snyders_worked_examples(R):
    """The device's Lambert-conformal-conic forward (rdr_project_points) on Snyder's published numerical examples (USGS PP 1395,
    pp. 295-298; see tests/test_oracle_golden.py): sphere and Clarke 1866 ellipsoid."""
    ys, xs, zs = np.linspace(0.0, 1.0, 4), np.linspace(0.0, 1.0, 4), np.linspace(0.0, 1.0, 4)
    cube = R.Cube(ys, xs, zs, np.zeros((4, 4, 4), np.float32), np.zeros((4, 4, 4), np.float32), order='zyx')
    par = dict(lat_1=33.0, lat_2=45.0, lat_0=23.0, lon_0=-96.0)
    cube.set_projection_lcc(a=1.0, es=0.0, **par)
    y, x = cube.project(np.array([35.0]), np.array([-75.0]))
    assert abs(x[0] - 0.2966785) < 5e-8 and abs(y[0] - 0.2462112) < 5e-8
    cube.set_projection_lcc(a=6378206.4, es=0.00676866, **par)
    y, x = cube.project(np.array([35.0]), np.array([-75.0]))
    assert abs(x[0] - 1894410.9) < 0.05 and abs(y[0] - 1564649.5) < 0.05


def test_raytrace_slices_bit_identical_to_slice_loop(R):
    """rdr_raytrace_slices (the height loop of _build_cube_ray as ONE launch pair): every slice must come out exactly as
    rdr_raytrace integrates it alone - own level table, per-level slice maxima, nParts, z-clamp decision, NaN pattern - for
    heights below the model, inside it, equal to a model node and above the integration top (K = 0: zeros), with ragged tile
    edges, shared and per-slice look vectors, host and device arrays, a Lambert cube and a scene of generic-geodesy rays."""
    import torch
    dev = torch.device('cuda')
    c = O.synthetic_cube(40, 44, 36, seed=3)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    zref = float(c['zs'].max() - 1)
    ny, nx = 53, 71
    xpts = np.linspace(-119.5, -115.5, nx); ypts = np.linspace(34.5, 31.5, ny)
    hts = np.array([-150.0, 0.0, 777.7, float(c['zs'][5]), 9000.0, 25000.0, zref + 10.0])
    rng = np.random.default_rng(11)
    inc = rng.uniform(20, 50, (ny, nx)); hd = np.full((ny, nx), -167.9)

    def loop(make_rays):
        res = []
        for s, ht in enumerate(hts):
            try:
                w, h, npz, fl = cube.raytrace(make_rays(s), float(ht), zref)
                res.append((w, h, npz, fl))
            except R.NoLevels:
                res.append(None)
        return res

    def check(batch, ref):
        w, h, K, nparts, flags = batch
        for s in range(len(hts)):
            if ref[s] is None:
                assert K[s] == 0 and not np.any(np.asarray(w[s].cpu() if hasattr(w, 'cpu') else w[s])) and not np.any(np.asarray(h[s].cpu() if hasattr(h, 'cpu') else h[s]))
                continue
            rw, rh, rnp, rfl = ref[s]
            assert K[s] == len(rnp) and np.array_equal(nparts[s, :K[s]], rnp) and flags[s] == rfl
            ws = w[s].cpu().numpy() if hasattr(w, 'cpu') else w[s]; hs = h[s].cpu().numpy() if hasattr(h, 'cpu') else h[s]
            rw = rw.cpu().numpy() if hasattr(rw, 'cpu') else rw; rh = rh.cpu().numpy() if hasattr(rh, 'cpu') else rh
            assert np.array_equal(ws, rw, equal_nan=True) and np.array_equal(hs, rh, equal_nan=True)

    # (a) incidence / heading rasters shared by every slice, host arrays
    ref = loop(lambda s: R.Rays.grid(xpts, ypts, inc=inc, hd=hd))
    assert ref[-1] is None and ref[0] is not None
    check(cube.raytrace_slices(R.Rays.grid(xpts, ypts, inc=inc, hd=hd), hts, zref), ref)
    # (b) per-slice look vectors (as orbit-based LOS gives them), device arrays
    los = np.stack([O.look_vectors_from_inc_hd(inc + 0.3 * s, hd, *np.meshgrid(ypts, xpts, indexing='ij'), float(ht)) for s, ht in enumerate(hts)])
    xt, yt = torch.from_numpy(xpts).to(dev), torch.from_numpy(ypts).to(dev)
    lt = torch.from_numpy(los).to(dev)
    ref = loop(lambda s: R.Rays.grid(xt, yt, los=lt[s].contiguous()))
    check(cube.raytrace_slices(R.Rays.grid(xt, yt, los=lt, slices=len(hts)), hts, zref), ref)
    # (c) shared look vectors, host arrays, maxseg 400
    ref2 = []
    for s, ht in enumerate(hts):
        try:
            ref2.append(cube.raytrace(R.Rays.grid(xpts, ypts, los=los[1]), float(ht), zref, 400.0))
        except R.NoLevels:
            ref2.append(None)
    check(cube.raytrace_slices(R.Rays.grid(xpts, ypts, los=los[1]), hts, zref, 400.0), ref2)
    # (d) point-list origins
    xx, yy = np.meshgrid(xpts, ypts)
    refp = loop(lambda s: R.Rays.points(lat=yy.ravel().copy(), lon=xx.ravel().copy(), inc=inc.ravel().copy(), hd=hd.ravel().copy()))
    check(cube.raytrace_slices(R.Rays.points(lat=yy.ravel().copy(), lon=xx.ravel().copy(), inc=inc.ravel().copy(), hd=hd.ravel().copy()), hts, zref), refp)
    # (e) workspace smaller than the batch (groups of slices) and smaller than ONE slice (chunked per slice)
    ctx = R.Context.default()
    ref = loop(lambda s: R.Rays.grid(xpts, ypts, inc=inc, hd=hd))
    for lim in (3 << 20, 1 << 20):
        ctx.set_workspace_limit(lim)
        try:
            check(cube.raytrace_slices(R.Rays.grid(xpts, ypts, inc=inc, hd=hd), hts, zref), ref)
        finally:
            ctx.set_workspace_limit(48 << 30)
    # (f) generic-geodesy rays (polar scene) and a Lambert cube
    cp = O.synthetic_cube(40, 44, 36, seed=7, y0=80.0, y1=89.9, x0=-60.0, x1=60.0)
    cubep = R.Cube(cp['ys'], cp['xs'], cp['zs'], cp['wet'], cp['hydro'], order='zyx')
    yp = np.linspace(89.0, 83.0, 37); xp = np.linspace(-20.0, 20.0, 41)
    incp = rng.uniform(15, 55, (37, 41)); hdp = rng.uniform(-180, 180, (37, 41))
    hp = np.array([0.0, 1200.0, 8000.0])
    zrefp = float(cp['zs'].max() - 1)
    refq = [cubep.raytrace(R.Rays.grid(xp, yp, inc=incp, hd=hdp), float(h), zrefp) for h in hp]
    wq, hq, Kq, npq, flq = cubep.raytrace_slices(R.Rays.grid(xp, yp, inc=incp, hd=hdp), hp, zrefp)
    for s in range(3):
        assert np.array_equal(npq[s, :Kq[s]], refq[s][2]) and np.array_equal(wq[s], refq[s][0], equal_nan=True) and np.array_equal(hq[s], refq[s][1], equal_nan=True)
    proj = dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=-97.5)
    xs = np.linspace(-2.0e6, -1.0e6, 44); ys = np.linspace(-9.0e5, 1.0e5, 40)
    cubel = R.Cube(ys, xs, c['zs'], c['wet'], c['hydro'], order='zyx').set_projection_lcc(**proj)
    lon = np.linspace(-118.0, -113.5, nx); lat = np.linspace(36.0, 32.0, ny)
    refl = [cubel.raytrace(R.Rays.grid(lon, lat, inc=inc, hd=hd), float(h), zref) for h in hts[:5]]
    wl, hl, Kl, npl, fll = cubel.raytrace_slices(R.Rays.grid(lon, lat, inc=inc, hd=hd), hts[:5], zref)
    assert np.isfinite(wl).mean() > 0.5
    for s in range(5):
        assert np.array_equal(npl[s, :Kl[s]], refl[s][2]) and np.array_equal(wl[s], refl[s][0], equal_nan=True) and np.array_equal(hl[s], refl[s][1], equal_nan=True)


def test_build_cube_ray_batched_equals_slice_loop(R):
    """_build_cube_ray routes the whole height loop through rdr_raytrace_slices; handing it output arrays (which it then
    accumulates into, delay.py:245-248,323) takes the slice-by-slice loop: the two must agree bit for bit, including the all-zero
    top slice above the integration top and the reference's TypeError when a NON-top slice has no contributing level."""
    from raider_amd.delay import _build_cube_ray
    from raider_amd.losreader import Raytracing
    c = O.synthetic_cube(30, 34, 28, seed=2)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    from raider_amd.delayFcns import FieldInterpolator
    fi = [FieldInterpolator(cube, 0), FieldInterpolator(cube, 1)]
    xpts = np.linspace(-119.0, -116.0, 45); ypts = np.linspace(34.0, 32.0, 33)
    zref = float(c['zs'].max() - 1)
    zpts = np.array([0.0, 500.0, 2500.0, 9000.0, zref + 5.0])
    los = Raytracing(inc=np.full((33, 45), 37.0), heading=-167.9)
    a = _build_cube_ray(xpts, ypts, zpts, los, 4326, 4326, fi, MAX_TROPO_HEIGHT=zref)
    b = [np.zeros((5, 33, 45)), np.zeros((5, 33, 45))]
    _build_cube_ray(xpts, ypts, zpts, los, 4326, 4326, fi, outputArrs=b, MAX_TROPO_HEIGHT=zref)
    assert np.array_equal(a[0], b[0]) and np.array_equal(a[1], b[1])
    assert not a[0][-1].any() and a[0][0].min() > 0
    with pytest.raises(TypeError):
        _build_cube_ray(xpts, ypts, np.array([0.0, zref + 5.0, 100.0]), los, 4326, 4326, fi, MAX_TROPO_HEIGHT=zref)


AK = dict(lat_0=90.0, lat_ts=60.0, lon_0=225.0, a=6371229.0, es=0.0)          # HRRR-AK: models/hrrr.py:22-25,359


def test_polar_stereographic_projection_on_device(R):
    """rdr_cube_set_projection(RDR_PROJ_STERE) + rdr_project_points: Snyder's worked example (south polar, ellipsoid, lat_ts),
    HRRR-AK's spherical north-polar CRS and the k_0 variant against the oracle restatement, poles included."""
    ys, xs, zs = np.linspace(0.0, 1.0, 4), np.linspace(0.0, 1.0, 4), np.linspace(0.0, 1.0, 4)
    cube = R.Cube(ys, xs, zs, np.zeros((4, 4, 4), np.float32), np.zeros((4, 4, 4), np.float32), order='zyx')
    cube.set_projection_stere(lat_0=-90.0, lat_ts=-71.0, lon_0=-100.0, a=6378388.0, es=0.00672267)
    y, x = cube.project(np.array([-75.0]), np.array([150.0]))
    assert abs(x[0] - (-1540033.6)) < 0.05 and abs(y[0] - (-560526.4)) < 0.05
    rng = np.random.default_rng(0)
    for par, lat in ((AK, rng.uniform(40, 90, 2000)), (dict(lat_0=-90.0, lat_ts=-71.0, lon_0=0.0, a=6378137.0, es=0.0066943799901413165), rng.uniform(-90, -50, 2000)),
                     (dict(lat_0=90.0, lat_ts=None, k_0=0.994, lon_0=-45.0, a=6378137.0, es=0.0066943799901413165), rng.uniform(45, 90, 2000))):
        lon = rng.uniform(-180, 180, lat.size)
        lat[:2] = par['lat_0']                                   # the pole itself
        cube.set_projection_stere(**par)
        y, x = cube.project(lat, lon)
        ox, oy = O.stere_forward(lat, lon, **par)
        np.testing.assert_allclose(x, ox, rtol=0, atol=2e-6); np.testing.assert_allclose(y, oy, rtol=0, atol=2e-6)
    with pytest.raises(ValueError):
        cube.set_projection_stere(lat_0=45.0, lat_ts=None, lon_0=0.0)            # oblique aspect: not built in


@pytest.mark.parametrize('case', ['hrrr_ak_sphere', 'south_ellipsoid'])
def test_raytrace_and_zenith_through_polar_stereographic_cube(R, case):
    """The ray tracer and the zenith gather on a weather cube whose x / y axes are polar-stereographic metres (HRRR-AK's grid:
    ecef_to_model = 4978 -> `+proj=stere`, delay.py:252-253,295): against the oracle with the same projection, 1e-9 m.  The
    spherical north-polar cube takes the light ray path (relative projection, the cone of constant n = 1), the ellipsoidal
    south-polar one the generic kernels."""
    rng = np.random.default_rng(4)
    if case == 'hrrr_ak_sphere':
        par = dict(AK); lat_c, lon_c = 63.0, -150.0
    else:
        par = dict(lat_0=-90.0, lat_ts=-71.0, lon_0=0.0, a=6378137.0, es=0.0066943799901413165); lat_c, lon_c = -72.0, 40.0
    c = O.synthetic_cube(40, 44, 36, seed=9, ztop=26000.0)
    # a 3-km grid around the scene
    cx, cy = O.stere_forward(lat_c, lon_c, **par)
    xs = cx + 3000.0 * (np.arange(44) - 22); ys = cy + 3000.0 * (np.arange(40) - 20)
    proj = dict(par, proj='stere')
    cube = R.Cube(ys, xs, c['zs'], c['wet'], c['hydro'], order='zyx').set_projection_stere(**par)
    ip = list(O.getInterpolators(xs, ys, c['zs'], c['wet'], c['hydro']))
    ny, nx = 21, 26
    dlat = 0.45; dlon = 0.45 / np.cos(np.radians(lat_c))
    ypts = np.linspace(lat_c + dlat, lat_c - dlat, ny); xpts = np.linspace(lon_c - dlon, lon_c + dlon, nx)
    inc = rng.uniform(18, 48, (ny, nx)); hd = rng.uniform(-180, 180, (ny, nx))
    zref = float(c['zs'].max() - 1)
    for ht in (0.0, 900.0):
        look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(inc, hd, llh[1], llh[0], llh[2])
        (ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([ht]), look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True, model_proj=proj)
        wet, hyd, nparts, _ = cube.raytrace(R.Rays.grid(xpts, ypts, inc=inc, hd=hd), ht, zref)
        assert np.array_equal(nparts, onp[0])
        assert np.array_equal(np.isnan(wet), np.isnan(ow[0])) and np.isfinite(ow[0]).mean() > 0.6
        np.testing.assert_allclose(wet, ow[0], rtol=0, atol=TIGHT, equal_nan=True)
        np.testing.assert_allclose(hyd, oh[0], rtol=0, atol=TIGHT, equal_nan=True)
    tot = R.Cube(ys, xs, c['zs'], c['wet_total'], c['hydro_total'], order='zyx').set_projection_stere(**par)
    it = list(O.getInterpolators(xs, ys, c['zs'], c['wet_total'], c['hydro_total']))
    zw, zh = tot.build_cube(xpts, ypts, np.array([0.0, 1500.0]))
    ozw, ozh = O.build_cube(xpts, ypts, np.array([0.0, 1500.0]), it, model_proj=proj)
    np.testing.assert_allclose(zw, ozw, rtol=0, atol=1e-11, equal_nan=True); np.testing.assert_allclose(zh, ozh, rtol=0, atol=1e-11, equal_nan=True)
    # through the public API with the CRS given as the PROJ string the HRRR-AK model carries
    from raider_amd.delay import _build_cube_ray
    from raider_amd.delayFcns import FieldInterpolator
    from raider_amd.losreader import Raytracing
    crs = ('+proj=stere +lat_0={lat_0} +lon_0={lon_0} +lat_ts={lat_ts} +a={a} +b={b} +units=m'
           .format(b=par['a'] * np.sqrt(1 - par['es']), **par))
    cube2 = R.Cube(ys, xs, c['zs'], c['wet'], c['hydro'], order='zyx')
    res = _build_cube_ray(xpts, ypts, np.array([0.0]), Raytracing(inc=inc, heading=hd), crs, 4326,
                          [FieldInterpolator(cube2, 0), FieldInterpolator(cube2, 1)], MAX_TROPO_HEIGHT=zref)
    (ow, oh) = O.build_cube_ray(xpts, ypts, np.array([0.0]), look, ip, MAX_TROPO_HEIGHT=zref, model_proj=proj)
    np.testing.assert_allclose(res[0][0], ow[0], rtol=0, atol=TIGHT, equal_nan=True)
    np.testing.assert_allclose(res[1][0], oh[0], rtol=0, atol=TIGHT, equal_nan=True)


def test_height_sharding_needs_no_collective(R):
    """raider_amd.distributed.raytrace_heights_sharded: three "ranks" taking disjoint blocks of the output heights reproduce the
    one-launch cube bit for bit - slices share nothing (own level table, maxima, nParts), so this sharding has no all-reduce."""
    from raider_amd import distributed as D
    c = O.synthetic_cube(30, 34, 28, seed=4, y0=31.0, y1=35.0, x0=-120.0, x1=-115.0)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    xp = np.linspace(-118.5, -116.5, 37); yp = np.linspace(33.9, 32.1, 29)
    hts = np.array([-50.0, 0.0, 300.0, 1200.0, 2500.0, 6000.0, 11000.0])
    zref = float(c['zs'].max() - 1)
    inc = np.broadcast_to(np.linspace(25, 44, 37), (29, 37)).copy()
    rays_for = lambda h: R.Rays.grid(xp, yp, inc=inc, hd=-167.9)
    w0, h0, K0, np0, f0 = cube.raytrace_slices(rays_for(hts), hts, zref)
    seen = np.zeros(hts.size, bool)
    for rank in range(3):
        a, n, w, h, K, npr, fl = D.raytrace_heights_sharded(cube, rays_for, hts, zref, world=3, rank=rank)
        assert n > 0 and not seen[a:a + n].any()
        seen[a:a + n] = True
        assert np.array_equal(w, w0[a:a + n]) and np.array_equal(h, h0[a:a + n]) and np.array_equal(npr, np0[a:a + n]) and np.array_equal(K, K0[a:a + n])
    assert seen.all()
    assert D.raytrace_heights_sharded(cube, rays_for, hts[:2], zref, world=3, rank=2)[1] == 0          # more ranks than heights
