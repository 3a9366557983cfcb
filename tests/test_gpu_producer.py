"""GPU cube producer (rdr_cubes_from_model_levels) vs the reference's WeatherModel processing chain
(models/weatherModel.py:235-262): golden g10 (made by the reference itself) and the NumPy oracle."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

# t, p and hydro only go through IEEE +,-,*,/ in a fixed order -> bit-exact.  e (and wet, built on it) passes through exp():
# the device libm and glibc may round the last bit differently, which shows up as <= 1 ulp of f32 after the cast.
E_RTOL = 2.5e-7


def _run(zs, p, t, hum, tag, newz, **kw):
    from raider_amd.weather import cubes_from_model_levels
    A, B, _ = zs.shape
    xs = np.arange(B) * 0.25 - 118.0
    ys = np.arange(A) * 0.25 + 33.0
    return cubes_from_model_levels(xs, ys, zs, p, t, hum, tag, newz, return_state=True, **kw)


@pytest.mark.parametrize('tag', ['q', 'rh'])
def test_producer_matches_reference_golden(golden, tag):
    g = golden('g10_cube_producer')
    m = _run(g[f'{tag}_zs'], g[f'{tag}_p'], g[f'{tag}_t'], g[f'{tag}_hum'], tag, g[f'{tag}_newz'])
    assert np.array_equal(m.zs, g[f'{tag}_out_zs'])
    assert m.pointwise.dtype == np.float32 and m.total.dtype == np.float64
    assert np.array_equal(m.t, g[f'{tag}_t_out'])
    assert np.array_equal(m.p, g[f'{tag}_p_out'])
    np.testing.assert_allclose(m.e, g[f'{tag}_e_out'], rtol=E_RTOL, atol=0)
    wet, hyd = m.pointwise.read()
    assert np.array_equal(hyd, g[f'{tag}_hydro'])
    np.testing.assert_allclose(wet, g[f'{tag}_wet'], rtol=2 * E_RTOL, atol=0)
    wt, ht = m.total.read()
    np.testing.assert_allclose(ht, g[f'{tag}_hydro_total'], rtol=1e-13, atol=1e-18)
    np.testing.assert_allclose(wt, g[f'{tag}_wet_total'], rtol=2 * E_RTOL, atol=1e-18)
    assert (np.mean(m.e == g[f'{tag}_e_out'])) > 0.9            # and nearly all of e is bit-identical too


def test_producer_vs_oracle_with_missing_data():
    """NaN holes in the model columns (interior runs, whole columns) -> fillna3D semantics on the device"""
    from oracle import raider_oracle as O
    rng = np.random.default_rng(77)
    A, B, nl = 9, 11, 37
    base = np.sort(rng.uniform(0, 1, (A, B, nl)), axis=2)
    zs = -80.0 + 300.0 * rng.uniform(0, 1, (A, B, 1)) + 42000.0 * base ** 1.4
    t = np.maximum(289.0 - 0.0063 * zs + rng.normal(0, 0.4, zs.shape), 203.0)
    p = 101000.0 * np.exp(-zs / 7700.0)
    q = 0.011 * np.exp(-zs / 2500.0)
    t[1, 2, 10:14] = np.nan                   # interior hole -> NaN in the resampled column around those heights
    t[3, 3, :] = np.nan                       # dead column: t -> 1e16, p/e untouched
    p[4, 5, 20:] = np.nan
    newz = np.concatenate([np.arange(-100.0, 3000.0, 150.0), np.arange(3000.0, 44000.0, 1500.0)])
    r = O.cube_from_model_levels(zs, p, t, q, 'q', newz)
    m = _run(zs, p, t, q, 'q', newz)
    assert np.array_equal(m.zs, r['zs'])
    assert np.array_equal(m.t, r['t']) and np.array_equal(m.p, r['p'])
    np.testing.assert_allclose(m.e, r['e'], rtol=E_RTOL, atol=0)
    wet, hyd = m.pointwise.read()
    assert np.array_equal(hyd, r['hydro'])
    np.testing.assert_allclose(wet, r['wet'], rtol=2 * E_RTOL, atol=1e-30)
    wt, ht = m.total.read()
    np.testing.assert_allclose(ht, r['hydro_total'], rtol=1e-13, atol=1e-18)
    np.testing.assert_allclose(wt, r['wet_total'], rtol=2 * E_RTOL, atol=1e-18)


@pytest.mark.parametrize('nzo', [64, 65, 66, 129, 193])
def test_producer_top_level_total_is_zero_at_strip_boundaries(nzo):
    """the suffix scan of _getZTD walks the output levels in strips of 64; with (nzo - 1) % 64 == 0 (65, 129, 193 levels)
    the top level used to belong to no strip and its total stayed unwritten LDS.  Its value is the empty sum: exactly 0 - and every
    other level agrees with the oracle."""
    from oracle import raider_oracle as O
    rng = np.random.default_rng(nzo)
    A, B, nl = 5, 6, 30
    base = np.sort(rng.uniform(0, 1, (A, B, nl)), axis=2)
    zs = -50.0 + 200.0 * rng.uniform(0, 1, (A, B, 1)) + 40000.0 * base ** 1.4
    t = np.maximum(288.0 - 0.0063 * zs, 205.0); p = 101000.0 * np.exp(-zs / 7700.0); q = 0.010 * np.exp(-zs / 2500.0)
    newz = np.linspace(-100.0, 38000.0, nzo)
    for _ in range(2):                                     # (twice: the second run sees the LDS the first one left behind)
        m = _run(zs, p, t, q, 'q', newz)
        r = O.cube_from_model_levels(zs, p, t, q, 'q', newz)
        wt, ht = m.total.read()
        assert wt.shape[2] == nzo and np.all(wt[..., -1] == 0.0) and np.all(ht[..., -1] == 0.0)
        np.testing.assert_allclose(ht, r['hydro_total'], rtol=1e-13, atol=1e-18)
        np.testing.assert_allclose(wt, r['wet_total'], rtol=2 * E_RTOL, atol=1e-18)


def test_producer_with_many_levels_needs_large_lds():
    """500 model levels resampled to 300 heights: the per-wavefront column buffers are 98 KB per workgroup, past the 64 KB a launch
    gets by default.  Same parity as above; a level count whose buffers exceed the device's LDS is refused by name."""
    from oracle import raider_oracle as O
    from raider_amd.weather import cubes_from_model_levels
    rng = np.random.default_rng(5)
    A, B, nl = 4, 5, 500
    base = np.sort(rng.uniform(0, 1, (A, B, nl)), axis=2)
    zs = -80.0 + 300.0 * rng.uniform(0, 1, (A, B, 1)) + 42000.0 * base ** 1.4
    t = np.maximum(289.0 - 0.0063 * zs + rng.normal(0, 0.4, zs.shape), 203.0)
    p = 101000.0 * np.exp(-zs / 7700.0)
    q = 0.011 * np.exp(-zs / 2500.0)
    newz = np.linspace(-100.0, 41000.0, 300)
    r = O.cube_from_model_levels(zs, p, t, q, 'q', newz)
    m = _run(zs, p, t, q, 'q', newz)
    assert np.array_equal(m.zs, r['zs']) and np.array_equal(m.t, r['t']) and np.array_equal(m.p, r['p'])
    wet, hyd = m.pointwise.read()
    assert np.array_equal(hyd, r['hydro'])
    np.testing.assert_allclose(wet, r['wet'], rtol=2 * E_RTOL, atol=1e-30)
    wt, ht = m.total.read()
    np.testing.assert_allclose(ht, r['hydro_total'], rtol=1e-13, atol=1e-18)
    big = np.repeat(zs[..., :1], 1000, axis=2) + np.arange(1000.0) * 40.0                 # 1000 levels -> 480 heights: 171 KB
    with pytest.raises(Exception, match='LDS'):
        cubes_from_model_levels(np.arange(B) * 0.25 - 118.0, np.arange(A) * 0.25 + 33.0, big, 101000.0 * np.exp(-big / 7700.0), 280.0 + 0 * big, 0.001 + 0 * big, 'q',
                                np.linspace(-100.0, 41000.0, 480))


def test_producer_feeds_the_delay_path():
    """producer cubes -> tropo_delay-style zenith cube and a ray-traced slice, without leaving the device"""
    import torch
    from oracle import raider_oracle as O
    from raider_amd import Rays
    from raider_amd.weather import cubes_from_model_levels, MODEL_LEVEL_HEIGHTS
    rng = np.random.default_rng(5)
    A, B, nl = 40, 48, 60
    xs = -119.0 + 0.1 * np.arange(B)
    ys = 32.0 + 0.1 * np.arange(A)
    base = np.linspace(0, 1, nl)[None, None, :] ** 1.6
    zs = -60.0 + 100.0 * rng.uniform(0, 1, (A, B, 1)) + 45000.0 * base
    t = np.maximum(288.0 - 0.0065 * zs, 210.0)
    p = 101325.0 * np.exp(-zs / 7600.0)
    rh = np.clip(75.0 * np.exp(-zs / 8000.0), 1.0, 100.0)
    dev = torch.device('cuda:0')
    m = cubes_from_model_levels(xs, ys, *(torch.from_numpy(a).to(dev) for a in (zs, p, t, rh)), 'rh', MODEL_LEVEL_HEIGHTS)
    r = O.cube_from_model_levels(zs, p, t, rh, 'rh', MODEL_LEVEL_HEIGHTS)
    assert m.pointwise.shape == (A, B, MODEL_LEVEL_HEIGHTS.size)            # table starts at zmin: no pad
    # zenith cube (delay._build_cube on the 'total' fields)
    xq, yq, zq = xs[3:-3:2] + 0.013, ys[3:-3:2] + 0.021, np.array([0.0, 450.0, 2200.0, 9000.0])
    wz, hz = m.total.build_cube(xq, yq, zq)
    tot_ip = [O.RGI((ys, xs, r['zs']), r['wet_total']), O.RGI((ys, xs, r['zs']), r['hydro_total'])]
    ow, oh = O.build_cube(xq, yq, zq, tot_ip)
    np.testing.assert_allclose(hz, oh, rtol=1e-12, atol=1e-14)
    np.testing.assert_allclose(wz, ow, rtol=2 * E_RTOL, atol=1e-14)
    assert 1.9 < hz[0].mean() < 2.6                                          # a sane hydrostatic ZTD in metres
    # one ray-traced slice through the pointwise cube
    xr, yr = xs[8:-8:3] + 0.01, ys[8:-8:3] + 0.02
    zref = 38000.0
    wr, hr, _, _ = m.pointwise.raytrace(Rays.grid(xr, yr, inc=35.0, hd=-12.0), 300.0, zref)
    pw_ip = [O.RGI((ys, xs, r['zs']), r['wet']), O.RGI((ys, xs, r['zs']), r['hydro'])]
    look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(np.full(yy.shape, 35.0), np.full(yy.shape, -12.0), llh[1], llh[0], llh[2])
    ow2, oh2 = O.build_cube_ray(xr, yr, np.array([300.0]), look, pw_ip, MAX_TROPO_HEIGHT=zref)
    np.testing.assert_allclose(hr, oh2[0], rtol=0, atol=1e-6)
    np.testing.assert_allclose(wr, ow2[0], rtol=0, atol=1e-6)


def test_producer_errors():
    from raider_amd.weather import cubes_from_model_levels
    z = np.tile(np.linspace(0, 30000, 8), (3, 3, 1))
    with pytest.raises(RuntimeError, match='Not a valid humidity type'):      # weatherModel.py:340-341
        cubes_from_model_levels(np.arange(3.0), np.arange(3.0), z, z, z, z, 'dewpoint', np.linspace(0, 20000, 5))
    with pytest.raises(ValueError):
        cubes_from_model_levels(np.arange(4.0), np.arange(3.0), z, z, z, z, 'q', np.linspace(0, 20000, 5))
    with pytest.raises(Exception, match='ascending'):
        cubes_from_model_levels(np.arange(3.0), np.arange(3.0), z, z + 1, z + 200, z * 0, 'q', np.linspace(20000, 0, 5))


def test_processed_model_is_a_weather_model_for_tropo_delay():
    """tropo_delay(dt, ProcessedModel, ...) == tropo_delay(dt, mapping read back from it, ...): the device-resident model is
    accepted wherever the reference takes the processed NetCDF (delay.py:35-130)."""
    import datetime as dt
    from raider_amd.delay import GridAOI, tropo_delay
    from raider_amd.losreader import Raytracing, Zenith
    from raider_amd.weather import cubes_from_model_levels, MODEL_LEVEL_HEIGHTS
    rng = np.random.default_rng(9)
    A, B, nl = 30, 34, 50
    xs = -100.0 + 0.2 * np.arange(B)
    ys = 20.0 + 0.2 * np.arange(A)
    zs = -70.0 + 120.0 * rng.uniform(0, 1, (A, B, 1)) + 44000.0 * np.linspace(0, 1, nl)[None, None, :] ** 1.7
    t = np.maximum(290.0 - 0.0064 * zs, 208.0)
    p = 101325.0 * np.exp(-zs / 7500.0)
    q = 0.013 * np.exp(-zs / 2300.0)
    m = cubes_from_model_levels(xs, ys, zs, p, t, q, 'q', MODEL_LEVEL_HEIGHTS)
    as_file = {k: m[k] for k in m.keys()}
    assert as_file['wet'].shape == (MODEL_LEVEL_HEIGHTS.size, A, B) and as_file['wet'].dtype == np.float32
    aoi = GridAOI(xs[4:-4:2] + 0.03, ys[4:-4:2] + 0.05)
    when = dt.datetime(2020, 1, 1)
    hl = [0.0, 800.0, 5000.0]
    for los in (Zenith(), Raytracing(inc=33.0, heading=-10.0)):
        a, _ = tropo_delay(when, m, aoi, los, hl)
        b, _ = tropo_delay(when, as_file, aoi, los, hl)
        assert np.array_equal(np.asarray(a['wet'][:]), np.asarray(b['wet'][:]))
        assert np.array_equal(np.asarray(a['hydro'][:]), np.asarray(b['hydro'][:]))
        assert np.isfinite(np.asarray(a['hydro'][:])).all()



@pytest.mark.parametrize('kind', ['hrrr', 'hrrr_ak'])
def test_projected_model_file_roundtrip_and_crs_without_pyproj(tmp_path, kind):
    """A processed model on HRRR's Lambert-conformal-conic grid / HRRR-AK's polar-stereographic grid written as the reference writes it
    (weatherModel.py:659-724: `proj` carries CRS.to_cf() = crs_wkt + CF grid-mapping attributes, 2-D geodetic latitude / longitude)
    and read back by tropo_delay by PATH: the model CRS comes from the CF attributes (or, stripped of them, from the WKT alone) -
    no pyproj - and the zenith and ray-traced cubes equal those of the in-memory model; latitude / longitude are the inverse
    projection of the grid nodes."""
    import datetime as dt
    from oracle import raider_oracle as O
    from raider_amd import crs, h5lite
    from raider_amd.delay import GridAOI, tropo_delay
    from raider_amd.losreader import Raytracing, Zenith
    from raider_amd.weather import cubes_from_model_levels, MODEL_LEVEL_HEIGHTS
    if kind == 'hrrr':
        P = dict(proj='lcc', lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, x_0=0.0, y_0=0.0, a=6371229.0, b=6371229.0)
        okw = dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, a=6371229.0, es=0.0)
        cx, cy = O.lcc_forward(34.0, -117.0, **okw); fwd, inv = O.lcc_forward, O.lcc_inverse
    else:
        P = dict(proj='stere', lat_0=90.0, lat_ts=60.0, lon_0=225.0, x_0=0.0, y_0=0.0, a=6371229.0, b=6371229.0)
        okw = dict(lat_0=90.0, lat_ts=60.0, lon_0=225.0, a=6371229.0, es=0.0)
        cx, cy = O.stere_forward(63.0, -150.0, **okw); fwd, inv = O.stere_forward, O.stere_inverse
    rng = np.random.default_rng(3)
    A, B, nl = 72, 76, 40                            # 216 x 228 km: rays to the 80 km top travel 46 km sideways
    xs = float(cx) + 3000.0 * (np.arange(B) - B / 2); ys = float(cy) + 3000.0 * (np.arange(A) - A / 2)
    zs = -70.0 + 120.0 * rng.uniform(0, 1, (A, B, 1)) + 44000.0 * np.linspace(0, 1, nl)[None, None, :] ** 1.7
    t = np.maximum(290.0 - 0.0064 * zs, 208.0); p = 101325.0 * np.exp(-zs / 7500.0); q = 0.013 * np.exp(-zs / 2300.0)
    m = cubes_from_model_levels(xs, ys, zs, p, t, q, 'q', MODEL_LEVEL_HEIGHTS)
    m.proj = P
    path = tmp_path / f'{kind}.nc'
    m.to_netcdf(path, time=dt.datetime(2020, 1, 1, 12), model_name='HRRR')
    f = h5lite.File(path)
    at = f['proj'].attrs
    assert at['grid_mapping_name'] == ('lambert_conformal_conic' if kind == 'hrrr' else 'polar_stereographic') and at['crs_wkt'].startswith('PROJCRS[')
    back = crs.crs_from_proj_var(at)
    assert back['proj'] == P['proj'] and all(abs(back[k] - P[k]) < 1e-9 for k in P if k != 'proj')
    assert crs.crs_from_wkt(at['crs_wkt']) == back
    xx, yy = np.meshgrid(xs, ys)
    la, lo = inv(xx, yy, **okw)
    np.testing.assert_allclose(f['latitude'].read(), la, rtol=0, atol=1e-10); np.testing.assert_allclose(f['longitude'].read(), lo, rtol=0, atol=1e-10)
    # output grid in lon/lat well inside the (rotated) model domain
    clat, clon = (float(v) for v in inv(float(cx), float(cy), **okw))
    aoi = GridAOI(np.linspace(clon - 0.3, clon + 0.3, 7), np.linspace(clat + 0.2, clat - 0.2, 6))
    when = dt.datetime(2020, 1, 1, 12)
    hl = [0.0, 900.0]
    wkt_only = {k: (f[k].read() if k != 'proj' else type('V', (), dict(attrs=dict(crs_wkt=at['crs_wkt'])))()) for k in ('x', 'y', 'z', 'wet', 'hydro', 'wet_total', 'hydro_total', 'proj')}
    for los in (Zenith(), Raytracing(inc=30.0, heading=-12.0)):
        a, _ = tropo_delay(when, m, aoi, los, hl)
        b, _ = tropo_delay(when, str(path), aoi, los, hl)
        c, _ = tropo_delay(when, wkt_only, aoi, los, hl)
        for name in ('wet', 'hydro'):
            ref = np.asarray(a[name][:])
            assert np.isfinite(ref).all() and ref.mean() > 0
            assert np.array_equal(ref, np.asarray(b[name][:])) and np.array_equal(ref, np.asarray(c[name][:]))
    # and against the oracle's own projection of the nodes: the zenith cube is the trilinear gather at the projected nodes
    za, _ = tropo_delay(when, str(path), aoi, Zenith(), hl)
    gx, gy = np.meshgrid(aoi.xpts, aoi.ypts)
    px, py = fwd(gy, gx, **okw)
    wt, ht = m.total.read()
    ip = list(O.getInterpolators(xs, ys, np.asarray(m.zs), wt.transpose(2, 0, 1), ht.transpose(2, 0, 1)))
    for k, h in enumerate(hl):
        pts = np.stack([py, px, np.full(px.shape, h)], -1)
        np.testing.assert_allclose(np.asarray(za['wet'][:])[k], ip[0](pts), rtol=0, atol=1e-12)
        np.testing.assert_allclose(np.asarray(za['hydro'][:])[k], ip[1](pts), rtol=0, atol=1e-12)
