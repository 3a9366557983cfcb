"""CPU: the oracle (oracle/raider_oracle.py) against the golden vectors produced by running the
reference itself (oracle/refharness/gen_golden.py).  This is what pins the oracle."""
import numpy as np
import pytest

from oracle import raider_oracle as O

TOL = 1e-9   # metres; observed ~1e-13.  north_star tolerance for delays is 1e-6 m.


def look_fn(inc, hd):
    def f(ht, llh, xyz, yy):
        i = np.broadcast_to(np.asarray(inc, float), yy.shape)
        h = np.broadcast_to(np.asarray(hd, float), yy.shape)
        return O.look_vectors_from_inc_hd(i, h, llh[1], llh[0], llh[2])
    return f


def interps(cube, kind):
    sfx = '_total' if kind == 'total' else ''
    return list(O.getInterpolators(cube['xs'], cube['ys'], cube['zs'], cube['wet' + sfx], cube['hydro' + sfx]))


@pytest.fixture(scope='module')
def c1():
    return O.synthetic_cube(50, 50, 40, seed=0)


def test_g1_makepoints(golden):
    g = golden('g1_makepoints')
    assert np.array_equal(O.makePoints(1000., g['a0_sp'], g['a0_slv'], 5.), g['a0_out'])
    assert np.array_equal(O.makePoints(1000., g['a1_sp'], g['a1_slv'], 5.), g['a1_out'])
    assert np.array_equal(O.makePoints(20., g['a2_sp'], g['a2_slv'], 5), g['a2_out'])
    assert np.array_equal(O.makePoints(100., g['a3_sp'], g['a3_slv'], 5), g['a3_out'])
    assert np.allclose(O.makePoints(100., g['a3_sp'], g['a3_slv'], 5), g['a3_txt'])
    ml, st = g['r2_args']
    assert np.array_equal(O.makePoints(ml, g['r2_sp'], g['r2_slv'], st), g['r2_out'])


@pytest.mark.parametrize('nd', [1, 2, 3, 4])
def test_g2_native_interpolate(golden, nd):
    g = golden('g2_interpolate')
    grids = [g[f'd{nd}_grid{k}'] for k in range(nd)]
    vals, q = g[f'd{nd}_vals'], g[f'd{nd}_q']
    np.testing.assert_allclose(O.native_interpolate(grids, vals, q, fill_value=np.nan), g[f'd{nd}_fill'],
                               rtol=1e-13, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(O.native_interpolate(grids, vals, q), g[f'd{nd}_extrap'], rtol=1e-12, atol=1e-12)
    np.testing.assert_allclose(O.native_interpolate(grids, vals, q, fill_value=7.0), g[f'd{nd}_fill7'],
                               rtol=1e-13, atol=1e-13)
    # the last-node quirk: row 1 of the appended node queries sits on every last node
    assert np.isnan(g[f'd{nd}_fill'][401])


@pytest.mark.parametrize('ax', [0, 1, 2])
def test_g2_along_axis(golden, ax):
    g = golden('g2_interpolate')
    P, V, Q = g[f'ax{ax}_P'], g[f'ax{ax}_V'], g[f'ax{ax}_Q']
    np.testing.assert_allclose(O.native_interpolate_along_axis(P, V, Q, axis=ax, fill_value=np.nan),
                               g[f'ax{ax}_fill'], rtol=1e-13, atol=1e-13, equal_nan=True)
    np.testing.assert_allclose(O.native_interpolate_along_axis(P, V, Q, axis=ax), g[f'ax{ax}_extrap'],
                               rtol=1e-12, atol=1e-12)


def test_g3_rays(golden):
    g = golden('g3_rays')
    lat, lon, zs = g['lat'], g['lon'], g['model_zs']
    for ht in (-500, 0, 2500):
        xyz = np.stack(O.lla2ecef(lat, lon, np.full(lat.shape, float(ht))), -1)
        np.testing.assert_allclose(xyz, g[f'xyz_ht{ht}'], rtol=0, atol=1e-8)
        for inc in (0, 20, 39, 55):
            for hd, hdt in ((-167.9, -167), (-12.1, -12)):
                tag = f'ht{ht}_inc{inc}_hd{hdt}'
                los = O.look_vectors_from_inc_hd(np.full(lat.shape, float(inc)), np.full(lat.shape, hd), lat, lon, float(ht))
                np.testing.assert_allclose(los, g[f'los_{tag}'], rtol=0, atol=1e-15)
                L, lo, hi = O.build_ray(zs, float(ht), xyz, los, 30000.0)
                np.testing.assert_allclose(L, g[f'len_{tag}'], rtol=0, atol=1e-8)
                np.testing.assert_allclose(lo[[0, 1, -1]], g[f'low_{tag}'], rtol=0, atol=1e-8)
                np.testing.assert_allclose(hi[[0, 1, -1]], g[f'high_{tag}'], rtol=0, atol=1e-8)
        los = O.look_vectors_from_inc_hd(np.full(lat.shape, 39.0), np.full(lat.shape, -167.9), lat, lon, float(ht))
        np.testing.assert_allclose(O.getTopOfAtmosphere(xyz, los, 15000.0), g[f'toa10_ht{ht}'], atol=1e-8, rtol=0)
        np.testing.assert_allclose(O.getTopOfAtmosphere(xyz, los, 15000.0, factor=np.full(lat.shape, 0.77)),
                                   g[f'toa3_ht{ht}'], atol=1e-8, rtol=0)
    llh = np.stack(O.ecef2lla(g['geo_xyz'][..., 0], g['geo_xyz'][..., 1], g['geo_xyz'][..., 2]), -1)
    np.testing.assert_allclose(llh, g['geo_llh'], rtol=0, atol=1e-9)


def test_g4_build_cube(golden, c1):
    g = golden('g4_build_cube')
    wet, hydro = O.build_cube(g['xpts'], g['ypts'], g['zpts'], interps(c1, 'total'))
    np.testing.assert_allclose(wet, g['wet'], rtol=0, atol=1e-14)
    np.testing.assert_allclose(hydro, g['hydro'], rtol=0, atol=1e-14)
    wet2, hydro2 = O.build_cube(g['xp2'], g['yp2'], g['zp2'], interps(c1, 'total'))
    assert np.isnan(g['wet2']).any() and not np.isnan(g['wet2']).all()
    np.testing.assert_allclose(wet2, g['wet2'], rtol=0, atol=1e-14, equal_nan=True)
    np.testing.assert_allclose(hydro2, g['hydro2'], rtol=0, atol=1e-14, equal_nan=True)


def test_g5_build_cube_ray(golden, c1):
    g = golden('g5_build_cube_ray')
    zref = float(g['c1_zref'])
    ip = interps(c1, 'pointwise')
    for tag, inc in (('fixed', 39.0), ('pp', g['c1_pp_inc'])):
        (wet, hydro), nparts = O.build_cube_ray(g['c1_xpts'], g['c1_ypts'], g['c1_zpts'], look_fn(inc, -167.9), ip,
                                                MAX_TROPO_HEIGHT=zref, return_nparts=True)
        for i in range(2):
            assert np.array_equal(nparts[i], g[f'c1_{tag}_nparts{i}'])
        np.testing.assert_allclose(wet, g[f'c1_{tag}_wet'], rtol=0, atol=TOL)
        np.testing.assert_allclose(hydro, g[f'c1_{tag}_hydro'], rtol=0, atol=TOL)
    (wet, hydro), nparts = O.build_cube_ray(g['c1_xpts'], g['c1_ypts'], np.array([100.0]), look_fn(20.0, -12.1), ip,
                                            MAX_SEGMENT_LENGTH=500.0, MAX_TROPO_HEIGHT=26000.0, return_nparts=True)
    assert np.array_equal(nparts[0], g['c1_z26_nparts0'])
    np.testing.assert_allclose(wet, g['c1_z26_wet'], rtol=0, atol=TOL)
    np.testing.assert_allclose(hydro, g['c1_z26_hydro'], rtol=0, atol=TOL)
    # lateral exit -> NaN in exactly the same pixels
    wet, hydro = O.build_cube_ray(g['c1_edge_xpts'], g['c1_edge_ypts'], np.array([0.0]), look_fn(45.0, -167.9), ip,
                                  MAX_TROPO_HEIGHT=zref)
    assert np.isnan(g['c1_edge_wet']).any() and not np.isnan(g['c1_edge_wet']).all()
    np.testing.assert_allclose(wet, g['c1_edge_wet'], rtol=0, atol=TOL, equal_nan=True)
    np.testing.assert_allclose(hydro, g['c1_edge_hydro'], rtol=0, atol=TOL, equal_nan=True)


def test_g5_constant_refractivity_invariant(golden, c1):
    """N==1  =>  delay*1e6 == sum of ray lengths (test/test_synthetic.py:217-274)."""
    g = golden('g5_build_cube_ray')
    cube1 = dict(c1); cube1['wet'] = np.ones_like(c1['wet']); cube1['hydro'] = np.ones_like(c1['hydro'])
    zref = float(g['c1_zref'])
    wet, hydro = O.build_cube_ray(g['c1_one_xpts'], g['c1_one_ypts'], g['c1_zpts'], look_fn(39.0, -167.9),
                                  interps(cube1, 'pointwise'), MAX_TROPO_HEIGHT=zref)
    np.testing.assert_allclose(wet, g['c1_one_wet'], rtol=0, atol=TOL)
    xx, yy = np.meshgrid(g['c1_one_xpts'], g['c1_one_ypts'])
    for i, ht in enumerate(g['c1_zpts']):
        xyz = np.stack(O.lla2ecef(yy, xx, np.full(yy.shape, ht)), -1)
        los = look_fn(39.0, -167.9)(ht, [xx, yy, np.full(yy.shape, ht)], xyz, yy)
        L, _, _ = O.build_ray(c1['zs'], ht, xyz, los, zref)
        np.testing.assert_allclose(wet[i] * 1e6, L.sum(0), rtol=1e-13)


def test_g5_big_cube(golden):
    g = golden('g5_build_cube_ray')
    big = O.synthetic_cube(300, 300, 80, seed=0)
    (wet, hydro), nparts = O.build_cube_ray(g['big_xpts'], g['big_ypts'], np.array([0.0]), look_fn(g['big_inc'], -167.9),
                                            interps(big, 'pointwise'), MAX_TROPO_HEIGHT=float(g['big_zref']),
                                            return_nparts=True)
    assert np.array_equal(nparts[0], g['big_nparts0'])
    np.testing.assert_allclose(wet, g['big_wet'], rtol=0, atol=TOL)
    np.testing.assert_allclose(hydro, g['big_hydro'], rtol=0, atol=TOL)


def test_g5b_halves_need_whole_slice_nparts(golden, c1):
    """shards must be driven with the batch-global nParts."""
    g = golden('g5b_whole_vs_halves')
    ip = interps(c1, 'pointwise')
    zref = float(g['zref'])
    xp, yp, inc = g['xpts'], g['ypts'], g['inc']
    for sl, side in ((slice(0, 32), 'left'), (slice(32, 64), 'right')):
        wet, hydro = O.build_cube_ray(xp[sl], yp, np.array([0.0]), look_fn(inc[:, sl], -167.9), ip,
                                      MAX_TROPO_HEIGHT=zref, nParts_override=[g['nparts']])
        np.testing.assert_allclose(hydro, g['hydro'][:, :, sl], rtol=0, atol=TOL)
        np.testing.assert_allclose(wet, g['wet'][:, :, sl], rtol=0, atol=TOL)
        # and shard-local nParts reproduces the reference run on the half (which differs from the whole)
        wet, hydro = O.build_cube_ray(xp[sl], yp, np.array([0.0]), look_fn(inc[:, sl], -167.9), ip, MAX_TROPO_HEIGHT=zref)
        np.testing.assert_allclose(hydro, g[f'{side}_hydro'], rtol=0, atol=TOL)
    assert np.abs(g['left_hydro'] - g['hydro'][:, :, :32]).max() > 1e-6


def test_g6_los_tables(golden):
    g = golden('g6_los')
    enu = O.inc_hd_to_enu(g['inc'], g['hd'])
    np.testing.assert_array_equal(enu, g['enu'])
    ecef = O.enu2ecef(enu[..., 0], enu[..., 1], enu[..., 2], g['lat'], g['lon'], 0 * g['lat'])
    np.testing.assert_array_equal(ecef, g['ecef'])
    np.testing.assert_array_equal(O.ecef2enu(ecef, g['lat'], g['lon'], 0 * g['lat']), g['enu_back'])
    np.testing.assert_array_equal(O.getZenithLookVecs(g['lat'], g['lon'], 0 * g['lat']), g['zen'])
    np.testing.assert_array_equal(O.conventional_project(g['delays'], enu), g['proj_last'])
    # the three ECEF values the reference test pins exactly (test/test_delayFcns.py:86-99):
    # transformPoints returns (y, x, z)
    x, y, z = O.lla2ecef(np.array([0., 0., 0.]), np.array([0., 90., 180.]), np.zeros(3))
    np.testing.assert_allclose(np.stack([y, x, z], -1), g['tp_equator'], atol=1e-9)
    np.testing.assert_allclose(np.stack([x, y, z], -1),
                               [[6378137.0, 0, 0], [0, 6378137.0, 0], [-6378137.0, 0, 0]], atol=1e-9)
    with pytest.raises(ValueError):
        O.inc_hd_to_enu(np.array([-1.0]), np.array([0.0]))


def test_g8_point_branch(golden, c1):
    g = golden('g8_points')
    hl = g['height_levels']
    wet, hydro = O.build_cube(g['xpts'], g['ypts'], hl, interps(c1, 'total'))
    w, h = O.points_from_cube(g['lats'], g['lons'], g['hgts'], g['xpts'], g['ypts'], hl, wet, hydro)
    np.testing.assert_allclose(w, g['wet_zen'], rtol=0, atol=1e-13)
    np.testing.assert_allclose(h, g['hydro_zen'], rtol=0, atol=1e-13)
    zref = c1['zs'].max() - 1
    wet, hydro = O.build_cube_ray(g['xpts_ray'], g['ypts_ray'], hl, look_fn(39.0, -167.9), interps(c1, 'pointwise'),
                                  MAX_TROPO_HEIGHT=zref)
    w, h = O.points_from_cube(g['lats'][:200], g['lons'][:200], g['hgts'][:200], g['xpts_ray'], g['ypts_ray'], hl, wet, hydro)
    np.testing.assert_allclose(w, g['wet_ray'], rtol=0, atol=TOL)
    np.testing.assert_allclose(h, g['hydro_ray'], rtol=0, atol=TOL)


def test_rgi_matches_scipy_edges():
    """Oracle RGI vs the real scipy (same third-party the reference calls): nodes, last node, NaN, outside,
    f32 values, descending axis."""
    from scipy.interpolate import RegularGridInterpolator
    rng = np.random.default_rng(5)
    ys = np.linspace(3, -2, 6); xs = np.sort(rng.uniform(0, 5, 7)); zs = np.array([0., 1., 3., 7.])
    v = rng.normal(size=(6, 7, 4)).astype(np.float32)
    sp = RegularGridInterpolator((ys, xs, zs), v, fill_value=np.nan, bounds_error=False)
    mine = O.RGI((ys, xs, zs), v)
    q = np.stack([rng.uniform(-2.5, 3.5, 500), rng.uniform(-0.5, 5.5, 500), rng.uniform(-1, 8, 500)], -1)
    q[:6] = [[3, xs[0], 0], [-2, xs[-1], 7], [ys[2], xs[3], zs[1]], [np.nan, 1, 1], [0, 1, 7.0000001], [0, 1, 7]]
    np.testing.assert_allclose(mine(q), sp(q), rtol=0, atol=1e-15, equal_nan=True)
    assert mine(q).dtype == np.float64


def test_time_weights_and_blend():
    w1, w2 = O.time_weights(1800.0, 0.0, 3600.0)
    assert (w1, w2) == (0.5, 0.5)
    w1, w2 = O.time_weights(900.0, 0.0, 3600.0)
    assert np.isclose(w1 + w2, 1) and np.isclose(w1, 0.75)
    a = np.arange(6, dtype=np.float32); b = a[::-1].copy()
    out = O.blend_cubes(0.5, a, 0.5, b)
    assert out.dtype == np.float32 and np.allclose(out, 2.5)   # mean of epochs (test_temporal_interpolate.py)
    assert O.blend_cubes(0.25, a.astype(np.float64), 0.75, b.astype(np.float64)).dtype == np.float64


@pytest.mark.parametrize('tag', ['q', 'rh'])
def test_g10_cube_producer(golden, tag):
    """models/weatherModel.py:235-262 restated (oracle.cube_from_model_levels) vs the reference's own WeatherModel."""
    g = golden('g10_cube_producer')
    r = O.cube_from_model_levels(g[f'{tag}_zs'], g[f'{tag}_p'], g[f'{tag}_t'], g[f'{tag}_hum'], tag, g[f'{tag}_newz'])
    np.testing.assert_allclose(r['e_levels'], g[f'{tag}_e_levels'], rtol=1e-15)
    for k in ('t', 'p', 'e'):
        assert np.array_equal(r[f'{k}_u'], g[f'{tag}_{k}_u'], equal_nan=True)          # after _uniform_in_z (with NaNs), f32
        assert np.array_equal(r[k], g[f'{tag}_{k}_out'])                                 # after _checkForNans + _adjust_grid
    assert np.array_equal(r['zs'], g[f'{tag}_out_zs'])
    assert r['wet'].dtype == np.float32 and np.array_equal(r['wet'], g[f'{tag}_wet']) and np.array_equal(r['hydro'], g[f'{tag}_hydro'])
    np.testing.assert_allclose(r['wet_total'], g[f'{tag}_wet_total'], rtol=1e-14, atol=1e-18)
    np.testing.assert_allclose(r['hydro_total'], g[f'{tag}_hydro_total'], rtol=1e-14, atol=1e-18)
    assert (tag == 'q') == (r['zs'].size == g[f'{tag}_newz'].size + 1)                   # q case is padded, rh case is not


def test_g10_fillna3d(golden):
    """interpolator.py:110-130 (pandas interpolate, limit_direction='backward') vs oracle.fillna_columns, incl. interior runs"""
    g = golden('g10_cube_producer')
    a = g['holes_in']
    assert np.array_equal(O.fillna_columns(a, 0.0), g['holes_fill0'])
    assert np.array_equal(O.fillna_columns(a, 1e16), g['holes_fill1e16'])


def test_g11_azimuth_time_weighting(golden):
    """s1_azimuth_timing.py date selection and inverse weights, cli/raider.py:817-819 combination, vs the reference's own
    functions (golden g11) and the known answers of test/test_s1_time_grid.py:157-205,358-386."""
    import datetime as dt
    g = golden('g11_aztime_weights')
    for spec, want in zip(g['closest_in'], g['closest_out']):
        t, n, step = str(spec).split('|')
        got = O.n_closest_datetimes(dt.datetime.fromisoformat(t), int(n), int(step))
        assert '|'.join(x.isoformat() for x in got) == str(want)
    for spec, want in zip(g['aztimes_in'], g['aztimes_out']):
        t, step, buf = str(spec).split('|')
        got = O.times_for_azimuth_interpolation(dt.datetime.fromisoformat(t), int(step), int(buf))
        assert '|'.join(x.isoformat() for x in got) == str(want)
    assert O.n_closest_datetimes(dt.datetime(2023, 1, 1, 11, 1, 1), 3, 6) == [dt.datetime(2023, 1, 1, 12), dt.datetime(2023, 1, 1, 6), dt.datetime(2023, 1, 1, 18)]
    with pytest.raises(ValueError):
        O.n_closest_datetimes(dt.datetime(2023, 1, 1, 20, 1, 1), 2, 5)
    for tag in ('w3', 'w3b', 'w3c', 'w2'):
        win = float(g[f'{tag}_window_h'])
        w = O.inverse_time_weights(g[f'{tag}_ms'] * 1e-3, g[f'{tag}_dates_s'], None if np.isnan(win) else win * 3600.0, float(g[f'{tag}_reg']))
        np.testing.assert_allclose(w, g[f'{tag}_weights'], rtol=1e-13, atol=1e-18)
        np.testing.assert_allclose(w.sum(0), 1.0, rtol=1e-14)
    # test_inverse_weighting's table (test_s1_time_grid.py:208-216): 07:00 between 06:00 / 12:00 / 00:00, 6 h window
    w = O.inverse_time_weights(np.array([7 * 3600.0]), [6 * 3600.0, 12 * 3600.0, 0.0], 6 * 3600.0)
    np.testing.assert_allclose(w[:, 0], [.833, .167, 0.0], atol=1e-3)
    with pytest.raises(ValueError):
        O.inverse_time_weights(np.zeros(3), [1.0, 1.0])
    comb = O.combine_weighted(list(g['w3b_weights']), list(g['comb_fields']))
    assert comb.dtype == np.float64 and np.array_equal(comb, g['comb_out'])


def test_uniform_in_z_known_answer():
    """test/test_weather_model.py::test_uniform_in_z_small replayed on the oracle's restatement of _uniform_in_z
    (weatherModel.py:603-629): uneven per-column heights averaging to [1, 2]; a query ON a column's lowest node returns the
    value, ON its top node the NaN fill (the native interpolate_along_axis edge rule)."""
    nan = np.nan
    zs = np.array([[[1., 2.], [0.9, 1.1]], [[1., 2.6], [1.1, 2.3]]])
    p = np.arange(8, dtype=np.float64).reshape(2, 2, 2)
    new_z = np.nanmean(zs, axis=(0, 1))
    np.testing.assert_allclose(new_z, [1.0, 2.0], rtol=0, atol=1e-15)
    r = O.cube_from_model_levels(zs, p, p * 2, np.zeros_like(p), 'q', new_z, zmin=1.0)
    want = np.array([[[0, nan], [2.5, nan]], [[4., 4.625], [nan, 6.75]]])
    assert np.allclose(r['p_u'], want, equal_nan=True, rtol=0)
    assert np.allclose(r['t_u'], want * 2, equal_nan=True, rtol=0)


def _mock_model():
    """test/test_weather_model.py:96-136 MockWeatherModel.load_weather (k1 = k2 = k3 = 1)"""
    nz = 32
    ys = np.arange(-2, 3) + 0.0; xs = np.arange(-3, 4) + 0.0
    zs = np.linspace(0, 1e5, nz)
    t = np.ones((ys.size, xs.size, nz)); e = t.copy(); e[:, 3:, :] = 2
    pl = np.arange(31, -1, -1).astype(np.float64)
    p = np.broadcast_to(pl, t.shape).copy()
    true_wet_ztd = 1e-6 * 2 * np.broadcast_to(np.flip(zs), t.shape).copy(); true_wet_ztd[:, 3:] *= 2
    true_hydro_ztd = np.zeros(t.shape)
    for layer in range(nz):
        true_hydro_ztd[:, :, layer] = 1e-6 * 0.5 * (zs[-1] - zs[layer]) * pl[layer]
    true_wet_refr = 2 * np.ones(t.shape); true_wet_refr[:, 3:] = 4
    return dict(ys=ys, xs=xs, zs=zs, t=t, e=e, p=p, wet_refr=true_wet_refr, hydro_refr=p.copy(), wet_ztd=true_wet_ztd, hydro_ztd=true_hydro_ztd)


def test_reference_known_answers_find_svp_and_ztd():
    """test/test_weather_model.py::test_find_svp (10 tabulated saturation pressures) and ::test_ztd (MockWeatherModel with
    closed-form refractivities and ZTDs) replayed on the oracle's restatements."""
    svp_true = np.array([611.21, 1227.5981, 2337.2825, 4243.5093, 7384.1753, 12369.2295, 20021.443, 31419.297, 47940.574, 71305.16])
    assert np.allclose(O.find_svp(np.arange(0, 100, 10) + 273.15), svp_true)
    m = _mock_model()
    wet = 1 * m['e'] / m['t'] + 1 * m['e'] / m['t'] ** 2                          # weatherModel.py:355-357 with k2 = k3 = 1
    hyd = 1 * m['p'] / m['t']
    assert np.allclose(wet, m['wet_refr']) and np.allclose(hyd, m['hydro_refr'])
    assert np.allclose(O.ztd_totals(wet, m['zs']), m['wet_ztd']) and np.allclose(O.ztd_totals(hyd, m['zs']), m['hydro_ztd'])


def test_lcc_against_snyders_worked_examples():
    """Lambert conformal conic forward: J. P. Snyder, Map Projections - A Working Manual (USGS Prof. Paper 1395, 1987), numerical
    examples pp. 295-298: standard parallels 33 and 45 N, origin 23 N / 96 W, point 35 N / 75 W.  Unit sphere: x = 0.2966785,
    y = 0.2462112; Clarke 1866 ellipsoid (a = 6378206.4 m, e^2 = 0.00676866): x = 1 894 410.9 m, y = 1 564 649.5 m.  An authority
    independent of PROJ (which implements the same formulas) for the projection the HRRR cubes live on."""
    par = dict(lat_1=33.0, lat_2=45.0, lat_0=23.0, lon_0=-96.0)
    x, y = O.lcc_forward(35.0, -75.0, a=1.0, es=0.0, **par)
    assert abs(x - 0.2966785) < 5e-8 and abs(y - 0.2462112) < 5e-8
    x, y = O.lcc_forward(35.0, -75.0, a=6378206.4, es=0.00676866, **par)
    assert abs(x - 1894410.9) < 0.05 and abs(y - 1564649.5) < 0.05


def test_wgs84_conversion_against_the_epsg_guidance_note_example():
    """Geographic <-> geocentric (EPSG method 9602), the worked example of IOGP Guidance Note 7-2 (WGS 84): 53 48'33.820" N,
    2 07'46.380" E, h = 73.0 m  <->  X = 3 771 793.968 m, Y = 140 253.342 m, Z = 5 124 304.349 m.  An authority independent of PROJ
    for utilFcns.lla2ecef / ecef2lla (which the reference delegates to pyproj)."""
    lat, lon, h = 53 + 48 / 60 + 33.820 / 3600, 2 + 7 / 60 + 46.380 / 3600, 73.0
    x, y, z = O.lla2ecef(np.array([lat]), np.array([lon]), np.array([h]))
    assert abs(x[0] - 3771793.968) < 1e-3 and abs(y[0] - 140253.342) < 1e-3 and abs(z[0] - 5124304.349) < 1e-3
    lo, la, hh = O.ecef2lla(np.array([3771793.968]), np.array([140253.342]), np.array([5124304.349]))
    assert abs(la[0] - lat) < 1e-8 and abs(lo[0] - lon) < 1e-8 and abs(hh[0] - h) < 1e-3


def test_polar_stereographic_against_snyders_worked_example():
    """Polar stereographic forward: J. P. Snyder, Map Projections - A Working Manual (USGS PP 1395, 1987), numerical example for
    the ellipsoidal polar aspect with a latitude of true scale (p. 315): International ellipsoid (a = 6 378 388.0 m, e^2 =
    0.00672267), lat_ts = -71, lon_0 = -100, point (75 S, 150 E) -> t = 0.1325120, t_c = 0.1684118, x = -1 540 033.6 m,
    y = -560 526.4 m.  An authority independent of PROJ (which is absent here); plus the closed-form properties the
    projection must have (pole -> origin, true scale on lat_ts, the meridian lon_0 maps to x = 0, north = mirrored south)."""
    x, y = O.stere_forward(-75.0, 150.0, lat_0=-90.0, lat_ts=-71.0, lon_0=-100.0, a=6378388.0, es=0.00672267)
    assert abs(x - (-1540033.6)) < 0.05 and abs(y - (-560526.4)) < 0.05
    # HRRR-AK's own CRS (models/hrrr.py:22-25): sphere, lat_ts 60, lon_0 225
    ak = dict(lat_0=90.0, lat_ts=60.0, lon_0=225.0, a=6371229.0, es=0.0)
    x, y = O.stere_forward(90.0, 17.0, **ak)
    assert abs(x) < 1e-6 and abs(y) < 1e-6
    x, y = O.stere_forward(np.array([55.0, 70.0]), np.array([-135.0, -135.0]), **ak)                  # lon_0 = 225 = -135: on the central meridian
    assert np.all(np.abs(x) < 1e-6) and np.all(y < 0) and y[0] < y[1]
    # true scale on lat_ts: a small meridional step has the length R dphi, a small step along the parallel R cos(phi) dlam
    d = 1e-6
    x0, y0 = O.stere_forward(60.0, -150.0, **ak); x1, y1 = O.stere_forward(60.0 + d, -150.0, **ak); x2, y2 = O.stere_forward(60.0, -150.0 + d, **ak)
    R0 = 6371229.0
    assert abs(np.hypot(x1 - x0, y1 - y0) / (R0 * np.radians(d)) - 1) < 1e-6
    assert abs(np.hypot(x2 - x0, y2 - y0) / (R0 * np.cos(np.radians(60.0)) * np.radians(d)) - 1) < 1e-6
    # scale k_0 at the pole (variant A) on the sphere: rho = 2 R k0 tan(pi/4 - phi/2)
    x, y = O.stere_forward(80.0, 90.0, lat_0=90.0, lat_ts=None, k_0=0.994, lon_0=0.0, a=R0, es=0.0)
    assert abs(x - 2 * R0 * 0.994 * np.tan(np.radians(5.0))) < 1e-6 and abs(y) < 1e-6
    # the southern aspect is the mirrored northern one
    xn, yn = O.stere_forward(72.0, 33.0, lat_0=90.0, lat_ts=70.0, lon_0=-45.0, a=6378137.0, es=0.0066943799901413165)
    xs_, ys_ = O.stere_forward(-72.0, 33.0, lat_0=-90.0, lat_ts=-70.0, lon_0=-45.0, a=6378137.0, es=0.0066943799901413165)
    assert abs(xn - xs_) < 1e-6 and abs(yn + ys_) < 1e-6


def test_orbit_solver_restatement_on_a_circular_orbit_and_the_reference_state_vectors():
    """tests/orbit_anchor.py: the oracle's zero-Doppler restatement against the closed form of a circular orbit (azimuth time
    lon / w, law-of-cosines range, look vector), and its Hermite interpolant against the Sentinel-1 state vectors of the
    reference's own fixture (every other vector predicts the skipped ones)."""
    from tests import orbit_anchor as A
    st, sp, sv = A.circular_orbit()
    T, t0, rg0, los0 = A.targets(np.random.default_rng(0))
    los, t, rg = O.orbit_look_vectors(st, sp, sv, T)
    assert np.isfinite(t).all()
    assert np.abs(t - t0).max() < 1e-6 and np.abs(rg - rg0).max() < 1e-3 and np.abs(los - los0).max() < 1e-8
    # Hermite through vectors 0, 2, 4, 6 (20 s apart) at the times of 1, 3, 5: a 4th-order interpolant of a 7 km/s orbit
    pos, vel = O.orbit_hermite(A.S1_T[::2], A.S1_POS[::2], A.S1_VEL[::2], A.S1_T[1:6:2])
    assert np.abs(pos - A.S1_POS[1:6:2]).max() < 2e-2 and np.abs(vel - A.S1_VEL[1:6:2]).max() < 2e-3
    # and through ALL of them it reproduces the nodes and stays on the orbit in between (radius / speed vary smoothly)
    pos, vel = O.orbit_hermite(A.S1_T, A.S1_POS, A.S1_VEL, A.S1_T)
    assert np.abs(pos - A.S1_POS).max() < 1e-6 and np.abs(vel - A.S1_VEL).max() < 1e-9


def test_transverse_mercator_against_published_examples():
    """Transverse Mercator (the transformPoints step of a UTM output grid): Snyder's UTM example (USGS PP 1395 p. 269: Clarke 1866,
    (40 30' N, 73 30' W), central meridian 75 W, k_0 = 0.9996 -> x = 127 106.5 m, y = 4 484 124.4 m) and the Transverse-Mercator
    example of IOGP Guidance Note 7-2 (OSGB 1936 / British National Grid: Airy 1830, lat_0 49 N, lon_0 2 W, k_0 = 0.9996012717,
    FE 400 000, FN -100 000; (50 30' N, 0 30' E) -> E 577 274.99 m, N 69 740.50 m - computed there with the truncated USGS series,
    hence centimetres).  Plus forward / inverse closure and the defining properties (scale k_0 on the central meridian)."""
    x, y = O.tm_forward(40.5, -73.5, lat_0=0.0, lon_0=-75.0, k_0=0.9996, x_0=0.0, y_0=0.0, a=6378206.4, es=0.00676866)
    assert abs(x - 127106.5) < 0.05 and abs(y - 4484124.4) < 0.05
    f = 1 / 299.32496
    osgb = dict(lat_0=49.0, lon_0=-2.0, k_0=0.9996012717, x_0=400000.0, y_0=-100000.0, a=6377563.396, es=2 * f - f * f)
    x, y = O.tm_forward(50.5, 0.5, **osgb)
    assert abs(x - 577274.99) < 0.02 and abs(y - 69740.50) < 0.02
    lat, lon = O.tm_inverse(577274.99, 69740.50, **osgb)
    assert abs(lat - 50.5) < 2e-7 and abs(lon - 0.5) < 2e-7
    rng = np.random.default_rng(0)
    utm11 = dict(lat_0=0.0, lon_0=-117.0, k_0=0.9996, x_0=500000.0, y_0=0.0)
    la = rng.uniform(-80, 84, 2000); lo = rng.uniform(-123, -111, 2000)
    x, y = O.tm_forward(la, lo, **utm11)
    la2, lo2 = O.tm_inverse(x, y, **utm11)
    assert np.abs(la2 - la).max() < 1e-11 and np.abs(lo2 - lo).max() < 1e-11
    # on the central meridian x = x_0 and dy/dlat = k_0 * meridional radius of curvature
    x0, y0 = O.tm_forward(35.0, -117.0, **utm11); x1, y1 = O.tm_forward(35.0 + 1e-6, -117.0, **utm11)
    es = 0.0066943799901413165; M = 6378137.0 * (1 - es) / (1 - es * np.sin(np.radians(35.0)) ** 2) ** 1.5
    assert abs(x0 - 500000.0) < 1e-6 and abs((y1 - y0) / np.radians(1e-6) / (0.9996 * M) - 1) < 1e-6


def test_conic_inverses_against_snyders_worked_examples():
    """The way back of transformPoints for the conic model CRSs (test/test_delayFcns.py:67-84 round-trips it through pyproj): Snyder's
    inverse numerical examples (USGS PP 1395: LCC pp. 296-298 on the unit sphere and the Clarke 1866 ellipsoid; polar stereographic
    p. 317) and forward / inverse closure over both hemispheres and both cones."""
    par = dict(lat_1=33.0, lat_2=45.0, lat_0=23.0, lon_0=-96.0)
    la, lo = O.lcc_inverse(0.2966785, 0.2462112, a=1.0, es=0.0, **par)
    assert abs(la - 35.0) < 5e-6 and abs(lo + 75.0) < 5e-6                          # (7 printed digits on the unit sphere)
    la, lo = O.lcc_inverse(1894410.9, 1564649.5, a=6378206.4, es=0.00676866, **par)
    assert abs(la - 35.0) < 5e-7 and abs(lo + 75.0) < 5e-7
    la, lo = O.stere_inverse(-1540033.6, -560526.4, lat_0=-90.0, lat_ts=-71.0, lon_0=-100.0, a=6378388.0, es=0.00672267)
    assert abs(la + 75.0) < 5e-7 and abs(lo - 150.0) < 5e-7
    rng = np.random.default_rng(0)
    lo = rng.uniform(-180, 180, 2000)
    for kw, la in ((dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, a=6371229.0, es=0.0), rng.uniform(10, 70, 2000)),
                   (dict(lat_1=-30.0, lat_2=-60.0, lat_0=-45.0, lon_0=20.0, x_0=1e6, y_0=2e6, a=6378137.0, es=0.0066943799901413165), rng.uniform(-80, -5, 2000))):
        lon = kw['lon_0'] + 0.9 * lo                                                  # (stay off the cone's cut meridian)
        x, y = O.lcc_forward(la, lon, **kw)
        la2, lo2 = O.lcc_inverse(x, y, **kw)
        assert np.abs(la2 - la).max() < 1e-12 and np.abs((lo2 - lon + 180) % 360 - 180).max() < 1e-12
    for kw, la in ((dict(lat_0=90.0, lat_ts=60.0, lon_0=225.0, a=6371229.0, es=0.0), rng.uniform(30, 89.9, 2000)),
                   (dict(lat_0=-90.0, lat_ts=-71.0, lon_0=0.0, a=6378137.0, es=0.0066943799901413165), rng.uniform(-89.9, -40, 2000)),
                   (dict(lat_0=90.0, lat_ts=None, k_0=0.994, lon_0=-45.0, a=6378137.0, es=0.0066943799901413165), rng.uniform(40, 89.9, 2000))):
        x, y = O.stere_forward(la, lo, **kw)
        la2, lo2 = O.stere_inverse(x, y, **kw)
        assert np.abs(la2 - la).max() < 1e-12 and np.abs((lo2 - lo + 180) % 360 - 180).max() < 1e-11


def test_every_reference_made_fixture_says_what_it_was_made_on(golden):
    """Round 5: each fixture written by oracle/refharness/gen_golden.py carries `_meta` - what the reference's third-party
    geometry ran on when it was made (`geodesy`: the builder's pyproj stand-in today, `pyproj <ver> / PROJ <ver>` the day the image has it;
    `look_vectors`: isce3 or absent).  g12 is a cut of a data file (no geometry), g14 exists only when isce3 made it."""
    import json
    from pathlib import Path
    gold = Path(__file__).resolve().parent / 'golden'
    made = sorted(p.stem for p in gold.glob('g*.npz') if p.stem != 'g12_gmao_time_interp')
    assert len(made) >= 12
    for name in made:
        g = golden(name)
        assert '_meta' in g.files, name
        meta = json.loads(str(g['_meta']))
        assert set(meta) >= {'geodesy', 'look_vectors', 'numpy', 'generator'} and meta['generator'].endswith('gen_golden.py'), (name, meta)
        assert meta['geodesy'].startswith(('builder stub', 'pyproj ')) and meta['look_vectors'].startswith(('absent', 'isce3 ')), (name, meta)
        if name.startswith('g14'):
            assert meta['look_vectors'].startswith('isce3 ')
