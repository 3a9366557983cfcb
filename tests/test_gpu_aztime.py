"""Azimuth-time-grid temporal weighting on the GPU vs the reference's own functions (golden g11) and the oracle."""
import datetime as dt

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

EPOCH = dt.datetime(2021, 1, 1)


def _dates(g, tag):
    return [EPOCH + dt.timedelta(seconds=float(s)) for s in g[f'{tag}_dates_s']]


@pytest.mark.parametrize('tag', ['w3', 'w3b', 'w3c', 'w2'])
def test_inverse_weights_match_reference(golden, tag):
    from raider_amd.s1_azimuth_timing import get_inverse_weights_for_dates
    g = golden('g11_aztime_weights')
    grid = np.datetime64('2021-01-01T00:00:00', 'ms') + g[f'{tag}_ms'].astype('timedelta64[ms]')
    win = float(g[f'{tag}_window_h'])
    w = get_inverse_weights_for_dates(grid, _dates(g, tag), inverse_regularizer=float(g[f'{tag}_reg']),
                                      temporal_window_hours=None if np.isnan(win) else win)
    assert len(w) == len(g[f'{tag}_dates_s']) and w[0].shape == grid.shape
    # the grid times are taken relative to dates[0] (exactly representable), so only the last-bit rounding of the divisions differs
    np.testing.assert_allclose(np.stack(w), g[f'{tag}_weights'], rtol=4e-16, atol=1e-20)
    np.testing.assert_allclose(np.stack(w).sum(0), 1.0, rtol=1e-15)


def test_reference_known_answers_and_errors():
    """test/test_s1_time_grid.py:157-216,273-309,358-396 replayed against the mirror"""
    from raider_amd.s1_azimuth_timing import get_inverse_weights_for_dates, get_n_closest_datetimes, get_times_for_azimuth_interpolation, get_s1_azimuth_time_grid
    assert get_n_closest_datetimes(dt.datetime(2023, 1, 1, 11, 1, 1), 3, 6) == [dt.datetime(2023, 1, 1, 12), dt.datetime(2023, 1, 1, 6), dt.datetime(2023, 1, 1, 18)]
    assert get_n_closest_datetimes(dt.datetime(2023, 2, 1, 8, 1, 1), 4, 2) == [dt.datetime(2023, 2, 1, 8), dt.datetime(2023, 2, 1, 10), dt.datetime(2023, 2, 1, 6), dt.datetime(2023, 2, 1, 12)]
    assert get_n_closest_datetimes(dt.datetime(2023, 1, 1, 20, 1, 1), 2, 4) == [dt.datetime(2023, 1, 1, 20), dt.datetime(2023, 1, 2, 0)]
    assert get_n_closest_datetimes(dt.datetime(2023, 1, 2, 0, 0, 0), 3, 1) == [dt.datetime(2023, 1, 2, 0), dt.datetime(2023, 1, 1, 23), dt.datetime(2023, 1, 2, 1)]
    with pytest.raises(ValueError):
        get_n_closest_datetimes(dt.datetime(2023, 1, 1, 20, 1, 1), 2, 5)
    t = dt.datetime(2023, 1, 1, 11, 1, 0)
    assert get_times_for_azimuth_interpolation(t, 1) == [dt.datetime(2023, 1, 1, 11), dt.datetime(2023, 1, 1, 12), dt.datetime(2023, 1, 1, 10)]
    assert get_times_for_azimuth_interpolation(t, 3) == [dt.datetime(2023, 1, 1, 12), dt.datetime(2023, 1, 1, 9)]
    assert get_times_for_azimuth_interpolation(dt.datetime(2023, 1, 1, 11, 29, 0), 1) == [dt.datetime(2023, 1, 1, 11), dt.datetime(2023, 1, 1, 12)]
    # test_inverse_weighting's table
    dates = [dt.datetime(2021, 1, 1, 6), dt.datetime(2021, 1, 1, 12), dt.datetime(2021, 1, 1, 0)]
    for t0, win, want in ((np.datetime64('2021-01-01T07:00:00'), 6, [.833, .167, 0]), (np.datetime64('2021-01-01T07:00:00'), 3, [1., 0., 0.]),
                          (np.datetime64('2021-01-01T06:00:00'), 6, [1., 0., 0.])):
        grid = np.full((4, 4), t0, dtype='datetime64[ms]') + (np.timedelta64(1, 's') * np.arange(16)).reshape(4, 4)
        w = get_inverse_weights_for_dates(grid, dates, temporal_window_hours=win)
        for k in range(3):
            np.testing.assert_almost_equal(want[k], w[k], 1e-3)
        np.testing.assert_almost_equal(1, np.stack(w, axis=1).sum(axis=1))
    # test_triple_date_usage
    dates = [dt.datetime(2021, 1, 1, 0), dt.datetime(2021, 1, 1, 6), dt.datetime(2020, 12, 31, 18)]
    grid = np.full((3,), np.datetime64('2021-01-01T00:00:00'), dtype='datetime64[ms]') + np.timedelta64(1, 's') * np.array([-10_000, 0, 10_000])
    w0, w1, w2 = get_inverse_weights_for_dates(grid, dates, temporal_window_hours=6, inverse_regularizer=1e-10)
    assert all(w > 0 for w in w0) and (w1[0] <= 0) and (w1[2] > 0) and (w2[0] > 0) and (w2[2] <= 0)
    with pytest.raises(ValueError):
        get_inverse_weights_for_dates(np.zeros((3, 3)), [dt.datetime(2023, 1, 1)] * 2)
    with pytest.raises(ValueError, match='within temporal window'):
        get_inverse_weights_for_dates(np.full((3,), 9e5), dates, temporal_window_hours=6)
    with pytest.raises(ValueError):
        get_s1_azimuth_time_grid(np.arange(10), np.arange(11), np.zeros((12, 12)), dt.datetime(2023, 1, 1))
    with pytest.raises(ValueError):
        get_s1_azimuth_time_grid(np.zeros((3, 3, 3, 3)), np.arange(11), np.arange(12), dt.datetime(2023, 1, 1))


def test_weighted_combination_and_time_grid(golden):
    """cli/raider.py:817-819 with per-voxel weights: bit-exact vs the reference arithmetic (g11 comb_out); then the whole
    chain orbit -> time grid -> weights -> combined cubes against the oracle."""
    import torch
    from oracle import raider_oracle as O
    from raider_amd import Cube
    from raider_amd.orbits import Orbit
    from raider_amd.s1_azimuth_timing import combine_cubes, combine_weather_cubes_azimuth_time, get_azimuth_time_grid
    g = golden('g11_aztime_weights')
    f = g['comb_fields']                       # (3, nz, ny, nx) f32
    nz, ny, nx = f.shape[1:]
    ys, xs, zs = np.linspace(30, 31, ny), np.linspace(-118, -117, nx), np.linspace(0, 9000, nz)
    cubes = [Cube(ys, xs, zs, f[i], f[(i + 1) % 3], order='zyx') for i in range(3)]
    out = combine_cubes(cubes, list(g['w3b_weights']))
    wet, hyd = out.read()                      # (y, x, z) f64
    assert wet.dtype == np.float64
    assert np.array_equal(wet.transpose(2, 0, 1), g['comb_out'])
    want_h = O.combine_weighted(list(g['w3b_weights']), [f[1], f[2], f[0]])
    assert np.array_equal(hyd.transpose(2, 0, 1), want_h)
    # device-resident weights take the same path
    dev = torch.device('cuda:0')
    out2 = combine_cubes(cubes, [torch.from_numpy(w).to(dev) for w in g['w3b_weights']])
    assert np.array_equal(out2.read()[0], wet)
    # --- orbit -> time grid (a circular-ish test orbit crossing the scene)
    t = np.arange(-60.0, 61.0, 10.0)
    r, w_ = 7.07e6, 2 * np.pi / 5900.0
    lat0, lon0 = np.radians(30.5), np.radians(-100.0)
    pos = np.stack([r * np.cos(lat0 + w_ * t) * np.cos(lon0), r * np.cos(lat0 + w_ * t) * np.sin(lon0), r * np.sin(lat0 + w_ * t)], -1)
    vel = np.stack([-r * w_ * np.sin(lat0 + w_ * t) * np.cos(lon0), -r * w_ * np.sin(lat0 + w_ * t) * np.sin(lon0), r * w_ * np.cos(lat0 + w_ * t)], -1)
    epoch = dt.datetime(2021, 1, 1, 6, 57, 0)
    orb = Orbit([epoch + dt.timedelta(seconds=float(x)) for x in t], pos, vel)
    hg, la, lo = np.meshgrid(zs, ys, xs, indexing='ij')
    sec = get_azimuth_time_grid(lo, la, hg, orb, as_datetime64=False)
    osec = O.azimuth_time_grid(orb.time, orb.position, orb.velocity, la, lo, hg)
    assert np.isfinite(sec).all() and np.abs(sec - osec).max() <= 1.001e-3        # both truncate to ms: at most one tick apart
    assert (np.abs(sec - osec) > 0).mean() < 0.01
    grid = get_azimuth_time_grid(lo, la, hg, orb)
    assert grid.dtype == np.dtype('datetime64[ms]') and grid.shape == (nz, ny, nx)
    dates = [dt.datetime(2021, 1, 1, 7), dt.datetime(2021, 1, 1, 6), dt.datetime(2021, 1, 1, 8)]
    pw, tot = combine_weather_cubes_azimuth_time(cubes, cubes, dates, grid)
    grid_s = (grid - np.datetime64(dates[0], 'ms')).astype(np.int64) * 1e-3
    ow = O.inverse_time_weights(grid_s, [0.0, -3600.0, 3600.0])
    np.testing.assert_allclose(pw.read()[0].transpose(2, 0, 1), O.combine_weighted(list(ow), [f[0], f[1], f[2]]), rtol=1e-14)
    assert 0.94 < ow[0].mean() < 0.96                                              # 06:57 + ~0 s: mostly the 07:00 model
    assert np.array_equal(pw.read()[0], tot.read()[0])


def test_gunw_phase_conversion(tmp_path):
    """aria/calcGUNW.py:54-59 (delay -> radians) bit-exact in f32 and f64, host and device buffers; then compute_delays_slc on
    two delay-cube files: later date = reference, GUNW layer names, float32 *Meta coordinates."""
    import torch
    from oracle import raider_oracle as O
    from raider_amd.delay import DelayCube
    from raider_amd.gunw import DIM_NAMES, compute_delays_slc, delays_to_phase
    rng = np.random.default_rng(5)
    lam = 0.05546576
    for dtype in (np.float32, np.float64):
        wet = rng.uniform(0, 0.4, (7, 33, 41)).astype(dtype); hyd = rng.uniform(1.5, 2.6, (7, 33, 41)).astype(dtype)
        wet[0, 0, 0] = np.nan
        pw, ph = delays_to_phase(wet, hyd, lam)
        assert pw.dtype == dtype and ph.dtype == dtype
        assert np.array_equal(pw, O.gunw_phase(wet, lam), equal_nan=True) and np.array_equal(ph, O.gunw_phase(hyd, lam))
        dw, dh = delays_to_phase(torch.from_numpy(wet).cuda(), torch.from_numpy(hyd).cuda(), lam)
        assert np.array_equal(dw.cpu().numpy(), pw, equal_nan=True) and np.array_equal(dh.cpu().numpy(), ph)
    assert pw[1, 1, 1] < 0 and abs(pw[1, 1, 1] / wet[1, 1, 1] + 4 * np.pi / lam) < 1e-9
    with pytest.raises(ValueError):
        delays_to_phase(wet, hyd, 0.0)
    # two dates through files
    z, y, x = np.array([0.0, 500.0, 1000.0]), np.linspace(34.0, 33.0, 5), np.linspace(-118.0, -117.0, 6)
    paths = []
    for i, stamp in enumerate(('20200130T135245', '20200118T135245')):
        ds = DelayCube({'wet': np.full((3, 5, 6), 0.1 * (i + 1)), 'hydro': np.full((3, 5, 6), 2.0 + i), 'z': z, 'y': y, 'x': x},
                       {'model_times_used': f'mt{i}', 'reference_time': f'rt{i}', 'interpolation_method': 'none', '_degrees': True})
        p = tmp_path / f'ERA5_tropo_{stamp}_ray.nc'
        ds.to_netcdf(p); paths.append(p)
    slc = compute_delays_slc(paths, lam)
    k = -4 * np.pi / lam
    assert np.array_equal(slc['reference_troposphereWet'], np.full((3, 5, 6), 0.1) * k)          # 2020-01-30 is the later date
    assert np.array_equal(slc['secondary_troposphereHydrostatic'], np.full((3, 5, 6), 3.0) * k)
    assert all(slc[d].dtype == np.float32 for d in DIM_NAMES) and np.array_equal(slc['latitudeMeta'], y.astype(np.float32))
    assert slc.attrs == {'model': 'ERA5', 'method': 'ray tracing'}
    la = slc.layer_attrs['secondary_troposphereWet']
    assert la['units'] == 'radians' and la['model_times_used'] == 'mt1' and la['scene_center_time'] == 'rt1'
    assert la['description'] == 'Delay due to Wet component of troposphere'
