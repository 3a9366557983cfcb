"""Parity on DATA FILES the reference's own tests hold (tests/golden/ref_files, tests/golden/g12_*.npz; made by
oracle/refharness/gen_ref_file_vectors.py): a processed ERA-5 cube written by the real RAiDER and the reference's own
two-epoch `timeInterp` product."""
import datetime as dt
from pathlib import Path

import numpy as np
import pytest

from oracle import raider_oracle as O

REF_CUBE = Path(__file__).parent / 'golden' / 'ref_files' / 'ERA-5_2019_11_17_T20_51_58_5S_2S_41W_37W.nc'
K1, K2, K3 = 0.776, 0.233, 3.75e3          # models/ecmwf.py:26-28


@pytest.fixture(scope='module')
def cube_file():
    from raider_amd import h5lite
    return h5lite.File(REF_CUBE)


def test_h5lite_reads_the_reference_cube(cube_file):
    """The built-in HDF5 reader on a NetCDF-4 file written by netCDF4/libhdf5 1.14 (dense links in a fractal heap, v2
    object headers, contiguous f32/f64 datasets, variable-length string attributes)."""
    f = cube_file
    assert set(f.keys()) >= {'x', 'y', 'z', 't', 'p', 'e', 'wet', 'hydro', 'wet_total', 'hydro_total', 'proj', 'latitude', 'longitude'}
    assert f.attrs['datetime'] == '2019_11_17T20_51_58' and f.attrs['Conventions'] == 'CF-1.6'
    z, y, x = f['z'].read(), f['y'].read(), f['x'].read()
    assert z.shape == (145,) and z[0] == -500.0 and z[-1] == 80301.65            # models/model_levels.py LEVELS_137_HEIGHTS
    assert np.all(np.diff(z) > 0) and np.all(np.diff(y) > 0) and np.all(np.diff(x) > 0)
    assert -5.01 <= y[0] and y[-1] <= -1.99 and -41.01 <= x[0] and x[-1] <= -36.99     # the bounds in the file name
    assert f['wet'].shape == (145, y.size, x.size) and f['wet'].dtype == np.float32 and f['wet_total'].dtype == np.float64
    wkt = f['proj'].attrs['crs_wkt']
    assert wkt.startswith('GEOGCRS["WGS 84"') and f['proj'].attrs['grid_mapping_name'] == 'latitude_longitude'
    assert f['wet'].attrs['grid_mapping'] == 'proj' and f['t'].attrs['units'] == 'K'
    lat2d = f['latitude'].read()
    assert np.array_equal(lat2d, np.broadcast_to(y[:, None], lat2d.shape))


def test_refractivity_and_ztd_stages_match_the_real_cube(cube_file):
    """weatherModel.py:355-361,389-403 on the reference's own output: wet / hydro recomputed from the file's t, p, e are
    bit-identical, and the ZTD cumulative trapezoids agree to rounding - both through the oracle's restatements."""
    f = cube_file
    t, p, e = (f[k].read().transpose(1, 2, 0) for k in ('t', 'p', 'e'))            # (y, x, z) as in the model object
    wet, hyd = (f[k].read().transpose(1, 2, 0) for k in ('wet', 'hydro'))
    k1, k2, k3 = np.float32(K1), np.float32(K2), np.float32(K3)
    assert np.array_equal(k2 * e / t + k3 * e / t ** 2, wet)
    assert np.array_equal(k1 * p / t, hyd)
    z = f['z'].read()
    tot_w, tot_h = O.ztd_totals(wet, z), O.ztd_totals(hyd, z)
    np.testing.assert_allclose(tot_w, f['wet_total'].read().transpose(1, 2, 0), rtol=1e-14, atol=1e-18)
    np.testing.assert_allclose(tot_h, f['hydro_total'].read().transpose(1, 2, 0), rtol=1e-14, atol=1e-18)


def test_two_epoch_blend_reproduces_the_reference_product(golden):
    """cli/raider.py:817-819,877-888: the 12:00 and 15:00 GMAO cubes blended for 13:52:44 equal the `timeInterp` file the
    reference wrote - bit for bit, f32 fields in f32 arithmetic (numpy < 2 scalar casting), f64 totals in f64."""
    g = golden('g12_gmao_time_interp')
    t1 = dt.datetime.strptime(str(g['t12_datetime']), '%Y_%m_%dT%H_%M_%S')
    t2 = dt.datetime.strptime(str(g['t15_datetime']), '%Y_%m_%dT%H_%M_%S')
    t = dt.datetime.fromisoformat(str(g['query_time']))
    w1, w2 = O.time_weights((t - t1).total_seconds(), 0.0, (t2 - t1).total_seconds())
    assert abs(w1 + w2 - 1) < 1e-15
    for v in ('wet', 'hydro', 'wet_total', 'hydro_total'):
        out = O.blend_cubes(w1, g[f't12_{v}'], w2, g[f't15_{v}'])
        assert out.dtype == g[f'interp_{v}'].dtype and np.array_equal(out, g[f'interp_{v}']), v


def test_delay_cube_netcdf4_roundtrip(tmp_path):
    """DelayCube.to_netcdf writes NetCDF-4 (HDF5) by default, as the reference's `ds.to_netcdf` does (delay.py:329-401 ->
    cli/raider.py:373-398): read back through the built-in reader, and by the delay path as a cube source."""
    from raider_amd import h5lite
    from raider_amd.delay import writeResultsToXarray, DelayCube
    rng = np.random.default_rng(0)
    x, y, z = np.linspace(-118, -117, 5), np.linspace(34, 33, 4), np.array([0.0, 500.0, 1000.0])
    wet, hyd = rng.normal(size=(3, 4, 5)), rng.normal(size=(3, 4, 5))
    ds = writeResultsToXarray(dt.datetime(2020, 1, 1, 12), x, y, z, 4326, wet, hyd, 'ERA5_x.nc', 'slant - raytracing')
    if not isinstance(ds, DelayCube):
        pytest.skip('xarray is installed here: writeResultsToXarray returned a real Dataset')
    path = tmp_path / 'delay.nc'
    ds.to_netcdf(path)
    assert open(path, 'rb').read(4) == b'\x89HDF'
    f = h5lite.File(path)
    assert np.array_equal(f['wet'].read(), wet) and np.array_equal(f['hydro'].read(), hyd) and np.array_equal(f['y'].read(), y)
    assert f['wet'].attrs['grid_mapping'] == 'crs' and f['wet'].attrs['units'] == 'm' and f['hydro'].attrs['description'] == 'hydrostatic slant - raytracing delay'
    assert f.attrs['description'] == 'RAiDER geo cube - slant - raytracing' and f['crs'].attrs['grid_mapping_name'] == 'latitude_longitude'
    assert f['z'].attrs['CLASS'] == 'DIMENSION_SCALE' and f['y'].attrs['units'] == 'degrees_north' and int(f['crs'].read()) == -2147483647
    from raider_amd.delayFcns import _load_fields
    var, get = _load_fields(str(path))
    assert np.array_equal(get('hydro'), hyd)


def test_delay_cube_netcdf3_roundtrip(tmp_path):
    """DelayCube.to_netcdf(format='NETCDF3_64BIT') (the classic-format variant of the delay-cube writer) -> scipy reads it back"""
    from scipy.io import netcdf_file
    from raider_amd.delay import writeResultsToXarray, DelayCube
    rng = np.random.default_rng(0)
    x, y, z = np.linspace(-118, -117, 5), np.linspace(34, 33, 4), np.array([0.0, 500.0, 1000.0])
    wet, hyd = rng.normal(size=(3, 4, 5)), rng.normal(size=(3, 4, 5))
    ds = writeResultsToXarray(dt.datetime(2020, 1, 1, 12), x, y, z, 4326, wet, hyd, 'ERA5_x.nc', 'zenith')
    if not isinstance(ds, DelayCube):
        pytest.skip('xarray is installed here: writeResultsToXarray returned a real Dataset')
    path = tmp_path / 'delay.nc'
    ds.to_netcdf(path, format='NETCDF3_64BIT')
    with netcdf_file(str(path), 'r', mmap=False) as f:
        assert f.variables['wet'].dimensions == ('z', 'y', 'x')
        assert np.array_equal(f.variables['wet'][:], wet) and np.array_equal(f.variables['hydro'][:], hyd)
        assert np.array_equal(f.variables['y'][:], y) and f.variables['wet'].grid_mapping == b'crs'
        assert f.description == b'RAiDER geo cube - zenith' and f.variables['crs'].grid_mapping_name == b'latitude_longitude'
    # and the delay path reads it back as a cube source (getInterpolators(ds, 'ztd') of delay.py:112 takes wet/hydro)
    from raider_amd.delayFcns import _load_fields
    var, get = _load_fields(str(path))
    assert np.array_equal(get('hydro'), hyd)


@pytest.mark.gpu
def test_gpu_blend_and_tropo_delay_from_netcdf4(golden):
    """GPU: the device blend reproduces the reference's timeInterp product bit for bit; tropo_delay takes the NetCDF-4 file
    path directly (no xarray / netCDF4 in this environment) and matches scipy-RGI semantics on the file's total fields."""
    import raider_amd as R
    from raider_amd import h5lite
    from raider_amd.delay import PointsAOI, GridAOI, tropo_delay
    from raider_amd.losreader import Zenith, Raytracing
    g = golden('g12_gmao_time_interp')
    t1 = dt.datetime.strptime(str(g['t12_datetime']), '%Y_%m_%dT%H_%M_%S'); t2 = dt.datetime.strptime(str(g['t15_datetime']), '%Y_%m_%dT%H_%M_%S')
    t = dt.datetime.fromisoformat(str(g['query_time']))
    w1, w2 = O.time_weights((t - t1).total_seconds(), 0.0, (t2 - t1).total_seconds())
    for a, b in (('wet', 'hydro'), ('wet_total', 'hydro_total')):
        c1 = R.Cube(g['y'], g['x'], g['z'], g[f't12_{a}'], g[f't12_{b}'], order='zyx')
        c2 = R.Cube(g['y'], g['x'], g['z'], g[f't15_{a}'], g[f't15_{b}'], order='zyx')
        w, h = c1.blend(w1, c2, w2).read()
        assert w.dtype == g[f'interp_{a}'].dtype
        assert np.array_equal(w.transpose(2, 0, 1), g[f'interp_{a}']) and np.array_equal(h.transpose(2, 0, 1), g[f'interp_{b}'])
    # ---- tropo_delay straight from the reference's NetCDF-4 cube
    f = h5lite.File(REF_CUBE)
    xs, ys, zs = f['x'].read().astype(np.float64), f['y'].read().astype(np.float64), f['z'].read()
    rng = np.random.default_rng(4)
    lats = rng.uniform(ys[1], ys[-2], 300); lons = rng.uniform(xs[1], xs[-2], 300); hgts = rng.uniform(0, 3000, 300)
    xpts, ypts = np.linspace(xs[1], xs[-2], 40), np.linspace(ys[-2], ys[1], 30)
    hl = [0.0, 500.0, 1500.0, 3500.0]
    wz, hz = tropo_delay(dt.datetime(2019, 11, 17, 20, 51, 58), str(REF_CUBE), PointsAOI(lats, lons, hgts, xpts, ypts), Zenith(), hl)
    ip = [O.RGI((ys, xs, zs), f[k].read().transpose(1, 2, 0)) for k in ('wet_total', 'hydro_total')]
    cw, ch = O.build_cube(xpts, ypts, np.array(hl), ip)
    ow, oh = O.points_from_cube(lats, lons, hgts, xpts, ypts, np.array(hl), cw, ch)
    np.testing.assert_allclose(wz, ow, rtol=0, atol=1e-13); np.testing.assert_allclose(hz, oh, rtol=0, atol=1e-13)
    assert 1.6 < hz.mean() < 2.4                                   # real ERA-5 hydrostatic ZTDs (m) between 0 and 3 km
    # ray tracing through the real 145-level cube, against the oracle
    aoi = GridAOI(np.linspace(xs[3], xs[-4], 12), np.linspace(ys[-4], ys[3], 10))
    ds, _ = tropo_delay(dt.datetime(2019, 11, 17, 20, 51, 58), str(REF_CUBE), aoi, Raytracing(inc=38.0, heading=-165.0), [0.0, 2000.0])
    pw = [O.RGI((ys, xs, zs), f[k].read().transpose(1, 2, 0)) for k in ('wet', 'hydro')]
    look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(np.full(yy.shape, 38.0), np.full(yy.shape, -165.0), llh[1], llh[0], llh[2])
    rw, rh = O.build_cube_ray(aoi.xpts, aoi.ypts, np.array([0.0, 2000.0]), look, pw, MAX_TROPO_HEIGHT=float(zs.max() - 1))
    np.testing.assert_allclose(np.asarray(ds['wet'][:]), rw, rtol=0, atol=5e-9)
    np.testing.assert_allclose(np.asarray(ds['hydro'][:]), rh, rtol=0, atol=5e-9)


def _aoi_grid(S, N, W, E, ll_res=0.25, cube_spacing_in_m=2000.0, digits=2):
    """The output grid calcDelays builds (cli/raider.py:257-260): AOI.add_buffer(model.getLLRes()) = 1.5 model cells, clipped
    outwards to multiples of the OUTPUT spacing (the default runtime_group.cube_spacing_in_m = 2000 m -> 0.02 deg,
    constants.py:22, llreader.py:76-89), rounded to 2 digits; then AOI.set_output_xygrid(4326) (llreader.py:177-192)."""
    sp = cube_spacing_in_m / 1e5
    buf = 1.5 * ll_res
    S, N, W, E = max(S - buf, -90), min(N + buf, 90), W - buf, E + buf
    S, N, W, E = (np.floor(S / sp) * sp, np.ceil(N / sp) * sp, np.floor(W / sp) * sp, np.ceil(E / sp) * sp)
    S, N, W, E = (np.round(a, digits) for a in (S, N, W, E))
    return np.arange(W, E + sp, sp), np.arange(N, S - sp, -sp)


REF_CUBE_0130 = Path(__file__).parent / 'golden' / 'ref_files' / 'ERA-5_2020_01_30_T13_52_45_32N_35N_120W_115W.nc'
WHEN_0130 = dt.datetime(2020, 1, 30, 13, 52, 45)


@pytest.mark.gpu
def test_reference_gnss_intersect_end_to_end(golden):
    """END TO END against the reference's own test suite: test/test_intersect.py::test_gnss_intersect runs `raider.py` on the
    processed ERA-5 cube of 2020-01-30T13:52:45 for scenario_6/stations.csv and expects a total zenith delay of 2.34514 m
    (4 decimals) at station TORP.  Same cube file (NetCDF-4, read by h5lite), same stations, same AOI buffering / output grid,
    tropo_delay on the GPU - and golden g13 = what the reference's own, unmodified tropo_delay returns for these inputs in the
    build container (to 1e-14 m)."""
    import csv
    from raider_amd.delay import PointsAOI, tropo_delay
    from raider_amd.losreader import Zenith
    g = golden('g13_gnss_intersect')
    with open(Path(__file__).parent / 'golden' / 'ref_files' / 'scenario_6_stations.csv') as fh:
        rows = list(csv.DictReader(fh))
    lats = np.array([float(r['Lat']) for r in rows]); lons = np.array([float(r['Lon']) for r in rows]); hgts = np.array([float(r['Hgt_m']) for r in rows])
    assert np.array_equal(lats, g['lats']) and [r['ID'] for r in rows] == list(g['ids'])
    xa, ya = _aoi_grid(lats.min(), lats.max(), lons.min(), lons.max())      # llreader.bounds_from_csv; ERA-5 getLLRes() = 0.25
    assert np.array_equal(xa, g['x_aoi']) and np.array_equal(ya, g['y_aoi'])
    wet, hyd = tropo_delay(WHEN_0130, str(REF_CUBE_0130), PointsAOI(lats, lons, hgts, xa, ya), Zenith(), height_levels=None, out_proj=4326, zref=None)
    np.testing.assert_allclose(wet, g['wet_aoi'], rtol=0, atol=1e-14); np.testing.assert_allclose(hyd, g['hydro_aoi'], rtol=0, atol=1e-14)
    torp = list(g['ids']).index('TORP')
    np.testing.assert_almost_equal(wet[torp] + hyd[torp], 2.34514, decimal=4)          # test/test_intersect.py:106,113
    wet_m, hyd_m = tropo_delay(WHEN_0130, str(REF_CUBE_0130), PointsAOI(lats, lons, hgts, g['x_model'], g['y_model']), Zenith())
    np.testing.assert_allclose(wet_m, g['wet_model'], rtol=0, atol=1e-14); np.testing.assert_allclose(hyd_m, g['hydro_model'], rtol=0, atol=1e-14)


@pytest.mark.gpu
def test_reference_slant_proj_end_to_end(golden):
    """test/test_slant.py::test_slant_proj: bounding box [33, 34, -118.25, -116.75], heights 0/100/500/1000 m, `ray_trace: False`
    with an orbit file - i.e. a PROJECTED line of sight on a cube AOI, which the reference turns into zenith delays (SURVEY
    0.8) - on the same ERA-5 cube; the test reads the node nearest (33.4, -117.8, 0) and expects 2.333865144 m to 7 decimals."""
    from raider_amd.delay import GridAOI, tropo_delay
    from raider_amd.losreader import Conventional
    g = golden('g13_gnss_intersect')
    xp, yp = _aoi_grid(33, 34, -118.25, -116.75)
    assert np.array_equal(xp, g['x_proj']) and np.array_equal(yp, g['y_proj'])
    ds, none = tropo_delay(WHEN_0130, str(REF_CUBE_0130), GridAOI(xp, yp), Conventional('orbit_file_never_opened_for_a_cube_aoi.EOF'),
                           [0, 100, 500, 1000], 4326, None)
    assert none is None
    wet, hyd = np.asarray(ds['wet'][:]), np.asarray(ds['hydro'][:])
    np.testing.assert_allclose(wet, g['wet_proj'], rtol=0, atol=1e-14); np.testing.assert_allclose(hyd, g['hydro_proj'], rtol=0, atol=1e-14)
    iy, ix = np.abs(yp - 33.4).argmin(), np.abs(xp + 117.8).argmin()
    np.testing.assert_almost_equal(2.333865144, wet[0, iy, ix] + hyd[0, iy, ix])       # test/test_slant.py:49,57 (decimal=7)


@pytest.mark.gpu
def test_reference_raytracing_on_the_real_cube(golden):
    """The reference's own _build_cube_ray (golden g13, generated by running the unmodified reference on the real 145-level
    ERA-5 cube of its test suite, rays to 80.3 km at 30-44 deg incidence) vs the GPU ray tracer through tropo_delay on the
    same file: same NaNs (lateral exits at the cube edge), delays within 1e-9 m."""
    from raider_amd.delay import GridAOI, tropo_delay
    from raider_amd.losreader import Raytracing
    g = golden('g13_gnss_intersect')
    cube = Path(__file__).parent / 'golden' / 'ref_files' / 'ERA-5_2020_01_30_T13_52_45_32N_35N_120W_115W.nc'
    ds, none = tropo_delay(dt.datetime(2020, 1, 30, 13, 52, 45), str(cube), GridAOI(g['x_ray'], g['y_ray']),
                           Raytracing(inc=g['inc_ray'], heading=-167.9), list(g['z_ray']), 4326, None)
    assert none is None
    wet, hyd = np.asarray(ds['wet'][:]), np.asarray(ds['hydro'][:])
    assert np.array_equal(np.isnan(hyd), np.isnan(g['hydro_ray'])) and np.isnan(g['hydro_ray']).sum() == 4
    np.testing.assert_allclose(wet, g['wet_ray'], rtol=0, atol=1e-9, equal_nan=True)
    np.testing.assert_allclose(hyd, g['hydro_ray'], rtol=0, atol=1e-9, equal_nan=True)


# ---- raw ERA-5 model-level files -> processed cubes: the whole producer chain against what the real RAiDER wrote ---------------
RAW_PAIRS = [('ERA-5_2019_11_17_T20_51_58.nc', 'ERA-5_2019_11_17_T20_51_58_5S_2S_41W_37W.nc'),
             ('ERA-5_2022_08_29_T17_00_01.nc', 'ERA-5_2022_08_29_T17_00_01_69N_73N_159W_152W.nc')]
# what float32 round-off allows: the reference computes the level heights in float32 (dlogP = log(P1) - log(P0) loses 3-4 digits), so
# a last-bit difference in log/exp moves a height by centimetres and the resampled fields by ~1e-6 relative; e and the wet
# refractivity are differences of large numbers near the tropopause and are compared on the scale of their maximum
_CHAIN_TOL = dict(t=dict(rtol=1e-6), p=dict(rtol=2e-5), hydro=dict(rtol=2e-5), hydro_total=dict(rtol=2e-5),
                  e=dict(rtol=0, atol_of_max=2e-6), wet=dict(rtol=0, atol_of_max=2e-6), wet_total=dict(rtol=0, atol_of_max=2e-6))


def _compare_with_processed(res, proc_path, tol_scale=1.0):
    from raider_amd import h5lite
    g = h5lite.File(proc_path)
    out = {}
    for k, tol in _CHAIN_TOL.items():
        want = g[k].read().astype(np.float64)
        got = np.asarray(res[k], dtype=np.float64)
        assert got.shape == want.shape and np.array_equal(np.isnan(got), np.isnan(want)), k
        m = np.isfinite(want) & (np.abs(want) < 1e15)            # (t holds the 1e16 fill of _checkForNans below the surface)
        d = np.abs(got - want)[m]
        if 'atol_of_max' in tol:
            lim = tol['atol_of_max'] * tol_scale * np.abs(want[m]).max()
            assert d.max() <= lim, (k, d.max(), lim)
        else:
            nz = want[m] != 0                                    # (p is 0 where _checkForNans filled below the surface)
            assert np.all(got[m][~nz] == 0), k
            rel = d[nz] / np.abs(want[m][nz])
            assert rel.max() <= tol['rtol'] * tol_scale, (k, rel.max())
        out[k] = d.max()
    return out


@pytest.mark.parametrize('raw,proc', RAW_PAIRS)
def test_oracle_producer_chain_reproduces_the_real_processed_cubes(raw, proc):
    """Raw ERA-5 model-level file (test/weather_files/, packed int16 on 137 hybrid levels) -> hybrid pressures + geometric heights
    (utilFcns.calcgeoh, geo_to_ht) -> e -> 145 uniform levels -> fill -> refractivities -> ZTDs, restated in the oracle, against
    the processed cube the real RAiDER wrote from the same raw file: same grid, same NaNs, t to 1e-6, p / hydro to 2e-5
    relative, ZTDs to 2e-7 m."""
    from raider_amd.weather import ecmwf_l137
    d = Path(__file__).parent / 'golden' / 'ref_files'
    tab = ecmwf_l137()
    r = O.read_ecmwf_model_level_file(d / raw)
    p, h = O.ecmwf_model_levels(r['z'], r['lnsp'], r['t'], r['q'], r['lats'], tab['a'], tab['b'])
    assert p.dtype == np.float32 and h.dtype == np.float32 and np.all(np.diff(h, axis=2) > 0)
    up = lambda v: np.flip(v.transpose(1, 2, 0), axis=2).astype(np.float64)
    res = O.cube_from_model_levels(h.astype(np.float64), p.astype(np.float64), up(r['t']), up(r['q']), 'q', np.flipud(tab['level_heights']))
    from raider_amd import h5lite
    g = h5lite.File(d / proc)
    assert np.array_equal(g['y'].read(), r['lats']) and np.array_equal(g['x'].read(), r['lons']) and np.array_equal(g['z'].read(), res['zs'])
    worst = _compare_with_processed({k: res[k].transpose(2, 0, 1) for k in _CHAIN_TOL}, d / proc)
    assert worst['hydro_total'] < 2e-7 and worst['wet_total'] < 5e-8          # metres of zenith delay
    # the float64 evaluation of the same formulas is NOT what the reference computed: its float32 heights sit metres away
    p64, h64 = O.ecmwf_model_levels(r['z'], r['lnsp'], r['t'], r['q'], r['lats'], tab['a'], tab['b'], dtype=np.float64)
    assert 0.5 < np.abs(h64 - h).max() < 5.0 and np.abs(p64 / np.maximum(p, 1e-30) - 1)[p > 0].max() < 3e-7


@pytest.mark.gpu
@pytest.mark.parametrize('raw,proc', RAW_PAIRS)
def test_gpu_producer_chain_from_raw_era5_file(raw, proc):
    """The same chain on the GPU (raider_amd.weather.load_ecmwf_model_levels: rdr_ecmwf_model_levels + rdr_cubes_from_model_levels).
    The device evaluates the level heights in float64 (the reference's float32 evaluation is ill-conditioned and platform
    dependent, see the CPU test): pinned on the oracle's float64 restatement (1e-7 m), the rest of the chain on the oracle's
    producer fed with those heights; against the real processed cube the difference is then the REFERENCE's own float32 round-off:
    up to 0.3 K / 0.4 % in the resampled state, <= 7e-4 m in the zenith delays - bounded here so that it cannot grow unnoticed."""
    from raider_amd import h5lite
    from raider_amd.weather import ecmwf_l137, ecmwf_model_levels, load_ecmwf_model_levels, read_ecmwf_model_level_file
    d = Path(__file__).parent / 'golden' / 'ref_files'
    tab = ecmwf_l137()
    r = read_ecmwf_model_level_file(d / raw)
    ro = O.read_ecmwf_model_level_file(d / raw)
    assert all(np.array_equal(r[k], ro[k], equal_nan=True) and r[k].dtype == ro[k].dtype for k in ro)
    p, zs = ecmwf_model_levels(r['z'], r['lnsp'], r['t'], r['q'], r['lats'])
    op, oh = O.ecmwf_model_levels(ro['z'], ro['lnsp'], ro['t'], ro['q'], ro['lats'], tab['a'], tab['b'], dtype=np.float64)
    np.testing.assert_allclose(p, op, rtol=1e-14, atol=0)
    assert np.abs(zs - oh).max() < 1e-7 and np.all(np.diff(zs, axis=2) > 0)
    import torch
    dev = torch.device('cuda:0')
    pd_, zd = ecmwf_model_levels(*(torch.from_numpy(r[k]).to(dev) for k in ('z', 'lnsp', 't', 'q')), r['lats'])
    assert np.array_equal(pd_.cpu().numpy(), p) and np.array_equal(zd.cpu().numpy(), zs)
    with pytest.raises(ValueError, match='these three numbers should be equal'):
        ecmwf_model_levels(r['z'], r['lnsp'], r['t'], r['q'], r['lats'], a=tab['a'][:-1], b=tab['b'])
    # the whole chain vs the oracle's producer on the same (float64) heights
    up = lambda v: np.flip(v.transpose(1, 2, 0), axis=2).astype(np.float64)
    want = O.cube_from_model_levels(oh, op, up(ro['t']), up(ro['q']), 'q', np.flipud(tab['level_heights']))
    model = load_ecmwf_model_levels(d / raw, return_state=True)
    wet, hyd = model.pointwise.read(); wt, ht_ = model.total.read()               # (y, x, z)
    got = dict(t=np.asarray(model.t), p=np.asarray(model.p), e=np.asarray(model.e), wet=wet, hydro=hyd, wet_total=wt, hydro_total=ht_)
    assert np.array_equal(model.zs, want['zs'])
    for k in ('t', 'p', 'hydro'):
        np.testing.assert_allclose(got[k], want[k], rtol=3e-7, atol=0, equal_nan=True)          # float32 state: within 2 ulp
    for k in ('e', 'wet'):
        assert np.nanmax(np.abs(got[k].astype(np.float64) - want[k])) <= 3e-7 * np.nanmax(np.abs(want[k]))
    assert np.nanmax(np.abs(got['hydro_total'] - want['hydro_total'])) < 1e-8 and np.nanmax(np.abs(got['wet_total'] - want['wet_total'])) < 1e-8
    # distance to the cube the real RAiDER wrote = the reference's float32 round-off in the level heights
    g = h5lite.File(d / proc)
    dist = {k: float(np.nanmax(np.abs(got[k].transpose(2, 0, 1).astype(np.float64) - g[k].read().astype(np.float64))[np.abs(g[k].read()) < 1e15])) for k in got}
    assert all(np.array_equal(np.isnan(got[k].transpose(2, 0, 1)), np.isnan(g[k].read())) for k in got)
    assert dist['t'] < 0.5 and dist['p'] < 500.0 and dist['hydro_total'] < 7e-4 and dist['wet_total'] < 7e-4, dist


@pytest.mark.gpu
def test_processed_model_file_roundtrip(tmp_path):
    """ProcessedModel.to_netcdf writes the processed-cube file of WeatherModel.write - NetCDF-4 by default, NetCDF-3 on request: read
    back by path either gives the same delays as the device-resident model; the NetCDF-4 file has the reference's own file
    (tests/golden/ref_files, written by the real RAiDER through xarray) as its template: same variables, element types, shapes,
    per-variable attribute names and values."""
    from scipy.io import netcdf_file
    from raider_amd import h5lite
    from raider_amd.delay import GridAOI, tropo_delay
    from raider_amd.losreader import Raytracing, Zenith
    from raider_amd.weather import load_ecmwf_model_levels
    d = Path(__file__).parent / 'golden' / 'ref_files'
    model = load_ecmwf_model_levels(d / 'ERA-5_2019_11_17_T20_51_58.nc', return_state=True)
    when = dt.datetime(2019, 11, 17, 20, 51, 58)
    path4 = model.to_netcdf(tmp_path / 'ERA-5_2019_11_17_T20_51_58_5S_2S_41W_37W.nc', time=when)
    mine, ref = h5lite.File(path4), h5lite.File(d / 'ERA-5_2019_11_17_T20_51_58_5S_2S_41W_37W.nc')
    assert set(mine.keys()) == set(ref.keys()) - {'datetime'}                      # (the scalar time coordinate xarray adds is not written)
    for k in mine.keys():
        a, b = mine[k], ref[k]
        assert a.shape == b.shape and a.dtype == b.dtype, k
        drop = {'_Netcdf4Dimid', 'coordinates'} if k in ('proj',) else {'_Netcdf4Dimid'}
        assert set(a.attrs) - {'_Netcdf4Dimid'} >= set(b.attrs) - drop - {'REFERENCE_LIST', 'DIMENSION_LIST'}, (k, set(a.attrs), set(b.attrs))
        for an in ('units', 'standard_name', 'grid_mapping', 'CLASS', 'NAME', 'grid_mapping_name', 'crs_wkt', 'semi_major_axis', 'inverse_flattening'):
            if an in b.attrs:
                va, vb = a.attrs[an], b.attrs[an]
                assert (va == vb) if isinstance(vb, str) else np.allclose(va, vb), (k, an, va, vb)
    for k in ('x', 'y', 'z', 'latitude', 'longitude'):
        assert np.array_equal(mine[k].read(), ref[k].read()), k
    assert {'Conventions', 'title', 'datetime', 'date_created', '_NCProperties'} <= set(mine.attrs) and mine.attrs['datetime'] == ref.attrs['datetime']
    path = model.to_netcdf(tmp_path / 'classic.nc', time=when, format='NETCDF3_64BIT')
    with netcdf_file(path, 'r', mmap=False) as f:
        assert set(f.variables) >= {'x', 'y', 'z', 't', 'p', 'e', 'wet', 'hydro', 'wet_total', 'hydro_total', 'latitude', 'longitude', 'proj'}
        assert f.variables['wet'].dimensions == ('z', 'y', 'x') and f.variables['wet'].data.dtype.itemsize == 4 and f.variables['wet_total'].data.dtype.itemsize == 8
        assert f.variables['hydro'].standard_name == b'hydrostatic_refractivity' and f.variables['t'].units == b'K' and f.variables['p'].grid_mapping == b'proj'
        assert f.datetime == b'2019_11_17T20_51_58' and b'4326' in f.variables['proj'].crs_wkt
    x, y = model['x'], model['y']
    aoi = GridAOI(x[2:-2], y[2:-2][::-1])
    for los in (Zenith(), Raytracing(inc=35.0, heading=-167.9)):
        a, _ = tropo_delay(when, model, aoi, los, [0.0, 1200.0], 4326, None)
        for pth in (path, path4):
            b, _ = tropo_delay(when, pth, aoi, los, [0.0, 1200.0], 4326, None)
            assert np.array_equal(np.asarray(a['hydro'][:]), np.asarray(b['hydro'][:]), equal_nan=True)
            assert np.array_equal(np.asarray(a['wet'][:]), np.asarray(b['wet'][:]), equal_nan=True) and np.isfinite(np.asarray(a['hydro'][:])).any()


HRRR_ZTD_CUBE = Path(__file__).parent / 'golden' / 'ref_files' / 'HRRR_tropo_20200101T120000_ztd.nc'


@pytest.mark.gpu
def test_getInterpolators_on_the_references_own_delay_cube(caplog):
    """test/test_delayFcns.py:30-45 replayed: the delay-cube file the reference's tests load (scenario_1/golden_data, written by the
    real RAiDER: f64 wet / hydro on (z, y, x), DESCENDING y, a scalar `crs` variable with the EPSG:4326 grid mapping) goes through
    getInterpolators; the device interpolators then agree with scipy's RegularGridInterpolator on the same arrays - the second
    stage of tropo_delay's point branch (delay.py:110-121) - and a NaN cell makes getInterpolators log 'Weather model contains NaNs!'."""
    import logging
    from scipy.interpolate import RegularGridInterpolator
    from raider_amd.delayFcns import _read_cube_file, getInterpolators
    var = _read_cube_file(HRRR_ZTD_CUBE)
    x, y, z = (np.array(var[k][:]) for k in 'xyz')
    wet, hyd = np.array(var['wet'][:]), np.array(var['hydro'][:])
    assert wet.shape == (5, 102, 101) and wet.dtype == np.float64 and y[0] > y[-1] and list(z) == [0.0, 50.0, 100.0, 500.0, 1000.0]
    assert var['wet'].attrs['grid_mapping'] == 'crs' and var['crs'].attrs['grid_mapping_name'] == 'latitude_longitude' and int(var['crs'][:]) == -2147483647
    ifw, ifh = getInterpolators(str(HRRR_ZTD_CUBE), kind='pointwise')
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(y.min() - 0.05, y.max() + 0.05, 4000), rng.uniform(x.min() - 0.05, x.max() + 0.05, 4000), rng.uniform(-20, 1020, 4000)], -1)
    pts[:50, 0] = y[rng.integers(0, y.size, 50)]; pts[50:100, 1] = x[rng.integers(0, x.size, 50)]; pts[100:150, 2] = z[rng.integers(0, z.size, 50)]   # exactly on nodes
    for f, arr in ((ifw, wet), (ifh, hyd)):
        ref = RegularGridInterpolator((y, x, z), arr.transpose(1, 2, 0), bounds_error=False, fill_value=np.nan)(pts)
        got = f(pts)
        assert np.array_equal(np.isnan(got), np.isnan(ref)) and np.isnan(ref).sum() > 100 and np.isfinite(ref).sum() > 2000
        np.testing.assert_allclose(got, ref, rtol=0, atol=4e-15, equal_nan=True)
    # the nearest node to (36.84 N, 91.84 W, 0 m): read straight from the file and through the device
    i, j = np.abs(x + 91.84).argmin(), np.abs(y - 36.84).argmin()
    assert ifh(np.array([y[j], x[i], 0.0])) == hyd[0, j, i] and ifw(np.array([y[j], x[i], 0.0])) == wet[0, j, i]
    # test_getInterpolators_2: a NaN in the cube is reported, not fatal
    ds = {k: np.array(var[k][:]) for k in ('x', 'y', 'z', 'wet', 'hydro')}
    ds['hydro'][0, 0, 0] = np.nan
    with caplog.at_level(logging.CRITICAL):
        getInterpolators(ds, kind='pointwise')
    assert 'Weather model contains NaNs!' in caplog.text
