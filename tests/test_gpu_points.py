"""GPU: the device-resident point branch of tropo_delay (delay.py:96-128; round 4) and the C-ABI entries behind it -
rdr_build_cube_to_cube, rdr_raytrace_slices_to_cube, rdr_interp3_project, rdr_project_cosinc / rdr_project_divide - against
(a) the reference's golden vectors (g6 `proj_last`, g8), (b) the reference's own SEQUENCE of calls run through the public pieces
(cube down, Dataset, getInterpolators(ds), two gathers, los()) - bit for bit - and (c) NumPy arithmetic.
Tolerances: bit-exact wherever the device runs the same IEEE operations as the host (gathers, divisions); 2 ulp where the device's
cos() stands in for libm's (delay / cosd(inc))."""
import ctypes as C
import datetime as dt
import logging

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import raider_oracle as O

WHEN = dt.datetime(2020, 1, 1)


@pytest.fixture(scope='module')
def c1():
    return O.synthetic_cube(50, 50, 40, seed=0)


def _wm(c):
    return dict(x=c['xs'], y=c['ys'], z=c['zs'], wet=c['wet'], hydro=c['hydro'], wet_total=c['wet_total'], hydro_total=c['hydro_total'])


def _ulp_close(a, b, ulps=2):
    a, b = np.asarray(a), np.asarray(b)
    nan = np.isnan(a)
    assert np.array_equal(nan, np.isnan(b))
    return bool(np.all(np.abs(a[~nan] - b[~nan]) <= ulps * np.spacing(np.abs(b[~nan]))))


# ---- A10: Conventional.__call__ on the device ---------------------------------------------------------------------------------
def test_project_cosinc_against_golden_g6(golden):
    """rdr_project_cosinc / rdr_project_divide called as a foreign binding would, against golden g6 `proj_last` - what the reference's
    Conventional.__call__ tail, delays / inc_hd_to_enu(inc, hd)[..., -1] (losreader.py:130-133), returned in the build container."""
    from raider_amd import _lib as L
    from raider_amd._lib import Context, ptr
    g = golden('g6_los')
    ctx = Context.default()
    inc = np.ascontiguousarray(g['inc'], dtype=np.float64).ravel()
    d = np.ascontiguousarray(g['delays'], dtype=np.float64).ravel()
    want = np.asarray(g['proj_last']).ravel()
    w, h = d.copy(), (2.0 * d).copy()
    assert ctx.lib.rdr_project_cosinc(ctx.handle, ptr(w), ptr(h), ptr(inc), d.size, L.RDR_HOST) == 0
    assert _ulp_close(w, want) and _ulp_close(h, 2.0 * want)
    w1 = d.copy()                                               # the reference projects wet and hydro in two calls: one field at a time
    assert ctx.lib.rdr_project_cosinc(ctx.handle, ptr(w1), None, ptr(inc), d.size, L.RDR_HOST) == 0
    assert np.array_equal(w1, w)
    h1 = d.copy()
    assert ctx.lib.rdr_project_cosinc(ctx.handle, None, ptr(h1), ptr(inc), d.size, L.RDR_HOST) == 0
    assert np.array_equal(h1, w)
    assert ctx.lib.rdr_project_cosinc(ctx.handle, None, None, ptr(inc), d.size, L.RDR_HOST) == L.RDR_ERR_INVALID
    # the divisor form (LOS_enu from an orbit file = cos(look angle), losreader.py:122-131): an IEEE division, bit-exact
    up = np.cos(np.radians(inc))
    w2 = d.copy()
    assert ctx.lib.rdr_project_divide(ctx.handle, ptr(w2), None, ptr(up), d.size, L.RDR_HOST) == 0
    assert np.array_equal(w2, d / up)
    # ... and on device arrays, in place
    import torch
    tw = torch.from_numpy(d.copy()).cuda(); ti = torch.from_numpy(inc).cuda()
    ctx.adopt_torch_stream(tw)
    assert ctx.lib.rdr_project_cosinc(ctx.handle, ptr(tw), None, ptr(ti), d.size, L.RDR_DEVICE) == 0
    torch.cuda.synchronize()
    ctx.set_stream(-1)
    assert np.array_equal(tw.cpu().numpy(), w)


def test_conventional_call_runs_on_the_device(golden):
    """Conventional.__call__ = the device kernel (round 3: host NumPy).  Shapes follow the reference's rule: a delay array of the
    shape of LOS_enu is divided by it, anything else by its last component, with NumPy broadcasting."""
    from raider_amd.losreader import Conventional, inc_hd_to_enu
    g = golden('g6_los')
    conv = Conventional(inc=g['inc'], heading=g['hd'])
    conv.setPoints(g['lat'], g['lon'], 0 * g['lat'])
    d = np.array(g['delays'])
    out = conv(d)
    assert out is not d and np.array_equal(d, g['delays'])                 # the caller's array is left alone
    assert _ulp_close(out, g['proj_last'])
    rng = np.random.default_rng(4)
    inc = rng.uniform(20, 50, (7, 9)); hd = rng.uniform(-180, 180, (7, 9))
    c2 = Conventional(inc=inc, heading=hd); c2.setPoints(inc, inc, inc)
    z = rng.uniform(2, 3, (7, 9))
    assert _ulp_close(c2(z), z / inc_hd_to_enu(inc, hd)[..., -1])
    z3 = rng.uniform(2, 3, (4, 7, 9))                                       # a stack of delay maps against one raster: broadcast
    assert _ulp_close(c2(z3), z3 / inc_hd_to_enu(inc, hd)[..., -1])
    e3 = rng.uniform(2, 3, (7, 9, 3))                                       # losreader.py:130-131: same shape as LOS_enu
    assert np.array_equal(c2(e3), e3 / inc_hd_to_enu(inc, hd))
    c0 = Conventional(inc=39.0, heading=-167.9); c0.setPoints(inc, inc, inc)
    assert _ulp_close(c0(z), z / np.cos(np.radians(39.0)))
    with pytest.raises(ValueError, match='Incidence angle cannot be less than 0'):
        Conventional(inc=np.array([10.0, -1.0]), heading=0.0)


# ---- the point branch ---------------------------------------------------------------------------------------------------------
def _host_sequence(wm, aoi, los, heights, out_proj=4326, zref=None):
    """delay.py:96-128 literally, through the public pieces: cube down -> Dataset -> getInterpolators(ds, 'ztd') -> the two
    interpolator calls -> los().  What round 3's tropo_delay did; the device route must give the same bits."""
    from raider_amd.delay import _get_delays_on_cube, transformPoints
    from raider_amd.delayFcns import getInterpolators
    zs = np.asarray(wm['z'])
    toa = zs.max() - 1
    zref = toa if zref is None else min(zref, toa)
    ds = _get_delays_on_cube(WHEN, wm, 4326, aoi, heights, los, out_proj, zref)
    ifw, ifh = getInterpolators(ds, 'ztd')
    lats, lons = aoi.readLL(); hg = aoi.readZ()
    pn = transformPoints(lats, lons, hg, 4326, out_proj)
    return ifw(pn), ifh(pn)


def test_point_branch_device_route_equals_the_reference_sequence(golden, c1, caplog):
    from raider_amd import delay as D
    from raider_amd.delay import PointsAOI, tropo_delay
    from raider_amd.losreader import Conventional, Raytracing, Zenith, inc_hd_to_enu
    g = golden('g8_points')
    hl = list(g['height_levels'])
    wm = _wm(c1)
    lats, lons, hgts = g['lats'], g['lons'], g['hgts']
    calls = []
    real = D._point_branch_on_device
    D._point_branch_on_device = lambda *a, **k: (calls.append(1), real(*a, **k))[1]
    try:
        # zenith: golden g8 (the reference itself) and the host sequence, bit for bit
        wz, hz = tropo_delay(WHEN, wm, PointsAOI(lats, lons, hgts, g['xpts'], g['ypts']), Zenith(), hl, 4326, None)
        sw, sh = _host_sequence(wm, PointsAOI(lats, lons, hgts, g['xpts'], g['ypts']), Zenith(), hl)
        assert np.array_equal(wz, sw, equal_nan=True) and np.array_equal(hz, sh, equal_nan=True)
        np.testing.assert_allclose(wz, g['wet_zen'], rtol=0, atol=1e-13)
        # projected: the same gather, then / cosd(inc) in the same launch
        inc = np.random.default_rng(1).uniform(25, 45, lats.shape); hd = np.full(lats.shape, -167.9)
        wp, hp = tropo_delay(WHEN, wm, PointsAOI(lats, lons, hgts, g['xpts'], g['ypts']), Conventional(inc=inc, heading=hd), hl, 4326, None)
        up = inc_hd_to_enu(inc, hd)[..., -1]
        assert _ulp_close(wp, sw / up) and _ulp_close(hp, sh / up)
        ws, hs = tropo_delay(WHEN, wm, PointsAOI(lats, lons, hgts, g['xpts'], g['ypts']), Conventional(inc=35.0, heading=0.0), hl, 4326, None)
        assert _ulp_close(hs, sh / np.cos(np.radians(35.0)))
        # ray traced: the intermediate cube comes from rdr_raytrace_slices_to_cube
        n = 200
        los = Raytracing(inc=39.0, heading=-167.9)
        wr, hr = tropo_delay(WHEN, wm, PointsAOI(lats[:n], lons[:n], hgts[:n], g['xpts_ray'], g['ypts_ray']), los, hl, 4326, None)
        rw, rh = _host_sequence(wm, PointsAOI(lats[:n], lons[:n], hgts[:n], g['xpts_ray'], g['ypts_ray']), los, hl)
        assert np.array_equal(wr, rw, equal_nan=True) and np.array_equal(hr, rh, equal_nan=True)
        np.testing.assert_allclose(hr, g['hydro_ray'], rtol=0, atol=1e-9)
        assert len(calls) == 4
        # points outside the intermediate grid / above its top height: NaN, as scipy's fill value
        la = np.array([33.0, 10.0, 33.0]); lo = np.array([-117.0, -117.0, -117.0]); hh = np.array([100.0, 100.0, 1e6])
        wo, ho = tropo_delay(WHEN, wm, PointsAOI(la, lo, hh, g['xpts'], g['ypts']), Zenith(), hl, 4326, None)
        assert np.isfinite(wo[0]) and np.isnan(wo[1]) and np.isnan(ho[2])
        # an intermediate grid that leaves the weather model: NaNs in the cube, reported as the reference reports them (delay.py:187-188)
        caplog.clear()
        with caplog.at_level(logging.CRITICAL):
            xo = np.linspace(c1['xs'].min() - 1.0, c1['xs'].max(), 30)
            tropo_delay(WHEN, wm, PointsAOI(lats, lons, hgts, xo, g['ypts']), Zenith(), hl, 4326, None)
        assert any('missing delay values' in r.getMessage() for r in caplog.records)
        assert len(calls) == 6
        # a job the device route leaves to the reference's sequence: ONE height level is no grid along z - scipy's error, as before
        with pytest.raises(ValueError, match='strictly ascending or descending'):
            tropo_delay(WHEN, wm, PointsAOI(lats, lons, np.full(lats.shape, hl[0]), g['xpts'], g['ypts']), Zenith(), hl[:1], 4326, None)
    finally:
        D._point_branch_on_device = real


def test_point_branch_projected_model_and_other_output_crs(c1):
    """(a) A Lambert-conformal-conic weather model (HRRR) with a lon/lat output grid: the intermediate cube is built with the
    projection on the device, stays there; == the host sequence.  (b) An output CRS other than lon/lat (UTM): the intermediate grid is
    in UTM metres, the query points are transformed (transformPoints on the GPU) and go up packed."""
    from raider_amd.delay import PointsAOI, tropo_delay
    from raider_amd.losreader import Zenith
    rng = np.random.default_rng(8)
    lcc = '+proj=lcc +lat_1=38.5 +lat_2=38.5 +lat_0=38.5 +lon_0=262.5 +x_0=0 +y_0=0 +a=6371229 +b=6371229 +units=m +no_defs'
    ny, nx, nz = 40, 44, 20
    xs = -60000.0 + 3000.0 * np.arange(nx); ys = -50000.0 + 3000.0 * np.arange(ny); zs = np.round(-100 + 26000 * np.linspace(0, 1, nz) ** 2, 3)
    f = lambda: rng.normal(2.0, 0.1, (nz, ny, nx))
    wm = dict(x=xs, y=ys, z=zs, wet=f().astype(np.float32), hydro=f().astype(np.float32), wet_total=f(), hydro_total=f(), proj=lcc)
    n = 500
    lats = rng.uniform(38.2, 38.8, n); lons = rng.uniform(-98.0, -97.1, n); hg = rng.uniform(0, 3000, n)
    xg = np.arange(-98.1, -97.0, 0.02); yg = np.arange(38.9, 38.1, -0.02)
    w, h = tropo_delay(WHEN, wm, PointsAOI(lats, lons, hg, xg, yg), Zenith(), list(zs[:12]), 4326, None)
    from raider_amd.delay import _get_delays_on_cube, transformPoints
    from raider_amd.delayFcns import getInterpolators
    ds = _get_delays_on_cube(WHEN, wm, lcc, PointsAOI(lats, lons, hg, xg, yg), list(zs[:12]), Zenith(), 4326, zs.max() - 1)
    iw, ih = getInterpolators(ds, 'ztd')
    pn = transformPoints(lats, lons, hg, 4326, 4326)
    assert np.array_equal(w, iw(pn)) and np.array_equal(h, ih(pn)) and np.isfinite(w).all()
    # (b) UTM 14N output grid over the same model... the model is lon/lat here (c1), the grid UTM 11N
    wm1 = _wm(c1)
    la = rng.uniform(32.0, 34.0, n); lo = rng.uniform(-118.5, -116.5, n)
    yx = transformPoints(la, lo, 0 * la, 4326, 32611)
    xu = np.arange(yx[:, 1].min() - 5000, yx[:, 1].max() + 5000, 4000.0); yu = np.arange(yx[:, 0].max() + 5000, yx[:, 0].min() - 5000, -4000.0)
    wu, hu = tropo_delay(WHEN, wm1, PointsAOI(la, lo, hg, xu, yu), Zenith(), list(c1['zs'][:15]), 32611, None)
    ds = _get_delays_on_cube(WHEN, wm1, 4326, PointsAOI(la, lo, hg, xu, yu), list(c1['zs'][:15]), Zenith(), 32611, c1['zs'].max() - 1)
    iw, ih = getInterpolators(ds, 'ztd')
    pn = transformPoints(la, lo, hg, 4326, 32611)
    assert np.array_equal(wu, iw(pn), equal_nan=True) and np.array_equal(hu, ih(pn), equal_nan=True) and np.isfinite(wu).mean() > 0.9


def test_point_delays_is_the_two_entries_in_one(c1):
    """rdr_point_delays = rdr_build_cube_to_cube + rdr_interp3_project with the cube in scratch and the uploads under the build: the
    same bits, for point sets on both sides of the chunked-transfer threshold, every projection mode, descending grid axes, NaNs."""
    import raider_amd as R
    tot = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet_total'], c1['hydro_total'], order='zyx')
    xp = np.linspace(-119.5, -115.5, 61); yp = np.linspace(34.5, 31.5, 47); zp = np.array([-50.0, 0.0, 300.0, 1000.0, 5000.0, 9000.0])
    d = tot.build_delay_cube(xp, yp, zp)
    rng = np.random.default_rng(5)
    for n in (1000, 700_000):
        y = rng.uniform(31.4, 34.6, n); x = rng.uniform(-119.6, -115.4, n); z = rng.uniform(-100, 9500, n)
        inc = rng.uniform(20, 50, n)
        for kw in ({}, {'inc': inc}, {'inc': 33.0}, {'divisor': np.cos(np.radians(inc))}):
            a = d.interp_project(y, x, z, **kw)
            b = tot.point_delays(xp, yp, zp, y, x, z, **kw)
            assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True) and b[2] is False
            assert np.isnan(a[0]).any() and np.isfinite(a[0]).mean() > 0.8
        p = tot.point_delays(xp, yp, zp, np.stack([y, x, z], -1), inc=inc)
        assert np.array_equal(p[0], b[0] * 0 + d.interp_project(y, x, z, inc=inc)[0], equal_nan=True)
    # a grid leaving the model: NaN nodes in the intermediate cube, reported
    w, h, nan = tot.point_delays(np.linspace(-125.0, -115.5, 61), yp, zp, y[:100], x[:100], z[:100])
    assert nan is True
    # no points: the cube is still built and scanned
    w0, h0, nan0 = tot.point_delays(xp, yp, zp, np.zeros(0), np.zeros(0), np.zeros(0))
    assert w0.size == 0 and nan0 is False
    with pytest.raises(ValueError):
        tot.point_delays(xp, yp, np.array([0.0, 0.0]), y, x, z)


def test_interp3_one_field_soa_and_projection_modes(c1):
    """rdr_interp3 with one output NULL; rdr_interp3_project with the points as three arrays / packed, every projection mode; all the
    same gather: identical bits."""
    import raider_amd as R
    cube = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet_total'], c1['hydro_total'], order='zyx')
    rng = np.random.default_rng(0)
    n = 5000
    y = rng.uniform(c1['ys'].min() - 0.1, c1['ys'].max() + 0.1, n); x = rng.uniform(c1['xs'].min(), c1['xs'].max(), n); z = rng.uniform(-50, 30000, n)
    pts = np.stack([y, x, z], -1)
    w, h = cube.interp(pts)
    assert np.isnan(w).any() and np.isfinite(w).mean() > 0.8
    w0, none = cube.interp(pts, field=0)
    none1, h1 = cube.interp(pts, field=1)
    assert none is None and none1 is None and np.array_equal(w0, w, equal_nan=True) and np.array_equal(h1, h, equal_nan=True)
    a = cube.interp_project(y, x, z)
    b = cube.interp_project(pts)
    for r in (a, b):
        assert np.array_equal(r[0], w, equal_nan=True) and np.array_equal(r[1], h, equal_nan=True)
    inc = rng.uniform(20, 50, n)
    pi = cube.interp_project(y, x, z, inc=inc)
    assert _ulp_close(pi[0], w / np.cos(np.radians(inc))) and _ulp_close(pi[1], h / np.cos(np.radians(inc)))
    ps = cube.interp_project(pts, inc=39.0)
    assert _ulp_close(ps[1], h / np.cos(np.radians(39.0)))
    up = np.cos(np.radians(inc))
    pd = cube.interp_project(y, x, z, divisor=up)
    assert np.array_equal(pd[0], w / up, equal_nan=True) and np.array_equal(pd[1], h / up, equal_nan=True)
    # shapes: a (ny, nx) raster of points keeps its shape; a scalar height broadcasts
    yy, xx = np.meshgrid(np.linspace(32, 34, 7), np.linspace(-118, -116, 9), indexing='ij')
    r2 = cube.interp_project(yy, xx, 100.0, inc=np.full((7, 9), 30.0))
    assert r2[0].shape == (7, 9) and np.isfinite(r2[0]).all()
    with pytest.raises(ValueError):
        cube.interp_project(y, x, z, inc=inc, divisor=up)


def test_interpolator_calls_are_stateless(c1):
    """ADVICE r2's regression, kept: an array edited in place between the wet and the hydro call must never be served the other
    call's result.  Since round 4 nothing is handed over at all (and nothing hashed): every call gathers its own field."""
    import raider_amd as R
    from raider_amd.delayFcns import interpolators_from_cube
    cube = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet'], c1['hydro'], order='zyx')
    ifw, ifh = interpolators_from_cube(cube)
    rng = np.random.default_rng(2)
    p = np.stack([rng.uniform(31, 35, 1000), rng.uniform(-120, -115, 1000), rng.uniform(0, 9000, 1000)], -1)
    w = ifw(p)
    q = p.copy()
    p[::2, 2] += 500.0                                       # edited in place
    h_edit = ifh(p)
    assert np.array_equal(h_edit, cube.interp(p)[1]) and not np.array_equal(h_edit, cube.interp(q)[1])
    assert np.array_equal(w, cube.interp(q)[0])


def test_delay_cube_on_device_is_the_downloaded_cube(c1):
    """rdr_build_cube_to_cube / rdr_raytrace_slices_to_cube leave on the device exactly what rdr_build_cube / rdr_raytrace_slices
    download (axes incl. the descending y axis of an AOI grid flipped as scipy flips it; the NaN verdict of the packing = the scan)."""
    import raider_amd as R
    tot = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet_total'], c1['hydro_total'], order='zyx')
    xp = np.linspace(-119.5, -115.5, 33); yp = np.linspace(34.5, 31.5, 21); zp = np.array([0.0, 300.0, 1000.0, 5000.0])
    wet, hyd = tot.build_cube(xp, yp, zp)
    d = tot.build_delay_cube(xp, yp, zp)
    assert d.shape == (21, 33, 4) and d.dtype == np.float64 and not d.has_nan()
    assert np.array_equal(d.grid[0], yp[::-1]) and np.array_equal(d.grid[1], xp) and np.array_equal(d.grid[2], zp)
    rw, rh = d.read()                                                         # (y ascending, x, z)
    assert np.array_equal(rw, wet.transpose(1, 2, 0)[::-1]) and np.array_equal(rh, hyd.transpose(1, 2, 0)[::-1])
    xo = np.linspace(-125.0, -115.5, 33)
    assert tot.build_delay_cube(xo, yp, zp).has_nan()
    with pytest.raises(ValueError):
        tot.build_delay_cube(xp, yp, np.array([0.0, 300.0, 300.0]))        # scipy's grid rule: strictly monotonic
    with pytest.raises(ValueError):
        tot.build_delay_cube(xp, yp, np.array([0.0]))
    pw = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet'], c1['hydro'], order='zyx')
    zref = float(c1['zs'].max() - 1)
    rays = R.Rays.grid(xp, yp, inc=39.0, hd=-167.9)
    w2, h2, K, nparts, flags = pw.raytrace_slices(rays, zp, zref)
    dc, K2, np2, fl2 = pw.raytrace_slices_to_cube(R.Rays.grid(xp, yp, inc=39.0, hd=-167.9), zp, zref)
    rw, rh = dc.read()
    assert np.array_equal(rw, w2.transpose(1, 2, 0)[::-1]) and np.array_equal(rh, h2.transpose(1, 2, 0)[::-1])
    assert np.array_equal(K, K2) and np.array_equal(nparts, np2) and np.array_equal(flags, fl2)
    import torch                                                            # a device-resident ray batch takes the same entry
    lv = np.ascontiguousarray(R.Rays.grid(xp, yp, inc=39.0, hd=-167.9).look_vectors())
    dc2, _, _, _ = pw.raytrace_slices_to_cube(R.Rays.grid(xp, yp, los=lv), zp, zref)
    np.testing.assert_allclose(dc2.read()[1], rh, rtol=0, atol=1e-11)       # (look vectors as an array / made in the kernel: last bits)
    dc3, _, _, _ = pw.raytrace_slices_to_cube(R.Rays.grid(torch.from_numpy(xp).cuda(), torch.from_numpy(yp).cuda(), los=torch.from_numpy(lv).cuda()), zp, zref)
    torch.cuda.synchronize()
    assert np.array_equal(dc3.read()[1], dc2.read()[1]) and np.array_equal(dc3.read()[0], dc2.read()[0])


def test_weather_file_is_opened_and_uploaded_once(tmp_path, c1):
    """The opened file and its device cubes are cached by FILE IDENTITY (path, size, mtime, inode): a second tropo_delay on the same
    file re-reads and re-uploads nothing; a rewritten file is read again; RAIDER_HIP_FILE_CACHE=0 switches the cache off."""
    import os
    from scipy.io import netcdf_file
    from raider_amd import delayFcns as F

    def write(path, scale):
        with netcdf_file(str(path), 'w', version=2) as f:
            for dname, k in (('z', 'zs'), ('y', 'ys'), ('x', 'xs')):
                f.createDimension(dname, c1[k].size)
                f.createVariable(dname, 'f8', (dname,))[:] = c1[k]
            for k in ('wet', 'hydro'):
                f.createVariable(k, 'f4', ('z', 'y', 'x'))[:] = c1[k] * scale
            for k in ('wet_total', 'hydro_total'):
                f.createVariable(k, 'f8', ('z', 'y', 'x'))[:] = c1[k] * scale
    p = tmp_path / 'model.nc'
    write(p, 1.0)
    F.clear_file_cache()
    a = F.getInterpolators(str(p), 'total')
    b = F.getInterpolators(str(p), 'total')
    assert a[0].cube is b[0].cube and a[0] is not b[0]
    assert F.getInterpolators(str(p), 'pointwise')[0].cube is not a[0].cube
    q = np.array([[33.0, -117.0, 500.0]])
    v1 = a[1](q)
    write(p, 2.0)
    os.utime(p, ns=(os.stat(p).st_atime_ns, os.stat(p).st_mtime_ns + 5_000_000))     # (a coarse file-system clock must not hide the rewrite)
    c = F.getInterpolators(str(p), 'total')
    assert c[0].cube is not a[0].cube
    np.testing.assert_allclose(c[1](q), 2.0 * v1, rtol=1e-15)
    os.environ['RAIDER_HIP_FILE_CACHE'] = '0'
    try:
        assert F.getInterpolators(str(p), 'total')[0].cube is not c[0].cube
    finally:
        del os.environ['RAIDER_HIP_FILE_CACHE']
    F.clear_file_cache()


def test_interp_blend_is_blend_then_interp_bit_for_bit():
    """rdr_interp3_blend: the two-epoch temporal interpolation applied at the corners of every query point == rdr_cube_blend followed by
    rdr_interp3, bit for bit - f32 cubes (products and sum rounded in f32, as the blended cube holds them) and f64 cubes, host and device
    point arrays, point sets on both sides of the chunked-transfer threshold, NaN / outside points; mismatched epochs are refused."""
    import torch
    import raider_amd as R
    rng = np.random.default_rng(12)
    ny, nx, nz = 37, 41, 19
    ys = np.linspace(30.0, 36.0, ny)[::-1].copy(); xs = np.linspace(-121.0, -113.0, nx); zs = np.round(-100 + 30000 * np.linspace(0, 1, nz) ** 2, 3)
    for dt_ in (np.float32, np.float64):
        ea = [rng.normal(100, 30, (nz, ny, nx)).astype(dt_) for _ in range(2)]; eb = [rng.normal(100, 30, (nz, ny, nx)).astype(dt_) for _ in range(2)]
        a = R.Cube(ys, xs, zs, ea[0], ea[1], order='zyx'); b = R.Cube(ys, xs, zs, eb[0], eb[1], order='zyx')
        for w1, w2 in ((0.25, 0.75), (0.6041666666666667, 0.3958333333333333)):
            m = a.blend(w1, b, w2)
            for n in (3000, 300_000):
                q = np.stack([rng.uniform(29.9, 36.1, n), rng.uniform(-121.0, -113.0, n), rng.uniform(-150, 30100, n)], -1)
                q[0] = [np.nan, -117.0, 100.0]
                r0 = m.interp(q)
                r1 = a.interp_blend(w1, b, w2, q)
                assert np.array_equal(r0[0], r1[0], equal_nan=True) and np.array_equal(r0[1], r1[1], equal_nan=True)
                assert np.isnan(r1[0][0]) and 0.8 < np.isfinite(r1[0]).mean() < 1.0
            qd = torch.from_numpy(q).cuda()
            rd = a.interp_blend(w1, b, w2, qd)
            torch.cuda.synchronize()
            assert np.array_equal(rd[0].cpu().numpy(), r0[0], equal_nan=True) and np.array_equal(rd[1].cpu().numpy(), r0[1], equal_nan=True)
    other = R.Cube(ys, xs[:-1], zs, ea[0][:, :, :-1].copy(), ea[1][:, :, :-1].copy(), order='zyx')
    with pytest.raises(ValueError, match='same grid'):
        a.interp_blend(0.5, other, 0.5, q)
    # the sharded station query picks the cheaper of the two forms for the rank's block - same bits
    from raider_amd import distributed as D
    assert D.blend_on_the_fly_pays(a, 10) and not D.blend_on_the_fly_pays(a, 10_000_000)
    p0, cnt, w, h = D.interp_points_sharded(a, q, world=3, rank=1, blend=(0.25, b, 0.75))
    full = a.blend(0.25, b, 0.75).interp(q)
    assert np.array_equal(w, full[0][p0:p0 + cnt], equal_nan=True) and np.array_equal(h, full[1][p0:p0 + cnt], equal_nan=True)


def test_interp_blend_via_paired_cube_is_blend_then_interp_bit_for_bit():
    """rdr_interp3_blend_cube (round 6): the blend made as a cube in the context's scratch, x columns PAIRED (the layout a random gather reads
    in 3 lines per point instead of 4), then gathered in the same call == rdr_cube_blend + rdr_interp3 bit for bit: f32 / f64 cubes, odd and
    even nx and nz (the unpaired last column, the one-z-per-thread blend), descending axes, points ON the last nodes, NaN / outside points, host
    and device point arrays on both sides of the chunked-transfer threshold; mismatched epochs refused; a NaN in an epoch stays where it was."""
    import torch
    import raider_amd as R
    rng = np.random.default_rng(13)
    for (ny, nx, nz) in ((37, 41, 19), (20, 24, 12), (5, 2, 2), (4, 3, 7)):
        ys = np.linspace(30.0, 36.0, ny)[::-1].copy(); xs = np.linspace(-121.0, -113.0, nx); zs = np.round(-100 + 30000 * np.linspace(0, 1, nz) ** 2, 3)
        for dt_ in (np.float32, np.float64):
            ea = [rng.normal(100, 30, (nz, ny, nx)).astype(dt_) for _ in range(2)]; eb = [rng.normal(100, 30, (nz, ny, nx)).astype(dt_) for _ in range(2)]
            if nx == 24:
                ea[0][3, 7, 9] = np.nan
            a = R.Cube(ys, xs, zs, ea[0], ea[1], order='zyx'); b = R.Cube(ys, xs, zs, eb[0], eb[1], order='zyx')
            for w1, w2 in ((0.25, 0.75), (0.6041666666666667, 0.3958333333333333)):
                m = a.blend(w1, b, w2)
                for n in (3000, 300_000):
                    q = np.stack([rng.uniform(29.9, 36.1, n), rng.uniform(-121.0, -113.0, n), rng.uniform(-150, 30100, n)], -1)
                    q[0] = [np.nan, -117.0, 100.0]
                    q[1] = [36.0, -113.0, zs[-1]]; q[2] = [30.0, -121.0, zs[0]]; q[3] = [33.0, xs[-1], 500.0]; q[4] = [33.0, xs[-2], 500.0]      # on the last / first nodes
                    r0 = m.interp(q)
                    r1 = a.interp_blend(w1, b, w2, q, via_cube=True)
                    assert np.array_equal(r0[0], r1[0], equal_nan=True) and np.array_equal(r0[1], r1[1], equal_nan=True)
                    assert np.isnan(r1[0][0]) and np.isfinite(r1[0][1:5]).all() and 0.8 < np.isfinite(r1[0]).mean() < 1.0
                qd = torch.from_numpy(q).cuda()
                rd = a.interp_blend(w1, b, w2, qd, via_cube=True)
                torch.cuda.synchronize()
                assert np.array_equal(rd[0].cpu().numpy(), r0[0], equal_nan=True) and np.array_equal(rd[1].cpu().numpy(), r0[1], equal_nan=True)
    other = R.Cube(ys, xs[:-1], zs, ea[0][:, :, :-1].copy(), ea[1][:, :, :-1].copy(), order='zyx')
    with pytest.raises(ValueError, match='same grid'):
        a.interp_blend(0.5, other, 0.5, q, via_cube=True)
    # the sharded station query takes this route when the rank's block is large against the cube - same bits
    from raider_amd import distributed as D
    big = np.tile(q, (8, 1))
    assert not D.blend_on_the_fly_pays(a, big.shape[0])
    p0, cnt, w, h = D.interp_points_sharded(a, big, world=1, rank=0, blend=(0.25, b, 0.75))
    full = a.blend(0.25, b, 0.75).interp(big)
    assert np.array_equal(w, full[0], equal_nan=True) and np.array_equal(h, full[1], equal_nan=True)


def test_point_branch_randomised_against_the_two_stage_host_sequence():
    """40 random jobs: model cubes with exact / jittered / descending axes (f64 totals), AOI grids ascending or descending and partly outside
    the model, height lists that leave the model's z range, query points inside / outside / NaN, every projection mode.  The fused call
    (rdr_point_delays), the two-call form (rdr_build_cube_to_cube + rdr_interp3_project) and the reference's own sequence through the public
    host-array entries (rdr_build_cube -> download -> rdr_cube_create from the (z,y,x) arrays -> rdr_interp3) must agree BIT FOR BIT, NaN
    masks and the cube's NaN verdict included."""
    import raider_amd as R
    rng = np.random.default_rng(2024)
    for trial in range(40):
        ny, nx, nz = (int(v) for v in rng.integers(5, 40, 3))
        ys = np.linspace(30.0, 30.0 + 0.2 * ny, ny); xs = np.linspace(-120.0, -120.0 + 0.25 * nx, nx); zs = np.sort(rng.uniform(-200, 30000, nz))
        if trial % 3 == 1:
            ys = ys + rng.uniform(-0.03, 0.03, ny); xs = xs + rng.uniform(-0.03, 0.03, nx)
        if trial % 4 == 2:
            ys = ys[::-1].copy()
        shape = (nz, ny, nx)
        wt = rng.uniform(0.0, 0.4, shape); ht = rng.uniform(1.0, 2.5, shape)
        tot = R.Cube(ys, xs, zs, wt, ht, order='zyx')
        # AOI grid: sometimes reaching beyond the model, either direction
        gx = np.linspace(xs.min() - (0.3 if trial % 5 == 0 else -0.05), xs.max() - 0.05, int(rng.integers(2, 30)))
        gy = np.linspace(ys.max() - 0.04, ys.min() + (-0.2 if trial % 7 == 0 else 0.04), int(rng.integers(2, 30)))
        if trial % 2:
            gx = gx[::-1].copy()
        gz = np.sort(rng.uniform(zs.min() - (100 if trial % 6 == 0 else -1), zs.max() - 1, int(rng.integers(2, 12))))
        n = int(rng.integers(1, 4000))
        py = rng.uniform(gy.min() - 0.05, gy.max() + 0.05, n); px = rng.uniform(gx.min() - 0.05, gx.max() + 0.05, n); pz = rng.uniform(gz.min() - 20, gz.max() + 20, n)
        if n > 3:
            py[1] = np.nan
        inc = rng.uniform(15, 55, n)
        kw = ({}, {'inc': inc}, {'inc': 41.5}, {'divisor': np.cos(np.radians(inc))})[trial % 4]
        fw, fh, fnan = tot.point_delays(gx, gy, gz, py, px, pz, **kw)
        d = tot.build_delay_cube(gx, gy, gz)
        tw, th = d.interp_project(py, px, pz, **kw)
        bw, bh = tot.build_cube(gx, gy, gz)                                   # the reference's sequence: cube down ...
        d2 = R.Cube(gy, gx, gz, np.asarray(bw), np.asarray(bh), order='zyx')     # ... Dataset -> getInterpolators(ds): cube up ...
        hw, hh = d2.interp(np.stack([py, px, pz], -1))                        # ... the gather
        if kw:
            up = np.cos(np.radians(kw['inc'])) if 'inc' in kw else kw['divisor']
            if 'inc' in kw:          # (the host cos against the device's: compare the un-projected values exactly, the projected to 2 ulp)
                assert _ulp_close(fw, hw / up) and _ulp_close(fh, hh / up), trial
            else:
                assert np.array_equal(fw, hw / up, equal_nan=True) and np.array_equal(fh, hh / up, equal_nan=True), trial
        else:
            assert np.array_equal(fw, hw, equal_nan=True) and np.array_equal(fh, hh, equal_nan=True), trial
        assert np.array_equal(fw, tw, equal_nan=True) and np.array_equal(fh, th, equal_nan=True), trial
        assert fnan == d.has_nan() == bool(np.isnan(bw).any() or np.isnan(bh).any()), trial


def test_point_branch_from_four_threads_sharing_the_default_context(tmp_path, c1):
    """tropo_delay's point branch called from four threads at once: they share the default context (calls serialised by its lock), the
    file-identity cache (one upload of the model for all of them) and the context's scratch cube - every thread must get exactly the
    serial result of ITS stations, the 'missing delay values' verdict of ITS grid, and the cache must hold ONE cube for the file."""
    import threading
    from scipy.io import netcdf_file
    from raider_amd import delayFcns as F
    from raider_amd.delay import PointsAOI, tropo_delay
    from raider_amd.losreader import Conventional, Zenith
    p = tmp_path / 'model.nc'
    with netcdf_file(str(p), 'w', version=2) as f:
        for dname, k in (('z', 'zs'), ('y', 'ys'), ('x', 'xs')):
            f.createDimension(dname, c1[k].size)
            f.createVariable(dname, 'f8', (dname,))[:] = c1[k]
        for k in ('wet', 'hydro'):
            f.createVariable(k, 'f4', ('z', 'y', 'x'))[:] = c1[k]
        for k in ('wet_total', 'hydro_total'):
            f.createVariable(k, 'f8', ('z', 'y', 'x'))[:] = c1[k]
    F.clear_file_cache()
    rng = np.random.default_rng(31)
    jobs = []
    for t in range(4):
        n = 20000 + 7000 * t
        la = rng.uniform(31.6 + 0.1 * t, 34.4, n); lo = rng.uniform(-119.4, -115.6 - 0.1 * t, n); hg = rng.uniform(0, 2500, n)
        los = Zenith() if t % 2 == 0 else Conventional(inc=rng.uniform(25, 45, n), heading=np.zeros(n))
        jobs.append((la, lo, hg, los, [0.0, 400.0 + 100 * t, 1500.0, 3200.0]))

    def work(job, out, reps):
        la, lo, hg, los, hl = job
        for _ in range(reps):
            r = tropo_delay(WHEN, str(p), PointsAOI(la, lo, hg), los, hl, 4326, None)
        out.append(r)
    serial = []
    for j in jobs:
        work(j, serial, 1)
    outs = [[] for _ in jobs]
    th = [threading.Thread(target=work, args=(jobs[i], outs[i], 8)) for i in range(4)]
    for t_ in th:
        t_.start()
    for t_ in th:
        t_.join()
    for i in range(4):
        assert len(outs[i]) == 1
        assert np.array_equal(outs[i][0][0], serial[i][0], equal_nan=True) and np.array_equal(outs[i][0][1], serial[i][1], equal_nan=True)
        assert np.isfinite(serial[i][1]).mean() > 0.95
    assert sum(1 for k in F._CUBE_CACHE if k[1] == 'total') == 1
    F.clear_file_cache()
