"""GPU: a cached device cube is shared between calls, threads and model-CRS arguments and is NEVER modified (round 5).

The reference rebuilds its interpolators and pyproj transformers in every call (delay.py:196-216,238-253; delayFcns.py:23-58) and
shares nothing.  raider_amd caches ONE device cube per weather-model file; these tests pin the three ways that sharing could leak
into results: (a) a projection set for one caller seen by another (`Cube.view` / rdr_cube_view instead of re-projecting in place),
(b) per-call verdicts kept as attributes of the shared object, (c) a cached upload of a foreign interpolator whose `.values` were
edited in place.  Plus the lifetime rules of views, rdr_trim, the chunked fall-back of the point branch and the 2-level f64 cube
that the staged marcher must refuse."""
import datetime as dt
import threading

import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import raider_oracle as O

WHEN = dt.datetime(2020, 1, 1)
LCC = '+proj=lcc +lat_1=38.5 +lat_2=38.5 +lat_0=38.5 +lon_0=262.5 +x_0=0 +y_0=0 +a=6371229 +b=6371229'


def _lcc_model_file(path, seed=0):
    """A processed weather model on an LCC lattice (metres) - NetCDF-3, the four fields - small enough to build in milliseconds."""
    from scipy.io import netcdf_file
    rng = np.random.default_rng(seed)
    ny, nx, nz = 41, 47, 24
    ys = -180e3 + 9e3 * np.arange(ny); xs = -200e3 + 9e3 * np.arange(nx); zs = np.round(-50.0 + 30000.0 * np.linspace(0, 1, nz) ** 2, 2)
    z3 = zs[:, None, None]
    hyd = (270.0 * np.exp(-z3 / 8000.0) * (1 + 0.02 * rng.standard_normal((1, ny, nx)))).astype(np.float32)
    wet = (60.0 * np.exp(-z3 / 2000.0) * (1 + 0.1 * rng.standard_normal((1, ny, nx)))).astype(np.float32)
    ht = np.cumsum(hyd[::-1].astype(np.float64), 0)[::-1] * 1e-3; wt = np.cumsum(wet[::-1].astype(np.float64), 0)[::-1] * 1e-3
    with netcdf_file(str(path), 'w', version=2) as f:
        for dname, a in (('z', zs), ('y', ys), ('x', xs)):
            f.createDimension(dname, a.size)
            f.createVariable(dname, 'f8', (dname,))[:] = a
        for k, a in (('wet', wet), ('hydro', hyd)):
            f.createVariable(k, 'f4', ('z', 'y', 'x'))[:] = a
        for k, a in (('wet_total', wt), ('hydro_total', ht)):
            f.createVariable(k, 'f8', ('z', 'y', 'x'))[:] = a
    return ys, xs, zs


def test_two_threads_one_file_two_model_crs(tmp_path):
    """Two threads on ONE cached cube, 50 alternations each: one passes the file's LCC CRS with lon/lat nodes (projected on the device),
    the other model_crs = 4326 with nodes already in axis units (no projection).  With a projection that was state of the shared cube
    (rounds 3-4: clear_projection / set_projection on the cached object) either thread could run with the other's; now each call works
    on its own view.  Every result equals the serial one bit for bit; the cached cube itself never carries a projection."""
    from raider_amd import delayFcns as F
    from raider_amd.delay import _build_cube, _build_cube_ray
    from raider_amd.losreader import Raytracing
    p = tmp_path / 'lcc_model.nc'
    ys, xs, zs = _lcc_model_file(p)
    F.clear_file_cache()
    ifs = list(F.getInterpolators(str(p), 'total'))
    ifs_pw = list(F.getInterpolators(str(p), 'pointwise'))
    shared = ifs[0].cube
    zp = np.array([0.0, 700.0, 2500.0, 9000.0])
    lon = np.linspace(-98.6, -96.4, 33); lat = np.linspace(37.4, 39.6, 29)              # lon/lat nodes around the cone's origin
    ax_x = np.linspace(xs[2], xs[-3], 31); ax_y = np.linspace(ys[2], ys[-3], 27)         # nodes in axis units

    def job_lcc():
        return _build_cube(lon, lat, zp, LCC, 4326, ifs)

    def job_plain():
        return _build_cube(ax_x, ax_y, zp, 4326, 4326, ifs)

    def job_ray():
        return _build_cube_ray(lon[::4], lat[::4], zp[:2], Raytracing(inc=30.0, heading=-167.9), LCC, 4326, ifs_pw, MAX_TROPO_HEIGHT=float(zs.max() - 1))

    want_lcc, want_plain, want_ray = job_lcc(), job_plain(), job_ray()
    assert np.isfinite(want_lcc[1]).all() and np.isfinite(want_plain[1]).all() and np.isfinite(want_ray[1]).all()
    assert not np.array_equal(want_lcc[1][:, :27, :31], want_plain[1][:, :27, :31])
    assert shared.projection is None and ifs_pw[0].cube.projection is None
    bad = []

    def loop(job, want, n):
        for _ in range(n):
            got = job()
            if not (np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True)) or got.has_nan != want.has_nan:
                bad.append(job.__name__)
    th = [threading.Thread(target=loop, args=(job_lcc, want_lcc, 50)), threading.Thread(target=loop, args=(job_plain, want_plain, 50)),
          threading.Thread(target=loop, args=(job_ray, want_ray, 12))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not bad, bad
    assert shared.projection is None and ifs_pw[0].cube.projection is None          # nobody re-projected the cached cubes
    assert F.getInterpolators(str(p), 'total')[0].cube is shared                    # ... and it is still THE cached cube


def test_nan_verdict_belongs_to_the_call_not_to_the_shared_cube(tmp_path):
    """Two grids on one cached cube, one inside the model (no NaN), one reaching outside (NaN): built alternately from two threads,
    each result carries ITS OWN np.isnan(...).any() verdict (delay.py:187) - it is returned with the result (build_cube(want_nan=True),
    raytrace_slices(want_nan=True)), not read back from an attribute of the shared cube."""
    from raider_amd import delayFcns as F
    from raider_amd.delay import _build_cube
    p = tmp_path / 'lcc_model.nc'
    ys, xs, zs = _lcc_model_file(p, seed=1)
    F.clear_file_cache()
    ifs = list(F.getInterpolators(str(p), 'total'))
    zp = np.array([0.0, 1500.0])
    inside = (np.linspace(xs[1], xs[-2], 40), np.linspace(ys[1], ys[-2], 40))
    outside = (np.linspace(xs[0] - 5e4, xs[-2], 40), np.linspace(ys[1], ys[-2], 40))
    wrong = []

    def loop(grid, expect):
        for _ in range(60):
            r = _build_cube(grid[0], grid[1], zp, 4326, 4326, ifs)
            if r.has_nan is not expect or bool(np.isnan(r[0]).any()) is not expect:
                wrong.append((expect, r.has_nan))
    th = [threading.Thread(target=loop, args=(inside, False)), threading.Thread(target=loop, args=(outside, True))]
    for t in th:
        t.start()
    for t in th:
        t.join()
    assert not wrong, wrong[:4]
    assert not hasattr(ifs[0].cube, 'last_build_cube_has_nan') and not hasattr(ifs[0].cube, 'last_nan_output')


def test_foreign_interpolator_values_edited_in_place():
    """scipy RGIs (what the reference's getInterpolators returns) handed to _build_cube: the reference reads `.values` at call time, so an
    in-place edit between two calls must show - rounds 3-4 served the cached upload (validated by identity of the array only)."""
    from scipy.interpolate import RegularGridInterpolator as RGI
    from raider_amd.delay import _build_cube, _build_cube_ray
    from raider_amd.losreader import Raytracing
    c = O.synthetic_cube(30, 30, 20, seed=3)
    wt = np.ascontiguousarray(c['wet_total'].transpose(1, 2, 0)); ht = np.ascontiguousarray(c['hydro_total'].transpose(1, 2, 0))
    ifw = RGI((c['ys'], c['xs'], c['zs']), wt, bounds_error=False, fill_value=np.nan)
    ifh = RGI((c['ys'], c['xs'], c['zs']), ht, bounds_error=False, fill_value=np.nan)
    xp = np.linspace(c['xs'][1], c['xs'][-2], 17); yp = np.linspace(c['ys'][1], c['ys'][-2], 13); zp = np.array([0.0, 800.0, 4000.0])
    a = _build_cube(xp, yp, zp, 4326, 4326, [ifw, ifh])
    ifh.values[...] *= 2.0                                          # the SAME array object, new contents
    ifw.values[3:9, 4:11, :] += 0.125
    b = _build_cube(xp, yp, zp, 4326, 4326, [ifw, ifh])
    np.testing.assert_allclose(b[1], 2.0 * a[1], rtol=2e-16, atol=0)
    assert not np.array_equal(a[0], b[0])
    yy, xx = np.meshgrid(yp, xp, indexing='ij')
    want = ifw(np.stack([yy, xx, np.full(yy.shape, 800.0)], -1))    # scipy itself on the edited array
    np.testing.assert_allclose(b[0][1], want, rtol=0, atol=1e-13)
    # the ray-traced entry takes the same route
    pw = RGI((c['ys'], c['xs'], c['zs']), np.ascontiguousarray(c['wet'].transpose(1, 2, 0)).astype(np.float64), bounds_error=False, fill_value=np.nan)
    ph = RGI((c['ys'], c['xs'], c['zs']), np.ascontiguousarray(c['hydro'].transpose(1, 2, 0)).astype(np.float64), bounds_error=False, fill_value=np.nan)
    zref = float(c['zs'].max() - 1)
    r1 = _build_cube_ray(xp, yp, zp[:1], Raytracing(inc=12.0, heading=-167.9), 4326, 4326, [pw, ph], MAX_TROPO_HEIGHT=zref)
    ph.values[...] *= 3.0
    r2 = _build_cube_ray(xp, yp, zp[:1], Raytracing(inc=12.0, heading=-167.9), 4326, 4326, [pw, ph], MAX_TROPO_HEIGHT=zref)
    np.testing.assert_allclose(r2[1], 3.0 * np.asarray(r1[1]), rtol=1e-14, atol=0)
    assert np.array_equal(r2[0], r1[0])


def test_view_shares_buffers_and_outlives_its_source():
    """rdr_cube_view: a view gathers the source's values without a copy, carries its own projection, and keeps the buffers alive when the
    source handle is destroyed first (the C side counts views); the corner-quad copy built through a view belongs to the buffers."""
    import gc
    import raider_amd as R
    c = O.synthetic_cube(40, 44, 30, seed=5)
    src = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    rng = np.random.default_rng(1)
    pts = np.stack([rng.uniform(c['ys'][0], c['ys'][-1], 5000), rng.uniform(c['xs'][0], c['xs'][-1], 5000), rng.uniform(c['zs'][0], c['zs'][-1], 5000)], -1)
    want = src.interp(pts)
    v = src.view(None)
    v2 = src.view(dict(proj='lcc', lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0))
    assert v.projection is None and v2.projection['proj'] == 'lcc' and src.projection is None
    assert v.shape == src.shape and v.grid is src.grid
    assert v.point_index(True) > 0 and src.point_index(True) == v.point_index(True)      # one copy, seen through both handles
    got = v.interp(pts)
    assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True)
    # a projected view projects geodetic nodes; the source, asked the same, does not
    lon = np.linspace(-97.6, -97.4, 5); lat = np.linspace(38.4, 38.6, 5)
    y2, x2 = v2.project(lat, lon)
    y0, x0 = src.project(lat, lon)
    assert np.abs(y2).max() > 1e3 and np.array_equal(y0, lat) and np.array_equal(x0, lon)
    # destroy the source handle first: the views still read the same buffers
    lib, h = src.ctx.lib, src.handle
    src.handle = None
    lib.rdr_cube_destroy(h)
    v._source = None; v2._source = None
    del src
    gc.collect()
    junk = R.Cube(c['ys'], c['xs'], c['zs'], np.zeros_like(c['wet']), np.zeros_like(c['hydro']), order='zyx')   # would recycle a pooled buffer
    got = v.interp(pts)
    assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True)
    assert float(np.nanmax(junk.interp(pts)[0])) == 0.0
    del v, v2, junk
    gc.collect()


def test_trim_gives_scratch_back():
    """rdr_trim: after a job that left large scratch behind, the context hands the memory back and still works afterwards."""
    import torch
    import raider_amd as R
    ctx = R.Context()
    c = O.synthetic_cube(50, 50, 40, seed=0)
    tot = R.Cube(c['ys'], c['xs'], c['zs'], c['wet_total'], c['hydro_total'], order='zyx', ctx=ctx)
    gx = np.linspace(c['xs'][1], c['xs'][-2], 700); gy = np.linspace(c['ys'][1], c['ys'][-2], 600); gz = np.linspace(0.0, 9000.0, 30)
    rng = np.random.default_rng(0)
    py, px, pz = rng.uniform(gy[0], gy[-1], 20000), rng.uniform(gx[0], gx[-1], 20000), rng.uniform(0, 9000, 20000)
    w0, h0, _ = tot.point_delays(gx, gy, gz, py, px, pz)                 # 12.6 M cells: 200 MB of cube + 200 MB of planar results in scratch
    free0 = torch.cuda.mem_get_info()[0]
    released = ctx.trim(1 << 20)
    free1 = torch.cuda.mem_get_info()[0]
    assert released >= 350 << 20 and free1 - free0 >= 300 << 20, (released, free1 - free0)
    w1, h1, _ = tot.point_delays(gx, gy, gz, py, px, pz)
    assert np.array_equal(w0, w1, equal_nan=True) and np.array_equal(h0, h1, equal_nan=True)
    assert ctx.trim(0) > 0 and ctx.trim(0) == 0
    with pytest.raises(ValueError):
        ctx.trim(-1)


def test_point_branch_falls_back_to_chunks_when_the_cube_would_not_fit(monkeypatch):
    """the device-resident point branch must not raise out-of-memory for a job the chunked sequence completes.  With
    the slice budget set below the job's needs the ray-traced point branch takes the host sequence (_build_cube_ray in chunks) - same
    delays; and a DeviceOutOfMemory raised inside the one-call route is turned into that fall-back instead of escaping tropo_delay."""
    import raider_amd.delay as D
    from raider_amd import _lib as L
    from raider_amd.delay import PointsAOI, tropo_delay
    from raider_amd.losreader import Raytracing, Zenith
    c = O.synthetic_cube(50, 50, 40, seed=0)
    wm = dict(x=c['xs'], y=c['ys'], z=c['zs'], wet=c['wet'], hydro=c['hydro'], wet_total=c['wet_total'], hydro_total=c['hydro_total'])
    rng = np.random.default_rng(4)
    n = 3000
    la = rng.uniform(32.2, 33.8, n); lo = rng.uniform(-118.8, -116.2, n); hg = rng.uniform(0, 2500, n)
    hl = [0.0, 500.0, 1500.0, 3200.0]

    los = Raytracing(inc=31.0, heading=-167.9)
    def aoi():
        a = PointsAOI(la, lo, hg); a.set_output_spacing(0.05); a.set_output_xygrid(4326)
        return a
    full = tropo_delay(WHEN, wm, aoi(), los, hl, 4326, None)
    monkeypatch.setenv('RAIDER_HIP_SLICE_BUDGET_BYTES', '200000')      # a few slices of this grid at most: the one-call route declines
    chunked = tropo_delay(WHEN, wm, aoi(), los, hl, 4326, None)
    assert np.array_equal(full[0], chunked[0], equal_nan=True) and np.array_equal(full[1], chunked[1], equal_nan=True)
    assert np.isfinite(full[1]).mean() > 0.9
    monkeypatch.delenv('RAIDER_HIP_SLICE_BUDGET_BYTES')
    # zenith route: the C entry reports RDR_ERR_OOM -> the chunked sequence answers
    calls = {'n': 0}
    real = D.Cube.point_delays

    def boom(self, *a, **k):
        calls['n'] += 1
        raise L.DeviceOutOfMemory('simulated: hipMalloc of the intermediate cube failed')
    zen = tropo_delay(WHEN, wm, aoi(), Zenith(), hl, 4326, None)
    monkeypatch.setattr(D.Cube, 'point_delays', boom)
    zen2 = tropo_delay(WHEN, wm, aoi(), Zenith(), hl, 4326, None)
    monkeypatch.setattr(D.Cube, 'point_delays', real)
    assert calls['n'] == 1
    assert np.array_equal(zen[0], zen2[0], equal_nan=True) and np.array_equal(zen[1], zen2[1], equal_nan=True)


def test_two_level_f64_cube_is_not_staged():
    """an f64 cube with TWO z levels on exactly uniform axes - the LDS-staged marcher loads z entries zb .. zb+2 and must
    not be chosen (nz >= 3); the direct gathers give the oracle's delays."""
    import raider_amd as R
    from oracle import oracle_c as OC
    ny = nx = 12
    ys = 30.0 + 0.1 * np.arange(ny); xs = -100.0 + 0.1 * np.arange(nx); zs = np.array([-100.0, 3000.0])       # (below the origins: a first sample ON the lowest node is in or out by round-off, as in the reference)
    rng = np.random.default_rng(7)
    wet = rng.uniform(5, 60, (2, ny, nx)); hyd = rng.uniform(150, 300, (2, ny, nx))
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx')
    assert cube.dtype == np.float64
    xp = np.linspace(-99.7, -99.3, 40); yp = np.linspace(30.3, 30.7, 36)
    zref = 2999.0
    w, h, nparts, flags = cube.raytrace(R.Rays.grid(xp, yp, inc=20.0, hd=-167.9), 0.0, zref)
    assert len(nparts) == 1 and np.isfinite(w).all() and np.isfinite(h).all()
    xx, yy = np.meshgrid(xp, yp)
    los = O.look_vectors_from_inc_hd(np.full(yy.shape, 20.0), np.full(yy.shape, -167.9), yy, xx, 0.0)
    c = dict(ys=ys, xs=xs, zs=zs, wet=wet, hydro=hyd)
    cw, ch, _ = OC.build_cube_ray_slice(c, xp, yp, 0.0, los, zref, nparts=nparts)
    np.testing.assert_allclose(w, cw, rtol=0, atol=1e-9)
    np.testing.assert_allclose(h, ch, rtol=0, atol=1e-9)


def test_cubes_from_device_sources_are_created_without_a_host_wait_and_still_ordered():
    """Round 5: rdr_cube_create from DEVICE arrays (tensors, intermediate delay cubes) and rdr_cube_blend no longer end with a host
    synchronisation - a ready event orders other streams after them, the packing kernel's NaN verdict is read lazily.  (a) a cube made on one
    torch stream and queried at once from ANOTHER stream gives the values of a synchronous build; (b) the lazy verdict is right for NaN and
    NaN-free cubes, also after more creations than the context has verdict words (256: unasked ones are settled on recycling); (c) a blend
    of two fresh cubes, read back immediately, is the host arithmetic bit for bit."""
    import torch
    import raider_amd as R
    dev = torch.device('cuda:0')
    c = O.synthetic_cube(60, 64, 40, seed=11)
    rng = np.random.default_rng(2)
    n = 40000
    pts = np.stack([rng.uniform(c['ys'][0], c['ys'][-1], n), rng.uniform(c['xs'][0], c['xs'][-1], n), rng.uniform(c['zs'][0], c['zs'][-1], n)], -1)
    ref = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')              # host sources: the synchronous route
    want = ref.interp(pts)
    pt = torch.from_numpy(pts).to(dev)
    s1, s2 = torch.cuda.Stream(device=dev), torch.cuda.Stream(device=dev)
    torch.cuda.synchronize()
    for _ in range(20):
        with torch.cuda.stream(s1):
            w = torch.from_numpy(c['wet']).to(dev, non_blocking=True); h = torch.from_numpy(c['hydro']).to(dev, non_blocking=True)
            cube = R.Cube(c['ys'], c['xs'], c['zs'], w, h, order='zyx')                      # packed on s1, nobody waits
        with torch.cuda.stream(s2):
            gw, gh = cube.interp(pt)                                                         # ... queried on s2 at once
        s2.synchronize()
        assert np.array_equal(gw.cpu().numpy(), want[0], equal_nan=True) and np.array_equal(gh.cpu().numpy(), want[1], equal_nan=True)
        assert cube.has_nan() is False
        del cube, w, h
    # (b) lazy verdicts, more cubes alive than verdict words
    small = O.synthetic_cube(12, 13, 9, seed=1)
    wt = torch.from_numpy(small['wet']).to(dev); ht = torch.from_numpy(small['hydro']).to(dev)
    wn = wt.clone(); wn[3, 4, 5] = float('nan')
    cubes = []
    for k in range(300):
        cubes.append(R.Cube(small['ys'], small['xs'], small['zs'], wn if k % 3 == 0 else wt, ht, order='zyx'))
    got = [q.has_nan() for q in cubes]
    assert got == [k % 3 == 0 for k in range(300)]
    assert cubes[0].view(None).has_nan() is True and cubes[1].view(None).has_nan() is False   # a view's verdict is its source's
    del cubes
    # (c) blend of two fresh device cubes, read back at once
    a = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet']).to(dev), torch.from_numpy(c['hydro']).to(dev), order='zyx')
    c2 = O.synthetic_cube(60, 64, 40, seed=12)
    b = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c2['wet']).to(dev), torch.from_numpy(c2['hydro']).to(dev), order='zyx')
    m = a.blend(0.25, b, 0.75)
    mw, mh = m.read()
    ew = (np.float32(0.25) * c['wet'] + np.float32(0.75) * c2['wet']).transpose(1, 2, 0)
    assert np.array_equal(mw, O.blend_cubes(0.25, c['wet'], 0.75, c2['wet']).transpose(1, 2, 0)) or np.array_equal(mw, ew)
    assert m.has_nan() is False


def test_cube_creation_does_not_wait_for_the_stream_and_forgotten_streams_are_left_alone():
    """Round 6: (a) the axes of a new cube go up through the context's page-locked ring, so making a cube from DEVICE arrays returns
    while a long kernel queued earlier on the same stream is still running (a pageable hipMemcpyAsync first waits for the stream), and the host
    axes arrays may be overwritten at once; (b) rdr_forget_stream: a caller stream that is about to be destroyed is dropped from the set the
    context records events on - cubes used on it are still recycled correctly afterwards; (c) out of memory in a cube allocation drains the
    context's own pool of retired buffers before giving up (rdr_trim reports what is left)."""
    import time
    import torch
    import raider_amd as R
    from raider_amd.synthetic import synthetic_cube, scene_grid
    dev = torch.device('cuda:0')
    ctx = R.Context.default()
    c = synthetic_cube(300, 300, 80, seed=0)
    big = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    zref = float(c['zs'].max() - 1.0)
    xp, yp, inc_cols, hd = scene_grid(4000, 4000)
    xt, yt = torch.from_numpy(xp).to(dev), torch.from_numpy(yp).to(dev)
    inc = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (4000, 4000)))).to(dev)
    rays = R.Rays.grid(xt, yt, inc=inc, hd=torch.full((4000, 4000), hd, dtype=torch.float64, device=dev))
    ow = torch.empty((4000, 4000), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
    big.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=False)            # warm-up (workspace)
    torch.cuda.synchronize()
    small = synthetic_cube(40, 44, 20, seed=3)
    wt, ht = torch.from_numpy(small['wet']).to(dev), torch.from_numpy(small['hydro']).to(dev)
    ref = R.Cube(small['ys'], small['xs'], small['zs'], small['wet'], small['hydro'], order='zyx')
    q = np.stack([np.linspace(30.5, 35.5, 500), np.linspace(-120.5, -113.5, 500), np.linspace(0.0, 30000.0, 500)], -1)
    want = ref.interp(q)
    t0 = time.perf_counter()
    for _ in range(5):
        big.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=False)        # ~30 ms of queued kernels
    t_enq = time.perf_counter() - t0
    ys = small['ys'].copy(); xs = small['xs'].copy(); zs = small['zs'].copy()
    t0 = time.perf_counter()
    cube = R.Cube(ys, xs, zs, wt, ht, order='zyx')                             # device sources: axes through the ring, packing kernel queued
    t_make = time.perf_counter() - t0
    ys[:] = 0.0; xs[:] = 0.0; zs[:] = 0.0                                     # the axes were consumed by the call
    torch.cuda.synchronize()
    t_all = time.perf_counter() - t0
    assert t_all > 0.010, (t_enq, t_make, t_all)                              # the queue really was long ...
    assert t_make < 0.5 * t_all, (t_enq, t_make, t_all)                       # ... and the creation did not wait for it
    got = cube.interp(q)
    assert np.array_equal(got[0], want[0], equal_nan=True) and np.array_equal(got[1], want[1], equal_nan=True)
    # (b) a short-lived caller stream
    h = torch.cuda.Stream(device=dev)
    with torch.cuda.stream(h):
        qt = torch.from_numpy(q).to(dev)
        c2 = R.Cube(small['ys'], small['xs'], small['zs'], wt, ht, order='zyx')
        gw, gh = c2.interp(qt)
    h.synchronize()
    assert np.array_equal(gw.cpu().numpy(), want[0], equal_nan=True)
    ctx.forget_stream(h.cuda_stream)
    del c2                                                                    # retired: no event may be recorded on the forgotten stream
    c3 = R.Cube(small['ys'], small['xs'], small['zs'], wt, ht, order='zyx')    # picks the pooled buffer up again
    g3 = c3.interp(q)
    assert np.array_equal(g3[0], want[0], equal_nan=True)
    ctx.forget_stream(h.cuda_stream)                                          # (idempotent)
    # (c) the pool is the context's to give back
    del c3, cube
    freed = ctx.trim(0)
    assert freed > 0
