"""GPU parity: the HIP engine (through the C ABI) against the oracle and the reference goldens."""
import numpy as np
import pytest

pytestmark = pytest.mark.gpu

from oracle import raider_oracle as O

TOL = 1e-6     # metres - the north_star tolerance for wet/hydro delays (observed: ~1e-12)
TIGHT = 1e-9


@pytest.fixture(scope='module')
def R():
    import raider_amd
    return raider_amd


@pytest.fixture(scope='module')
def c1():
    return O.synthetic_cube(50, 50, 40, seed=0)


@pytest.fixture(scope='module')
def cubes(R, c1):
    pw = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet'], c1['hydro'], order='zyx')
    tot = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet_total'], c1['hydro_total'], order='zyx')
    return pw, tot


def test_interp3_matches_scipy_semantics(R):
    from scipy.interpolate import RegularGridInterpolator
    rng = np.random.default_rng(5)
    ys = np.linspace(3, -2, 6); xs = np.sort(rng.uniform(0, 5, 7)); zs = np.array([0., 1., 3., 7.])
    for dt in (np.float32, np.float64):
        w = rng.normal(size=(6, 7, 4)).astype(dt); h = rng.normal(size=(6, 7, 4)).astype(dt)
        cube = R.Cube(ys, xs, zs, w, h, order='yxz')
        q = np.stack([rng.uniform(-2.5, 3.5, 5000), rng.uniform(-0.5, 5.5, 5000), rng.uniform(-1, 8, 5000)], -1)
        q[:6] = [[3, xs[0], 0], [-2, xs[-1], 7], [ys[2], xs[3], zs[1]], [np.nan, 1, 1], [0, 1, 7.0000001], [0, 1, 7]]
        gw, gh = cube.interp(q)
        sw = RegularGridInterpolator((ys, xs, zs), w, fill_value=np.nan, bounds_error=False)(q)
        sh = RegularGridInterpolator((ys, xs, zs), h, fill_value=np.nan, bounds_error=False)(q)
        np.testing.assert_allclose(gw, sw, rtol=0, atol=1e-14, equal_nan=True)
        np.testing.assert_allclose(gh, sh, rtol=0, atol=1e-14, equal_nan=True)
        assert np.isnan(gw[3]) and np.isnan(gw[4]) and not np.isnan(gw[5])


def test_cube_roundtrip_and_orders(R, c1):
    a = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet'], c1['hydro'], order='zyx')
    w, h = a.read()
    assert np.array_equal(w, c1['wet'].transpose(1, 2, 0)) and np.array_equal(h, c1['hydro'].transpose(1, 2, 0))
    b = R.Cube(c1['ys'][::-1], c1['xs'], c1['zs'], np.ascontiguousarray(w[::-1]), np.ascontiguousarray(h[::-1]), order='yxz')
    w2, _ = b.read()
    assert np.array_equal(w2, w)           # descending axis flipped on device, like scipy
    assert np.array_equal(b.grid[0], c1['ys'])


def test_g4_build_cube(R, golden, cubes):
    g = golden('g4_build_cube')
    _, tot = cubes
    wet, hydro = tot.build_cube(g['xpts'], g['ypts'], g['zpts'])
    np.testing.assert_allclose(wet, g['wet'], rtol=0, atol=1e-14)
    np.testing.assert_allclose(hydro, g['hydro'], rtol=0, atol=1e-14)
    wet2, hydro2 = tot.build_cube(g['xp2'], g['yp2'], g['zp2'])
    np.testing.assert_allclose(wet2, g['wet2'], rtol=0, atol=1e-14, equal_nan=True)
    np.testing.assert_allclose(hydro2, g['hydro2'], rtol=0, atol=1e-14, equal_nan=True)


@pytest.mark.parametrize('tag', ['fixed', 'pp'])
def test_g5_raytrace_c1(R, golden, cubes, tag):
    g = golden('g5_build_cube_ray')
    pw, _ = cubes
    zref = float(g['c1_zref'])
    inc = 39.0 if tag == 'fixed' else g['c1_pp_inc']
    for i, ht in enumerate(g['c1_zpts']):
        rays = R.Rays.grid(g['c1_xpts'], g['c1_ypts'], inc=inc, hd=-167.9)
        wet, hyd, nparts, flags = pw.raytrace(rays, ht, zref)
        assert np.array_equal(nparts, g[f'c1_{tag}_nparts{i}'])
        np.testing.assert_allclose(wet, g[f'c1_{tag}_wet'][i], rtol=0, atol=TIGHT)
        np.testing.assert_allclose(hyd, g[f'c1_{tag}_hydro'][i], rtol=0, atol=TIGHT)


def test_g5_los_vector_input_and_maxseg(R, golden, cubes):
    g = golden('g5_build_cube_ray')
    pw, _ = cubes
    xx, yy = np.meshgrid(g['c1_xpts'], g['c1_ypts'])
    los = O.look_vectors_from_inc_hd(np.full(yy.shape, 20.0), np.full(yy.shape, -12.1), yy, xx, 100.0)
    rays = R.Rays.grid(g['c1_xpts'], g['c1_ypts'], los=los)
    wet, hyd, nparts, _ = pw.raytrace(rays, 100.0, 26000.0, max_seg=500.0)
    assert np.array_equal(nparts, g['c1_z26_nparts0'])
    np.testing.assert_allclose(wet, g['c1_z26_wet'][0], rtol=0, atol=TIGHT)
    np.testing.assert_allclose(hyd, g['c1_z26_hydro'][0], rtol=0, atol=TIGHT)


def test_g5_lateral_exit_nan(R, golden, cubes):
    g = golden('g5_build_cube_ray')
    pw, _ = cubes
    rays = R.Rays.grid(g['c1_edge_xpts'], g['c1_edge_ypts'], inc=45.0, hd=-167.9)
    wet, hyd, nparts, _ = pw.raytrace(rays, 0.0, float(g['c1_zref']))
    assert np.array_equal(nparts, g['c1_edge_nparts0'])
    assert np.array_equal(np.isnan(wet), np.isnan(g['c1_edge_wet'][0]))
    np.testing.assert_allclose(wet, g['c1_edge_wet'][0], rtol=0, atol=TIGHT, equal_nan=True)
    np.testing.assert_allclose(hyd, g['c1_edge_hydro'][0], rtol=0, atol=TIGHT, equal_nan=True)


def test_g5_constant_refractivity_invariant(R, golden, c1):
    g = golden('g5_build_cube_ray')
    ones = np.ones_like(c1['wet'])
    cube = R.Cube(c1['ys'], c1['xs'], c1['zs'], ones, ones, order='zyx')
    zref = float(g['c1_zref'])
    xx, yy = np.meshgrid(g['c1_one_xpts'], g['c1_one_ypts'])
    for i, ht in enumerate(g['c1_zpts']):
        rays = R.Rays.grid(g['c1_one_xpts'], g['c1_one_ypts'], inc=39.0, hd=-167.9)
        wet, hyd, _, _ = cube.raytrace(rays, ht, zref)
        np.testing.assert_allclose(wet, g['c1_one_wet'][i], rtol=0, atol=TIGHT)
        xyz = np.stack(O.lla2ecef(yy, xx, np.full(yy.shape, ht)), -1)
        los = O.look_vectors_from_inc_hd(np.full(yy.shape, 39.0), np.full(yy.shape, -167.9), yy, xx, ht)
        Lk, _, _ = O.build_ray(c1['zs'], ht, xyz, los, zref)
        # delay*1e6 == sum of ray lengths (test/test_synthetic.py:217-274 asserts 6 decimals).  The kernel's level
        # crossings use the exact geodetic height, the oracle's use PROJ's single-pass formula (off by 1.7e-5 m at
        # 40 km): total ray length agrees to ~1e-5 m of 5e4 m, i.e. 1e-11 m of delay.
        np.testing.assert_allclose(wet * 1e6, Lk.sum(0), rtol=0, atol=1e-4)


def test_g5_big_cube(R, golden):
    g = golden('g5_build_cube_ray')
    big = O.synthetic_cube(300, 300, 80, seed=0)
    cube = R.Cube(big['ys'], big['xs'], big['zs'], big['wet'], big['hydro'], order='zyx')
    rays = R.Rays.grid(g['big_xpts'], g['big_ypts'], inc=g['big_inc'], hd=-167.9)
    wet, hyd, nparts, _ = cube.raytrace(rays, 0.0, float(g['big_zref']))
    assert np.array_equal(nparts, g['big_nparts0'])
    np.testing.assert_allclose(wet, g['big_wet'][0], rtol=0, atol=TIGHT)
    np.testing.assert_allclose(hyd, g['big_hydro'][0], rtol=0, atol=TIGHT)


def test_g5b_shards_need_global_nparts(R, golden, cubes):
    g = golden('g5b_whole_vs_halves')
    pw, _ = cubes
    zref = float(g['zref'])
    xp, yp, inc = g['xpts'], g['ypts'], g['inc']
    whole = R.Rays.grid(xp, yp, inc=inc, hd=-167.9)
    maxlen, flags = pw.ray_prepass(whole, 0.0, zref)
    nparts = R.nparts_from_maxlen(maxlen)
    assert np.array_equal(nparts, g['nparts'])
    mls = []
    for sl in (slice(0, 32), slice(32, 64)):
        rays = R.Rays.grid(xp[sl], yp, inc=np.ascontiguousarray(inc[:, sl]), hd=-167.9)
        ml, fl = pw.ray_prepass(rays, 0.0, zref)
        mls.append(ml)
        wet, hyd = pw.ray_march(rays, 0.0, zref, nparts, flags)
        np.testing.assert_allclose(hyd, g['hydro'][0][:, sl], rtol=0, atol=TIGHT)
        np.testing.assert_allclose(wet, g['wet'][0][:, sl], rtol=0, atol=TIGHT)
    assert np.array_equal(np.maximum(*mls), maxlen)       # MAX all-reduce of shard maxima == whole-slice maxima
    # shard-local partition reproduces the reference run on the half, which differs from the whole by > 1e-6 m
    rays = R.Rays.grid(xp[:32], yp, inc=np.ascontiguousarray(inc[:, :32]), hd=-167.9)
    wet, hyd, npl, _ = pw.raytrace(rays, 0.0, zref)
    assert np.array_equal(npl, g['left_nparts'])
    np.testing.assert_allclose(hyd, g['left_hydro'][0], rtol=0, atol=TIGHT)
    assert np.abs(hyd - g['hydro'][0][:, :32]).max() > 1e-6


@pytest.mark.parametrize('region', ['arctic', 'equator_east', 'south', 'dateline', 'polar'])
def test_raytrace_domain_sweep(R, region):
    """The light-fp64 geodesy (delta lat/lon, rsq/rcp + NR, exact-height Newton) across latitudes, hemispheres,
    incidence 15..60 deg and all headings, against the oracle on the same inputs."""
    box = {'arctic': (68.0, 76.0, 10.0, 40.0), 'equator_east': (-4.0, 4.0, 95.0, 105.0),
           'south': (-48.0, -40.0, -75.0, -63.0), 'dateline': (10.0, 18.0, 168.0, 179.9),
           'polar': (82.0, 89.9, -60.0, 60.0)}[region]     # polar: most rays fail the static classification -> generic kernels
    c = O.synthetic_cube(40, 44, 36, seed=7, y0=box[0], y1=box[1], x0=box[2], x1=box[3])
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro']))
    rng = np.random.default_rng(3)
    ny, nx = 20, 24
    ymid, xmid = 0.5 * (box[0] + box[1]), 0.5 * (box[2] + box[3])
    ypts = np.linspace(ymid + 1.0, ymid - 1.0, ny); xpts = np.linspace(xmid - 1.2, xmid + 1.2, nx)
    if region == 'polar':
        ypts = np.linspace(89.2, 84.0, ny); xpts = np.linspace(-20.0, 20.0, nx)
    inc = rng.uniform(15, 60, (ny, nx)); hd = rng.uniform(-180, 180, (ny, nx))
    zref = c['zs'].max() - 1
    for ht in (0.0, 1500.0):
        look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(inc, hd, llh[1], llh[0], llh[2])
        (ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([ht]), look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True)
        wet, hyd, nparts, _ = cube.raytrace(R.Rays.grid(xpts, ypts, inc=inc, hd=hd), ht, zref)
        assert np.array_equal(nparts, onp[0])
        assert np.array_equal(np.isnan(wet), np.isnan(ow[0]))
        assert np.isfinite(ow[0]).mean() > 0.5
        # 5e-9 m (200x inside the 1e-6 m tolerance): at 55-60 deg incidence the crossing iteration contracts more slowly,
        # so the cheap early iterates move the final crossing by ~1e-4 m and the delay by ~1e-9 m
        np.testing.assert_allclose(wet, ow[0], rtol=0, atol=5 * TIGHT, equal_nan=True)
        np.testing.assert_allclose(hyd, oh[0], rtol=0, atol=5 * TIGHT, equal_nan=True)


@pytest.mark.parametrize('case', ['steep_80km', 'arctic_80km', 'polar_80km', 'nonunit_los', 'jittered_axes', 'mixed_axes', 'negative_ht'])
def test_ray_polynomial_stress(R, case):
    """Corner cases of the ray-polynomial kernels against the oracle: the longest rays the static classification admits
    (52-62 deg incidence through an 80 km cube: the classification cuts at ~60 deg there, so both the light and the
    generic kernels take part), look vectors that are not unit length (the ray
    parameter is then not metres), x/y axes that are only NEARLY uniform (LDS guess-and-verify instead of the arithmetic
    cell) and origins below the ellipsoid."""
    rng = np.random.default_rng(11)
    if case == 'steep_80km':
        c = O.synthetic_cube(48, 80, 40, seed=3, ztop=80000.0, y0=20.0, y1=34.0, x0=-125.0, x1=-100.0)
        ypts = np.linspace(28.0, 26.5, 9); xpts = np.linspace(-114.0, -111.0, 11)
        inc = rng.uniform(52, 62, (9, 11)); hd = rng.uniform(-180, 180, (9, 11)); ht = 0.0
    elif case == 'arctic_80km':       # a real-model-sized top (80 km) at 69-75 deg N: up to 0.07 rad of longitude travel on the light path
        c = O.synthetic_cube(60, 90, 40, seed=8, ztop=80000.0, y0=64.0, y1=79.0, x0=-175.0, x1=-130.0)
        ypts = np.linspace(75.0, 69.0, 10); xpts = np.linspace(-158.0, -150.0, 12)
        inc = rng.uniform(25, 45, (10, 12)); hd = rng.uniform(-180, 180, (10, 12)); ht = 0.0
    elif case == 'polar_80km':        # 79-85.5 deg N: up to the classification's 0.2 rad of longitude travel (the hand-over to the generic
        c = O.synthetic_cube(90, 120, 40, seed=8, ztop=80000.0, y0=70.0, y1=89.5, x0=-175.0, x1=-95.0)       # kernels sits near 84.5 deg N here)
        ypts = np.linspace(85.5, 79.0, 14); xpts = np.linspace(-150.0, -120.0, 10)
        inc = rng.uniform(25, 46, (14, 10)); hd = rng.uniform(-180, 180, (14, 10)); ht = 0.0
    else:
        c = O.synthetic_cube(40, 44, 36, seed=5, y0=30.0, y1=38.0, x0=-122.0, x1=-110.0)
        ypts = np.linspace(35.0, 33.0, 12); xpts = np.linspace(-117.5, -114.5, 14)
        inc = rng.uniform(10, 50, (12, 14)); hd = rng.uniform(-180, 180, (12, 14)); ht = -60.0 if case == 'negative_ht' else 250.0
    if case == 'jittered_axes':
        c['xs'] = c['xs'] + 1e-7 * rng.uniform(-1, 1, c['xs'].size)
        c['ys'] = c['ys'] + 1e-7 * rng.uniform(-1, 1, c['ys'].size)
    if case == 'mixed_axes':          # x exactly uniform (index-space polynomial), y only nearly (LDS table): both in one kernel
        c['ys'] = c['ys'] + 1e-7 * rng.uniform(-1, 1, c['ys'].size)
    zref = float(c['zs'].max() - 1)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro']))
    scale = rng.uniform(0.8, 1.2, inc.shape) if case == 'nonunit_los' else 1.0      # (the reference's own iteration diverges beyond |l| cos(inc) = 2)
    look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(inc, hd, llh[1], llh[0], llh[2]) * np.asarray(scale)[..., None]
    (ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([ht]), look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True)
    xx, yy = np.meshgrid(xpts, ypts)
    los = look(ht, [xx, yy, np.full(yy.shape, ht)], None, yy)
    wet, hyd, nparts, _ = cube.raytrace(R.Rays.grid(xpts, ypts, los=np.ascontiguousarray(los)), ht, zref)
    assert np.array_equal(nparts, onp[0])
    assert np.array_equal(np.isnan(wet), np.isnan(ow[0])) and np.isfinite(ow[0]).mean() > 0.5
    tol = 2e-8 if case == 'steep_80km' else 5 * TIGHT       # 5 m of slant delay, and the reference's own 3-step crossings are cm off there
    np.testing.assert_allclose(wet, ow[0], rtol=0, atol=tol, equal_nan=True)
    np.testing.assert_allclose(hyd, oh[0], rtol=0, atol=tol, equal_nan=True)


def test_diverged_lengths_are_refused(R, c1):
    """Look vectors 3x unit length make getTopOfAtmosphere's fixed-point iteration diverge (|1 - |l| cos(inc)| > 1, in the
    reference as well): the per-level maximum length is astronomically large.  The library must not try to integrate
    ~1e300 parts: synchronous calls raise, asynchronous ones return NaN."""
    cube = R.Cube(c1['ys'], c1['xs'], c1['zs'], c1['wet'], c1['hydro'], order='zyx')
    xp = np.linspace(-118.0, -116.0, 9); yp = np.linspace(34.0, 33.0, 7)
    los = 3.0 * R.Rays.grid(xp, yp, inc=20.0, hd=-167.9).look_vectors()
    zref = float(c1['zs'].max() - 1)
    with pytest.raises(Exception, match='diverged'):
        cube.raytrace(R.Rays.grid(xp, yp, los=np.ascontiguousarray(los)), 0.0, zref)
    import torch
    dev = torch.device('cuda:0')
    rays = R.Rays.grid(torch.from_numpy(xp).to(dev), torch.from_numpy(yp).to(dev), los=torch.from_numpy(np.ascontiguousarray(los)).to(dev))
    wet, hyd, _, _ = cube.raytrace(rays, 0.0, zref, want_nparts=False)         # device arrays, no host sync: NaN outputs
    torch.cuda.synchronize()
    assert torch.isnan(wet).all() and torch.isnan(hyd).all()


def test_generic_ray_side_buffer_full_partial_and_absent(R):
    """The level crossings of generic-geodesy rays travel from pass 1 to pass 2 in a compact side buffer; a ray that finds it
    full has them recomputed by pass 2.  A polar scene (every ray generic) and a scene straddling the 84.4 deg hand-over
    (mixed) must give bit-identical delays with an ample buffer, one that holds a fraction of the generic rays, and none."""
    ctx = R.Context.default()
    c = O.synthetic_cube(40, 44, 36, seed=7, y0=78.0, y1=89.9, x0=-60.0, x1=60.0)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    zref = c['zs'].max() - 1
    rng = np.random.default_rng(5)
    for ypts in (np.linspace(89.2, 86.0, 40), np.linspace(87.0, 80.0, 40)):
        xpts = np.linspace(-20.0, 20.0, 50)
        inc = rng.uniform(15, 60, (40, 50)); hd = rng.uniform(-180, 180, (40, 50))
        res = []
        try:
            for cap in (-1, 300, 0):
                ctx.set_side_capacity(cap)
                res.append(cube.raytrace(R.Rays.grid(xpts, ypts, inc=inc, hd=hd), 0.0, zref))
        finally:
            ctx.set_side_capacity(-1)
        assert np.isfinite(res[0][0]).mean() > 0.5
        for w, h, npz, _ in res[1:]:
            assert np.array_equal(npz, res[0][2])
            assert np.array_equal(w, res[0][0], equal_nan=True) and np.array_equal(h, res[0][1], equal_nan=True)


def test_randomised_sweep():
    """tools/fuzz_parity.py (random cubes, axes kinds incl. a descending latitude axis, scenes partly outside the cube, heights,
    integration tops, incidence to 70 deg, NaN look vectors, segment lengths, LCC model grids): 500 trials here; 3750 trials over 10
    seeds were run in round 1 - no mismatch in nParts, NaN pattern, error behaviour; worst |delay difference| 1.1e-9 m."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / 'tools' / 'fuzz_parity.py'), '500', '21'], capture_output=True, text=True, timeout=1200)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    res = json.loads(lines[0])
    assert res['stats']['trials'] == 500 and res['stats']['rays'] > 10000
    assert res['n_bad'] == 0, lines[1:6]
    assert max(res['worst_abs_m'].values()) < 5e-9


def test_randomised_sweep_other_entry_points():
    """tools/fuzz_natives.py: the zenith cube, station queries, `interpolate` 1-3 D, `interpolate_along_axis` and `makePoints0D..3D`
    over random grids (exact / jittered / irregular / descending axes, NaN cells), queries outside / on the last node / NaN, fill
    values: 120 trials here (3650 when written: bit-exact natives and makePoints, <= 6e-16 relative for the scipy-RGI gathers).  When
    the reference's own extensions are there as binaries (oracle/_ref, built by oracle/build_ref.sh) the GPU results are compared
    with THEM as well, bit for bit."""
    import json
    import subprocess
    import sys
    from pathlib import Path
    root = Path(__file__).resolve().parent.parent
    out = subprocess.run([sys.executable, str(root / 'tools' / 'fuzz_natives.py'), '120', '5'], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [ln for ln in out.stdout.splitlines() if ln.startswith('{')]
    res = json.loads(lines[0])
    assert res['trials'] == 120 and res['n_bad'] == 0, lines[1:6]
    assert max(res['worst_rel'].values()) < 1e-13


def test_survey_sanity_numbers_slant_to_zenith_ratio(R):
    """SURVEY.md 8(a) / 8(d), measured there with the reference-faithful probe: on the 80-level synthetic profile (laterally uniform
    here) a 35 deg ray from ht = 0 takes S = 169 samples and its hydrostatic delay is 1.22005 x the zenith delay - below
    1/cos(35 deg) = 1.22077 because the Earth curves away under the ray.  The ray tracer reproduces both numbers."""
    ys = np.linspace(30, 36, 300); xs = np.linspace(-121, -113, 300); zs = np.round(-100 + 41000 * np.linspace(0, 1, 80) ** 2, 3)
    hyd = np.broadcast_to((270.0 * np.exp(-zs / 8000.0)).astype(np.float32)[:, None, None], (80, 300, 300)).copy()
    wet = np.broadcast_to((60.0 * np.exp(-zs / 2000.0)).astype(np.float32)[:, None, None], (80, 300, 300)).copy()
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx')
    xp = np.linspace(-118.0, -117.0, 5); yp = np.linspace(33.5, 32.5, 5)
    zref = float(zs.max() - 1)
    wz, hz, nz_, _ = cube.raytrace(R.Rays.grid(xp, yp, zenith=True), 0.0, zref)
    for hd in (-167.9, 0.0, 90.0):
        w, h, n, _ = cube.raytrace(R.Rays.grid(xp, yp, inc=35.0, hd=hd), 0.0, zref)
        assert int(n.sum()) == 169
        ratio = h / hz
        assert np.all(np.abs(ratio - 1.22005) < 1e-5) and np.all(ratio < 1 / np.cos(np.radians(35.0)))
    # zenith rays integrate the profile itself: the trapezoid of the (piecewise linear) refractivity over [0, zref]
    f = 270.0 * np.exp(-zs / 8000.0)
    zz = np.concatenate([[0.0], zs[zs > 0]]); zz[-1] = zref
    prof = np.interp(zz, zs, f.astype(np.float32).astype(np.float64))
    assert abs(hz.mean() - 1e-6 * np.sum(0.5 * (prof[1:] + prof[:-1]) * np.diff(zz))) < 1e-9


@pytest.mark.parametrize('case', ['east_of_cube_end', 'west_side', 'global_axis', 'stere_across_dateline'])
def test_dateline_scenes(R, case):
    """Scenes at the +-180 deg meridian.  The light ray path carries longitude unwrapped, the reference wraps every sample
    (atan2): they agree when a crossing sample is outside the cube either way (a regional lon/lat cube ending at the meridian:
    NaN in both), which is what lets such scenes stay on the light path; a cube whose axis spans both ends (global model) keeps
    the generic kernels, where rays really do re-enter at the other end; on a polar-stereographic (HRRR-AK) grid the geographic
    dateline is an ordinary meridian and rays cross it with finite delays.  All against the oracle, NaN patterns included."""
    rng = np.random.default_rng(8)
    ny, nx = 14, 40
    proj = None
    if case == 'east_of_cube_end':
        c = O.synthetic_cube(30, 60, 30, seed=5, y0=50.0, y1=58.0, x0=172.0, x1=179.99)
        ypts = np.linspace(56.0, 52.0, ny); xpts = np.linspace(178.6, 179.97, nx)
    elif case == 'west_side':
        c = O.synthetic_cube(30, 60, 30, seed=5, y0=-20.0, y1=-12.0, x0=-179.99, x1=-172.0)
        ypts = np.linspace(-14.0, -18.0, ny); xpts = np.linspace(-179.97, -178.6, nx)
    elif case == 'global_axis':
        c = O.synthetic_cube(24, 361, 24, seed=5, y0=50.0, y1=58.0, x0=-180.0, x1=180.0)
        ypts = np.linspace(56.0, 52.0, ny); xpts = np.linspace(178.6, 179.97, nx)
    else:
        par = dict(lat_0=90.0, lat_ts=60.0, lon_0=225.0, a=6371229.0, es=0.0)
        c = O.synthetic_cube(50, 50, 30, seed=5, ztop=26000.0)
        cx, cy = O.stere_forward(52.0, 180.0, **par)
        c['xs'] = cx + 6000.0 * (np.arange(50) - 25); c['ys'] = cy + 6000.0 * (np.arange(50) - 25)
        proj = dict(par, proj='stere')
        ypts = np.linspace(52.8, 51.2, ny); xpts = np.concatenate([np.linspace(178.9, 179.98, nx // 2), np.linspace(-179.98, -178.9, nx // 2)])
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    if proj is not None:
        cube.set_projection_stere(**par)
    ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro']))
    inc = rng.uniform(25, 50, (ny, nx)); hd = rng.choice([-167.9, -12.1, 90.0, -90.0], (ny, nx))       # incl. due east / west looks
    zref = c['zs'].max() - 1
    look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(inc, hd, llh[1], llh[0], llh[2])
    (ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([0.0]), look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True, model_proj=proj)
    wet, hyd, nparts, _ = cube.raytrace(R.Rays.grid(xpts, ypts, inc=inc, hd=hd), 0.0, zref)
    assert np.array_equal(nparts, onp[0])
    assert np.array_equal(np.isnan(wet), np.isnan(ow[0])) and np.array_equal(np.isnan(hyd), np.isnan(oh[0]))
    generic = cube.ctx.generic_ray_count()
    assert (generic == ny * nx) if case == 'global_axis' else (generic == 0), generic          # which kernels did the work
    if case in ('east_of_cube_end', 'west_side'):
        assert 0.02 < np.isnan(oh[0]).mean() < 0.9            # some rays leave through the meridian, most do not
    else:
        assert np.isfinite(oh[0]).mean() > 0.95               # rays crossing the dateline keep finite delays
    np.testing.assert_allclose(wet, ow[0], rtol=0, atol=5 * TIGHT, equal_nan=True)
    np.testing.assert_allclose(hyd, oh[0], rtol=0, atol=5 * TIGHT, equal_nan=True)


def test_point_index_gives_the_same_bits(R):
    """Large random point sets on a cube beyond the caches read one 128 B line per point from the corner-quad copy instead of four from
    the (y,x,z) cube (cube_kernels.h).  Same cell search, same weights, same summation order: the values must be IDENTICAL - f32 and f64
    cubes, level counts that do and do not fill the last block, points on the last nodes / outside / NaN, descending axes."""
    rng = np.random.default_rng(21)
    for dt, shape in ((np.float32, (37, 41, 23)), (np.float32, (12, 9, 20)), (np.float32, (5, 6, 2)), (np.float64, (17, 19, 11)), (np.float64, (6, 5, 2))):
        ny, nx, nz = shape
        ys = np.sort(rng.uniform(-3, 9, ny))[::-1].copy(); xs = np.linspace(-7.0, 4.0, nx); zs = np.sort(rng.uniform(0, 9000, nz))
        w = rng.normal(size=shape).astype(dt); h = rng.normal(size=shape).astype(dt)
        cube = R.Cube(ys, xs, zs, w, h, order='yxz')
        n = 20000
        q = np.stack([rng.uniform(ys.min() - 0.2, ys.max() + 0.2, n), rng.uniform(-7.3, 4.3, n), rng.uniform(zs[0] - 50, zs[-1] + 50, n)], -1)
        q[:7] = [[ys[0], xs[0], zs[0]], [ys[-1], xs[-1], zs[-1]], [ys[1], xs[2], zs[-1]], [ys[-1], xs[3], zs[0]], [np.nan, 0, 100], [0, 0, zs[-1] + 1e-9], [ys[2], xs[-1], zs[nz // 2]]]
        a_w, a_h = cube.interp(q)
        assert cube.ctx.lib.rdr_cube_point_index_bytes(cube.handle) == 0
        nbytes = cube.point_index()
        cpb = 3 if dt == np.float32 else 1
        assert nbytes == (ny - 1) * (nx - 1) * -(-(nz - 1) // cpb) * 128
        b_w, b_h = cube.interp(q)
        assert np.array_equal(a_w, b_w, equal_nan=True) and np.array_equal(a_h, b_h, equal_nan=True)
        assert np.isnan(a_w[4]) and np.isnan(a_w[5]) and np.isfinite(a_w[:4]).all() and 0.2 < np.isfinite(a_w).mean() < 1.0
        assert cube.point_index(build=False) == 0
        c_w, _ = cube.interp(q)
        assert np.array_equal(a_w, c_w, equal_nan=True)


def test_point_index_is_built_by_the_second_large_call(R):
    import torch
    rng = np.random.default_rng(3)
    ny, nx, nz = 520, 500, 110                                             # 229 MB as float2: beyond the 192 MB threshold (round 5: smaller cubes live in the Infinity Cache,
                                                                            # where the copy LOSES - measured on the 38.8 MB intermediate cube of configs[1])
    ys = np.linspace(30, 40, ny); xs = np.linspace(-120, -110, nx); zs = np.round(-100 + 30000 * np.linspace(0, 1, nz) ** 2, 3)
    dev = torch.device('cuda:0')
    w = torch.randn((ny, nx, nz), dtype=torch.float32, device=dev); h = torch.randn_like(w)
    cube = R.Cube(ys, xs, zs, w, h, order='yxz')
    held = lambda: cube.ctx.lib.rdr_cube_point_index_bytes(cube.handle)
    small = torch.from_numpy(np.stack([rng.uniform(30, 40, 1000), rng.uniform(-120, -110, 1000), rng.uniform(0, 9000, 1000)], -1)).to(dev)
    n = 300000
    big = torch.from_numpy(np.stack([rng.uniform(30, 40, n), rng.uniform(-120, -110, n), rng.uniform(0, 9000, n)], -1)).to(dev)
    cube.interp(small); cube.interp(small); cube.interp(small)
    assert held() == 0                                                       # small point sets never trigger it
    r1 = cube.interp(big)
    assert held() == 0                                                       # a cube queried once never pays for the copy
    r2 = cube.interp(big)
    assert held() == (ny - 1) * (nx - 1) * -(-(nz - 1) // 3) * 128
    r3 = cube.interp(small)
    torch.cuda.synchronize()
    assert torch.equal(r1[0], r2[0]) and torch.equal(r1[1], r2[1]) and torch.isfinite(r3[0]).all()
    # round 4: a point set so large that the build pays for itself WITHIN the call (by time, from the measured rates: n x 175 B > the
    # copy's 1.23 GB here -> from 7.0 M points on) builds it at the FIRST call; same bits as the direct gather
    cube2 = R.Cube(ys, xs, zs, w, h, order='yxz')
    held2 = lambda: cube2.ctx.lib.rdr_cube_point_index_bytes(cube2.handle)
    n2 = 7500000
    huge = torch.from_numpy(np.stack([rng.uniform(30, 40, n2), rng.uniform(-120, -110, n2), rng.uniform(0, 9000, n2)], -1)).to(dev)
    direct = cube2.interp(huge[:200000].contiguous())
    assert held2() == 0
    q1 = cube2.interp(huge)
    assert held2() == held()
    torch.cuda.synchronize()
    assert torch.equal(q1[0][:200000], direct[0]) and torch.equal(q1[1][:200000], direct[1])


def test_f64_cube_marcher_lds_staging_gives_the_f32_kernels_bits(R):
    """Round 4: on f64 cubes the light marcher stages every level's footprint of a wave (4 x 4 columns x 3 z entries) in LDS and reads
    the samples' corners from there; samples whose cells leave the block take the direct gathers.  An f64 cube holding f32 values
    widened gives the f32 kernel the SAME operands (its f32 -> f64 conversions are exact), so the two instantiations must agree
    bit for bit - on a fine scene (staging engaged everywhere), a scene hugging the cube's corner (block origin clamped at the
    first / last columns), a coarse scene (a wave spans several cells: staging switches itself off) and across the level count."""
    import torch
    from raider_amd.synthetic import synthetic_cube
    dev = torch.device('cuda:0')
    c = synthetic_cube(60, 64, 48, seed=5)
    w32 = torch.from_numpy(c['wet']).to(dev); h32 = torch.from_numpy(c['hydro']).to(dev)
    c32 = R.Cube(c['ys'], c['xs'], c['zs'], w32, h32, order='zyx')
    c64 = R.Cube(c['ys'], c['xs'], c['zs'], w32.double(), h32.double(), order='zyx')
    assert c64.dtype == np.float64 and c32.dtype == np.float32
    zref = float(c['zs'].max() - 1)
    dy = c['ys'][1] - c['ys'][0]; dx = c['xs'][1] - c['xs'][0]
    scenes = {
        'fine': (np.linspace(-118.0, -117.6, 400), np.linspace(33.4, 33.1, 320)),                                   # 0.001 deg pixels: a wave sits in 1-2 cells
        'corner': (np.linspace(c['xs'][0] + 0.02 * dx, c['xs'][0] + 1.9 * dx, 256), np.linspace(c['ys'][-1] - 0.03 * dy, c['ys'][-1] - 2.2 * dy, 192)),
        'coarse': (np.linspace(-119.5, -115.5, 300), np.linspace(34.5, 31.5, 280)),                                   # 0.013 deg pixels: several cells per wave
    }
    for name, (xp, yp) in scenes.items():
        for inc, hd in ((39.0, -167.9), (20.0, 12.0)):
            for ht in (0.0, 1500.0):
                a = c32.raytrace(R.Rays.grid(xp, yp, inc=inc, hd=hd), ht, zref)
                b = c64.raytrace(R.Rays.grid(xp, yp, inc=inc, hd=hd), ht, zref)
                assert np.array_equal(a[2], b[2]), (name, inc, ht)
                assert np.array_equal(a[0], b[0], equal_nan=True) and np.array_equal(a[1], b[1], equal_nan=True), (name, inc, ht)
                assert np.isfinite(a[1]).mean() > (0.3 if name == 'corner' else 0.9)


def test_origin_above_zref_walks_its_one_segment_downwards(R):
    """Round 6, found by new fuzz seeds in a path round 5 already shipped: an origin ABOVE zref but inside zref's own model interval gets ONE
    segment from the reference's level tests (losreader.py:785-808: low_ht = ht > high_ht = zref, |high - low| >= 1 m) - walked downwards, with
    a POSITIVE length (np.linalg.norm, losreader.py:821) and hence positive trapezoid weights and nParts = ceil(|L| / 1000) + 1.  The kernels took
    the signed difference of the two crossings: negative weights (delays of the wrong sign) and a zero slice maximum.  Slices, per-pixel heights
    straddling zref, point lists, the generic-geodesy kernels (polar scene) and both cube dtypes against the oracle; an origin above the
    interval's top node still has no level at all."""
    from oracle import oracle_c as OC
    c = O.synthetic_cube(18, 17, 6, seed=5, ztop=15000.0)             # zs = -100, 500, 2300, 5300, 9500, 14900
    assert list(c['zs']) == [-100.0, 500.0, 2300.0, 5300.0, 9500.0, 14900.0]
    zref = 8940.0
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro']))
    xpts = np.linspace(c['xs'][5], c['xs'][11], 19); ypts = np.linspace(c['ys'][12], c['ys'][5], 14)
    xx, yy = np.meshgrid(xpts, ypts)
    for inc in (12.0, 38.0, 61.0):
        look = lambda ht, llh, xyz, yy_: O.look_vectors_from_inc_hd(np.full(yy_.shape, inc), np.full(yy_.shape, -167.9), llh[1], llh[0], llh[2])
        for ht in (8000.0, 8939.5, 8941.5, 9000.0, 9400.0, 9499.0):
            (ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([ht]), look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True)
            rays = R.Rays.grid(xpts, ypts, inc=inc, hd=-167.9)
            if onp[0] is None:                                         # |zref - ht| < 1 m: no level, zero delay (a top slice)
                with pytest.raises(R.NoLevels):
                    cube.raytrace(rays, ht, zref)
                continue
            wet, hyd, nparts, _ = cube.raytrace(rays, ht, zref)
            assert np.array_equal(nparts, onp[0]) and (hyd > 0).all() and (oh[0] > 0).all(), (inc, ht, nparts, onp[0])
            np.testing.assert_allclose(wet, ow[0], rtol=0, atol=TIGHT)
            np.testing.assert_allclose(hyd, oh[0], rtol=0, atol=TIGHT)
            # the same rays as a point list with ECEF look vectors
            los = O.look_vectors_from_inc_hd(np.full(yy.shape, inc), np.full(yy.shape, -167.9), yy, xx, ht)
            pw, ph, pn, _ = cube.raytrace(R.Rays.points(lat=yy.ravel().copy(), lon=xx.ravel().copy(), los=np.ascontiguousarray(los).reshape(-1, 3)), ht, zref)
            assert np.array_equal(pn, nparts)
            np.testing.assert_allclose(ph.reshape(yy.shape), oh[0], rtol=0, atol=TIGHT)
    with pytest.raises(R.NoLevels):                                     # above the interval's top node: zref's interval is skipped (high_ht < ht), the next starts above zref
        cube.raytrace(R.Rays.grid(xpts, ypts, inc=38.0, hd=-167.9), 9600.0, zref)
    # a long reversed segment: the slice maximum is the REVERSED rays' length (nParts = 3 at 61 deg from 9499 m with 400 m segments)
    look = lambda ht, llh, xyz, yy_: O.look_vectors_from_inc_hd(np.full(yy_.shape, 61.0), np.full(yy_.shape, -167.9), llh[1], llh[0], llh[2])
    (ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([9499.0]), look, ip, MAX_SEGMENT_LENGTH=400.0, MAX_TROPO_HEIGHT=zref, return_nparts=True)
    wet, hyd, nparts, _ = cube.raytrace(R.Rays.grid(xpts, ypts, inc=61.0, hd=-167.9), 9499.0, zref, max_seg=400.0)
    assert np.array_equal(nparts, onp[0]) and nparts[0] >= 4
    np.testing.assert_allclose(hyd, oh[0], rtol=0, atol=TIGHT)
    # per-pixel heights on both sides of zref (the fuzz case): against the C oracle's per-ray rule
    rng = np.random.default_rng(8)
    hts = rng.uniform(5400.0, 9490.0, yy.shape)
    hts[0, :4] = [8939.7, 8940.4, 8941.2, 9499.9]
    los = O.look_vectors_from_inc_hd(np.full(yy.shape, 33.0), np.full(yy.shape, -167.9), yy, xx, hts)
    cc = dict(c, ys=c['ys'], wet=c['wet'], hydro=c['hydro'])
    qw, qh, qnp = OC.build_cube_ray_per_pixel(cc, yy, xx, hts, los, zref, max_seg=700.0)
    pw, ph, pnp, _ = cube.raytrace(R.Rays.grid(xpts, ypts, los=np.ascontiguousarray(los), hts=hts), None, zref, 700.0)
    kzt = cube.ray_levels(float(hts.min()), zref)[2]
    assert np.array_equal(pnp, qnp[kzt]) and (hts > zref).sum() > 20
    np.testing.assert_allclose(pw, qw, rtol=0, atol=TIGHT)
    np.testing.assert_allclose(ph, qh, rtol=0, atol=TIGHT)
    # the generic-geodesy kernels (a polar scene: every ray is classified generic) and an f64 cube
    cp = O.synthetic_cube(12, 40, 6, seed=6, ztop=15000.0, y0=86.0, y1=89.9, x0=-60.0, x1=60.0)
    for dt_ in (np.float32, np.float64):
        cubep = R.Cube(cp['ys'], cp['xs'], cp['zs'], cp['wet'].astype(dt_), cp['hydro'].astype(dt_), order='zyx')
        ipp = list(O.getInterpolators(cp['xs'], cp['ys'], cp['zs'], cp['wet'].astype(dt_), cp['hydro'].astype(dt_)))
        xp = np.linspace(-20.0, 20.0, 9); yp = np.linspace(88.9, 88.0, 7)
        look = lambda ht, llh, xyz, yy_: O.look_vectors_from_inc_hd(np.full(yy_.shape, 30.0), np.full(yy_.shape, -167.9), llh[1], llh[0], llh[2])
        (ow, oh), onp = O.build_cube_ray(xp, yp, np.array([9300.0]), look, ipp, MAX_TROPO_HEIGHT=zref, return_nparts=True)
        wet, hyd, nparts, _ = cubep.raytrace(R.Rays.grid(xp, yp, inc=30.0, hd=-167.9), 9300.0, zref)
        assert np.array_equal(nparts, onp[0]) and (cubep.ctx.generic_ray_count() > 0 or dt_ is np.float64)
        np.testing.assert_allclose(wet, ow[0], rtol=0, atol=5 * TIGHT)
        np.testing.assert_allclose(hyd, oh[0], rtol=0, atol=5 * TIGHT)
