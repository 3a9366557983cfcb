"""CPU: the C/OpenMP restatement (oracle/oracle_c.c, the multi-core CPU baseline of bench.py) against the reference
goldens and the NumPy oracle."""
import numpy as np
import pytest

from oracle import raider_oracle as O
from oracle import oracle_c as OC


@pytest.fixture(scope='module')
def c1():
    return O.synthetic_cube(50, 50, 40, seed=0)


def _los(xpts, ypts, inc, hd, ht):
    xx, yy = np.meshgrid(xpts, ypts)
    return O.look_vectors_from_inc_hd(np.broadcast_to(np.asarray(inc, float), yy.shape), np.full(yy.shape, hd), yy, xx, ht)


def test_c_oracle_vs_reference_goldens(golden, c1):
    g = golden('g5_build_cube_ray')
    zref = float(g['c1_zref'])
    for tag, inc in (('fixed', 39.0), ('pp', g['c1_pp_inc'])):
        for i, ht in enumerate(g['c1_zpts']):
            w, h, npn = OC.build_cube_ray_slice(c1, g['c1_xpts'], g['c1_ypts'], float(ht), _los(g['c1_xpts'], g['c1_ypts'], inc, -167.9, ht), zref)
            assert np.array_equal(npn, g[f'c1_{tag}_nparts{i}'])
            np.testing.assert_allclose(w, g[f'c1_{tag}_wet'][i], rtol=0, atol=1e-11)
            np.testing.assert_allclose(h, g[f'c1_{tag}_hydro'][i], rtol=0, atol=1e-11)
    # lateral exits -> the same NaN mask
    w, h, npn = OC.build_cube_ray_slice(c1, g['c1_edge_xpts'], g['c1_edge_ypts'], 0.0, _los(g['c1_edge_xpts'], g['c1_edge_ypts'], 45.0, -167.9, 0.0), zref)
    assert np.array_equal(npn, g['c1_edge_nparts0'])
    np.testing.assert_allclose(w, g['c1_edge_wet'][0], rtol=0, atol=1e-11, equal_nan=True)
    # other MAX_SEGMENT_LENGTH / zref
    w, h, npn = OC.build_cube_ray_slice(c1, g['c1_xpts'], g['c1_ypts'], 100.0, _los(g['c1_xpts'], g['c1_ypts'], 20.0, -12.1, 100.0), 26000.0, max_seg=500.0)
    assert np.array_equal(npn, g['c1_z26_nparts0'])
    np.testing.assert_allclose(h, g['c1_z26_hydro'][0], rtol=0, atol=1e-11)


def test_c_oracle_whole_slice_partition(golden, c1):
    g = golden('g5b_whole_vs_halves')
    zref = float(g['zref'])
    xp, yp, inc = g['xpts'], g['ypts'], g['inc']
    w, h, npn = OC.build_cube_ray_slice(c1, xp[:32], yp, 0.0, _los(xp[:32], yp, inc[:, :32], -167.9, 0.0), zref, nparts=g['nparts'])
    np.testing.assert_allclose(h, g['hydro'][0][:, :32], rtol=0, atol=1e-11)
    assert OC.num_threads() >= 1


def test_per_pixel_height_rule_reduces_to_the_slice_algorithm(golden, c1):
    """Rays with their own origin heights have no reference semantics; the stated rule must BE the
    reference's slice algorithm when all heights are equal - pinned here on the reference goldens g5, for the NumPy restatement
    (built ray by ray from the pinned slice functions) and for the C one."""
    g = golden('g5_build_cube_ray')
    zref = float(g['c1_zref'])
    xp, yp = g['c1_xpts'][:12], g['c1_ypts'][:10]
    xx, yy = np.meshgrid(xp, yp)
    ip = list(O.getInterpolators(c1['xs'], c1['ys'], c1['zs'], c1['wet'], c1['hydro']))
    for i, ht in enumerate(g['c1_zpts']):
        los = _los(xp, yp, 39.0, -167.9, ht)
        ws, hs, nps = OC.build_cube_ray_slice(c1, xp, yp, float(ht), los, zref)
        idx = [zz for zz, _, _ in O.ray_levels_idx(c1['zs'], float(ht), zref)]
        w, h, npp = OC.build_cube_ray_per_pixel(c1, yy, xx, np.full(yy.shape, float(ht)), los, zref)
        assert np.array_equal(npp[idx], nps) and not npp[[z for z in range(npp.size) if z not in idx]].any()
        assert np.array_equal(w, ws) and np.array_equal(h, hs)                          # same arithmetic, bit for bit
        wn, hn, npn = O.build_cube_ray_per_pixel(yy, xx, np.full(yy.shape, float(ht)), los, ip, MAX_TROPO_HEIGHT=zref)
        assert np.array_equal(npn, npp)
        np.testing.assert_allclose(wn.reshape(yy.shape), ws, rtol=0, atol=1e-12); np.testing.assert_allclose(hn.reshape(yy.shape), hs, rtol=0, atol=1e-12)
        # ... and with the whole-slice nParts of the golden it IS the golden (the sub-block's own maxima differ from the slice's)
        full = np.zeros(c1['zs'].size - 1, dtype=int); full[idx] = g[f'c1_fixed_nparts{i}']
        w2, h2, _ = OC.build_cube_ray_per_pixel(c1, yy, xx, np.full(yy.shape, float(ht)), los, zref, nparts=full)
        np.testing.assert_allclose(h2, g['c1_fixed_hydro'][i][:10, :12], rtol=0, atol=1e-11)


def test_per_pixel_heights_c_vs_numpy(c1):
    """Mixed heights (below the model, between nodes, ON a node, within 1 m of a node, above the integration top): the C restatement
    against the NumPy one, which applies the pinned build_ray to every ray with its own height."""
    rng = np.random.default_rng(5)
    n = 60
    lat = rng.uniform(31.0, 35.0, n); lon = rng.uniform(-120.0, -114.5, n)
    zs = c1['zs']
    hts = rng.uniform(-99.0, 4000.0, n)          # (an origin BELOW the model puts its first sample on the bottom node +- round-off: NaN or not is a coin toss in the reference too)
    hts[:6] = [zs[3], zs[4] - 0.4, zs[4] + 0.3, -99.5, zs[0] + 0.5, 36000.0]
    zref = 30000.0
    inc = rng.uniform(15.0, 55.0, n)
    los = O.look_vectors_from_inc_hd(inc, np.full(n, -167.9), lat, lon, hts)
    ip = list(O.getInterpolators(c1['xs'], c1['ys'], c1['zs'], c1['wet'], c1['hydro']))
    wn, hn, npn = O.build_cube_ray_per_pixel(lat, lon, hts, los, ip, MAX_TROPO_HEIGHT=zref)
    wc, hc, npc = OC.build_cube_ray_per_pixel(c1, lat, lon, hts, los, zref)
    assert np.array_equal(npn, npc)
    np.testing.assert_allclose(wc, wn, rtol=0, atol=1e-11); np.testing.assert_allclose(hc, hn, rtol=0, atol=1e-11)
    assert hc[5] == 0.0 and wc[5] == 0.0                       # origin above the integration top: no interval contributes
    assert hc[3] > hc[0] > 0                                   # a lower origin integrates more atmosphere
    # a different partition than any single slice would give: the per-level maximum comes from the rays that REACH the level
    lo_only = OC.build_cube_ray_per_pixel(c1, lat[6:], lon[6:], np.full(n - 6, hts[6:].min()), los[6:], zref)[2]
    assert (npc >= 0).all() and npc.max() >= 2 and lo_only.shape == npc.shape


@pytest.mark.parametrize('proj', [dict(lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5 - 360.0, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0),
                                  dict(lat_1=33.0, lat_2=45.0, lat_0=38.5, lon_0=-97.5, x_0=1000.0, y_0=-2000.0, a=6378137.0, es=0.0066943799901413165)])
def test_c_oracle_on_a_lambert_cube_vs_numpy(proj):
    """A weather cube on a Lambert-conformal-conic grid (HRRR's spherical cone; a two-parallel ellipsoidal one): every sample is projected
    to the model's x / y metres before the interpolation (delay.py:253,295).  C restatement == NumPy restatement (which the GPU
    parity tests of the conic path are written against), partition handed over and own."""
    rng = np.random.default_rng(4)
    lat = np.linspace(38.4, 37.6, 14); lon = np.linspace(-106.5, -104.9, 17)
    cx, cy = O.lcc_forward(*np.meshgrid(lat, lon, indexing='ij'), **proj)
    ny, nx, nz = 90, 110, 24
    ys = cy.min() - 60e3 + 3000.0 * np.arange(ny); xs = cx.min() - 60e3 + 3000.0 * np.arange(nx)
    assert ys[-1] > cy.max() + 30e3 and xs[-1] > cx.max() + 30e3
    zs = np.round(-100 + 26100 * np.linspace(0, 1, nz) ** 2, 3)
    z3 = zs[:, None, None]
    c = dict(ys=ys, xs=xs, zs=zs, wet=(60 * np.exp(-z3 / 2000) * (1 + 0.1 * rng.standard_normal((ny, nx))[None])).astype(np.float32),
             hydro=(270 * np.exp(-z3 / 8000) * (1 + 0.01 * rng.standard_normal((ny, nx))[None])).astype(np.float32))
    zref = float(zs.max() - 1)
    ip = list(O.getInterpolators(xs, ys, zs, c['wet'], c['hydro']))
    look = lambda ht, llh, xyz, yy: O.look_vectors_from_inc_hd(np.full(yy.shape, 36.0), np.full(yy.shape, -167.9), llh[1], llh[0], llh[2])
    for ht in (0.0, 1500.0):
        (rw, rh), onp = O.build_cube_ray(lon, lat, np.array([ht]), look, ip, MAX_TROPO_HEIGHT=zref, return_nparts=True, model_proj=proj)
        assert np.isfinite(rw).all()
        w, h, npn = OC.build_cube_ray_slice(c, lon, lat, ht, _los(lon, lat, 36.0, -167.9, ht), zref, model_proj=proj)
        assert np.array_equal(npn, onp[0])
        np.testing.assert_allclose(w, rw[0], rtol=0, atol=1e-11); np.testing.assert_allclose(h, rh[0], rtol=0, atol=1e-11)
    # outside the grid: NaN in both
    w, h, _ = OC.build_cube_ray_slice(c, lon - 5.0, lat, 0.0, _los(lon - 5.0, lat, 36.0, -167.9, 0.0), zref, model_proj=proj)
    assert np.isnan(w).all() and np.isnan(h).all()
