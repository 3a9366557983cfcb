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
