"""CPU: the interpolant the whole path rests on is scipy's RegularGridInterpolator (delayFcns.py:55-56); the reference does not pin scipy
(environment.yml).  The oracle restates the 1.15.3 algorithm - here it is run against the scipy of THIS interpreter and, when the image
has one, against a much older scipy (1.7.1, a pure-NumPy implementation of the same interpolant, in /opt/conda's Anaconda): all three
must agree bit for bit, NaN masks included, so the parity target does not depend on the scipy a RAiDER installation happens to have."""
import os
import subprocess

import numpy as np
import pytest

from oracle import raider_oracle as O

OLD_PY = '/opt/conda/bin/python3.9'


def _points(c, n=100000):
    rng = np.random.default_rng(0)
    pts = np.stack([rng.uniform(30, 36, n), rng.uniform(-121, -113, n), rng.uniform(-100, 40900, n)], -1)
    pts[:100, 0] = c['ys'][-1]; pts[100:200, 2] = c['zs'][-1]; pts[200:300, 1] = c['xs'][0]       # ON the edges: inside
    pts[300:310, 0] = 37.0; pts[310:320, 2] = -100.001; pts[320:330, 1] = np.nan                  # outside / NaN: NaN
    return pts


def test_oracle_rgi_equals_this_scipy_bit_for_bit():
    import scipy
    from scipy.interpolate import RegularGridInterpolator as RGI
    c = O.synthetic_cube(50, 50, 40, seed=0)
    pts = _points(c)
    iw, ih = O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro'])
    for mine, field in ((iw, 'wet'), (ih, 'hydro')):
        ref = RGI((c['ys'], c['xs'], c['zs']), c[field].transpose(1, 2, 0), bounds_error=False, fill_value=np.nan)(pts)
        got = mine(pts)
        assert np.array_equal(got, ref, equal_nan=True), (scipy.__version__, field)
        assert np.isnan(got[300:330]).all() and np.isfinite(got[:300]).all()


def test_an_old_scipy_gives_the_same_bits(tmp_path):
    if not os.path.exists(OLD_PY):
        pytest.skip('no second interpreter in this image')
    c = O.synthetic_cube(50, 50, 40, seed=0)
    pts = _points(c)
    np.savez(tmp_path / 'in.npz', ys=c['ys'], xs=c['xs'], zs=c['zs'], wet=c['wet'].transpose(1, 2, 0), hydro=c['hydro'].transpose(1, 2, 0), pts=pts)
    script = ("import sys, numpy as np\ntry:\n    import scipy\n    from scipy.interpolate import RegularGridInterpolator as RGI\nexcept Exception:\n    sys.exit(77)\n"
              "d = np.load(sys.argv[1])\n"
              "out = {k: RGI((d['ys'], d['xs'], d['zs']), d[k], bounds_error=False, fill_value=np.nan)(d['pts']) for k in ('wet', 'hydro')}\n"
              "np.savez(sys.argv[2], version=scipy.__version__, **out)\n")
    r = subprocess.run([OLD_PY, '-W', 'ignore', '-c', script, str(tmp_path / 'in.npz'), str(tmp_path / 'out.npz')], capture_output=True, text=True, timeout=600,
                       env={k: v for k, v in os.environ.items() if not k.startswith('PYTHON')})
    if r.returncode == 77:
        pytest.skip('no scipy in the second interpreter')
    assert r.returncode == 0, r.stderr[-2000:]
    old = np.load(tmp_path / 'out.npz')
    import scipy
    if str(old['version']) == scipy.__version__:
        pytest.skip('the second interpreter has the same scipy')
    iw, ih = O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro'])
    assert np.array_equal(iw(pts), old['wet'], equal_nan=True) and np.array_equal(ih(pts), old['hydro'], equal_nan=True), str(old['version'])
