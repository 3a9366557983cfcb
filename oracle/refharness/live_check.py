#!/usr/bin/env python3
"""LIVE check of the oracle against the unmodified reference imported in place (ref_import.py; build container only): fresh random
cases, not the ones frozen in tests/golden/.  Run as a script (tests/test_oracle_vs_reference.py spawns it) so that the stub
packages the reference needs (pyproj / xarray / rasterio stand-ins) never enter the test process.  Prints one JSON line."""
import json
import sys
from pathlib import Path

import numpy as np

HERE = Path(__file__).resolve().parent
sys.path.insert(0, str(HERE))
import gen_golden as H      # noqa: E402  imports the reference in place (module level)
from oracle import raider_oracle as O   # noqa: E402

out = {}
worst = 0.0
for seed in (101, 102, 103, 104):
    rng = np.random.default_rng(seed)
    ny, nx, nz = int(rng.integers(20, 40)), int(rng.integers(20, 40)), int(rng.integers(15, 35))
    cube = O.synthetic_cube(ny, nx, nz, seed=seed, ztop=float(rng.choice([26000.0, 41000.0])))
    gy, gx = int(rng.integers(4, 9)), int(rng.integers(4, 9))
    xpts = np.linspace(-119.0, -115.0, gx); ypts = np.linspace(35.0, 31.0, gy)
    zpts = np.array([0.0, float(rng.uniform(300, 3000))])
    inc = rng.uniform(5, 60, (gy, gx)); hd = rng.uniform(-180, 180, (gy, gx))
    zref = float(cube['zs'].max() - 1)
    maxseg = float(rng.choice([1000.0, 600.0]))
    wet, hydro, nparts = H.run_ray(cube, xpts, ypts, zpts, inc, hd, zref, maxseg)
    ip = list(O.getInterpolators(cube['xs'], cube['ys'], cube['zs'], cube['wet'], cube['hydro']))
    look = lambda ht, llh, xyz, yy: O.look_vectors_from_inc_hd(inc, hd, llh[1], llh[0], llh[2])
    (ow, oh), onp = O.build_cube_ray(xpts, ypts, zpts, look, ip, MAX_SEGMENT_LENGTH=maxseg, MAX_TROPO_HEIGHT=zref, return_nparts=True)
    assert all(np.array_equal(a, b) for a, b in zip(nparts, onp)), 'nParts'
    assert np.isfinite(hydro).all()
    worst = max(worst, float(np.abs(ow - wet).max()), float(np.abs(oh - hydro).max()))
out['ray_max_abs_m'] = worst
# An origin ABOVE zref inside zref's own model interval (round 6): the reference's level tests leave ONE reversed segment (losreader.py:785-808),
# integrated with its positive length (np.linalg.norm, :821) - the quirk the GPU kernels got wrong until the fuzz drew it; an origin above the
# interval's top node has no level.  The unmodified reference against the oracle on exactly that.
cube = O.synthetic_cube(18, 17, 6, seed=5, ztop=15000.0)          # zs = -100, 500, 2300, 5300, 9500, 14900
xpts = np.linspace(cube['xs'][5], cube['xs'][11], 7); ypts = np.linspace(cube['ys'][12], cube['ys'][5], 6)
rng = np.random.default_rng(9)
inc = rng.uniform(5, 60, (6, 7)); hd = rng.uniform(-180, 180, (6, 7))
zpts = np.array([8000.0, 8941.5, 9000.0, 9400.0, 9499.0, 9600.0])
wet, hydro, nparts = H.run_ray(cube, xpts, ypts, zpts, inc, hd, 8940.0, 400.0)
ip = list(O.getInterpolators(cube['xs'], cube['ys'], cube['zs'], cube['wet'], cube['hydro']))
look = lambda ht, llh, xyz, yy: O.look_vectors_from_inc_hd(inc, hd, llh[1], llh[0], llh[2])
(ow, oh), onp = O.build_cube_ray(xpts, ypts, zpts, look, ip, MAX_SEGMENT_LENGTH=400.0, MAX_TROPO_HEIGHT=8940.0, return_nparts=True)
assert all((np.size(a) == 0 and b is None) or (b is not None and np.array_equal(a, b)) for a, b in zip(nparts, onp)), 'nParts (reversed segment)'
assert (hydro[1:5] > 0).all() and (hydro[5] == 0).all() and np.isfinite(hydro).all()
out['reversed_segment_max_abs_m'] = float(max(np.abs(ow - wet).max(), np.abs(oh - hydro).max()))

rng = np.random.default_rng(7)
cube = O.synthetic_cube(30, 28, 25, seed=77)
xpts = np.linspace(-121.3, -112.8, 17); ypts = np.linspace(36.2, 29.9, 13); zpts = np.array([-150.0, 0.0, 777.0, float(cube['zs'][-1])])
wet, hydro = H.rdelay._build_cube(xpts, ypts, zpts, H.EPSG4326, H.EPSG4326, H.scipy_interps(cube, 'total'))
ow, oh = O.build_cube(xpts, ypts, zpts, list(O.getInterpolators(cube['xs'], cube['ys'], cube['zs'], cube['wet_total'], cube['hydro_total'])))
assert np.array_equal(np.isnan(ow), np.isnan(wet)) and np.isnan(wet).any()
m = np.isfinite(wet)
out['zenith_max_rel'] = float(max(np.abs(ow - wet)[m].max() / np.abs(wet[m]).max(), np.abs(oh - hydro)[m].max() / np.abs(hydro[m]).max()))     # of the field maximum
import scipy          # noqa: E402
out.update(H.ref_import.provenance())       # geodesy: real pyproj / PROJ or the builder stub; look_vectors: isce3 or absent
out['python'] = '.'.join(str(v) for v in sys.version_info[:3]); out['numpy'] = np.__version__; out['scipy'] = scipy.__version__
if not H.HAVE_NATIVES:          # (another interpreter than the one oracle/_ref was built for: the Python path only)
    print(json.dumps(out))
    sys.exit(0)
grids = (np.sort(rng.uniform(0, 5, 9)), np.linspace(-1, 1, 7), np.sort(rng.uniform(10, 20, 5)))
vals = rng.standard_normal((9, 7, 5)); q = np.stack([rng.uniform(-0.5, 5.5, 500), rng.uniform(-1.2, 1.2, 500), rng.uniform(9, 21, 500)], -1)
out['natives_bit_exact'] = all(np.array_equal(O.native_interpolate(grids, vals, q, fill_value=f), H.r_interpolate(grids, vals, q, fill_value=f), equal_nan=True)
                               for f in (None, np.nan, 3.5))
sp = rng.uniform(-6e6, 6e6, (4, 5, 3)); slv = rng.standard_normal((4, 5, 3))
out['makepoints_bit_exact'] = bool(np.array_equal(O.makePoints(1234.5, sp, slv, 100.0), np.asarray(H.r_mp.makePoints2D(1234.5, sp, slv, 100.0))))
print(json.dumps(out))
