"""ctypes front end of oracle/oracle_c.c (TEST INFRASTRUCTURE: multi-core CPU restatement of the ray-traced path,
used by tests/test_oracle_c.py and bench.py's cpu_baseline leg - never by the product)."""
import ctypes as C
import subprocess
from pathlib import Path

import numpy as np

from . import raider_oracle as O

HERE = Path(__file__).resolve().parent
SO = HERE / 'liboracle_c.so'
_lib = None


def build():
    src = HERE / 'oracle_c.c'
    if not SO.exists() or SO.stat().st_mtime < src.stat().st_mtime:
        subprocess.run(['gcc', '-O3', '-fopenmp', '-shared', '-fPIC', '-ffp-contract=off', str(src), '-o', str(SO), '-lm'], check=True)


def usable_cpus():
    """CPUs this process may really use: scheduler affinity, capped by a cgroup CPU quota when there is one."""
    import os
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    for path in ('/sys/fs/cgroup/cpu.max', '/sys/fs/cgroup/cpu/cpu.cfs_quota_us'):
        try:
            txt = open(path).read().split()
            if path.endswith('cpu.max'):
                if txt[0] != 'max':
                    n = min(n, max(1, int(float(txt[0]) / float(txt[1]))))
            else:
                q = int(txt[0]); per = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
                if q > 0:
                    n = min(n, max(1, q // per))
        except (OSError, ValueError, IndexError):
            pass
    return n


def lib():
    global _lib
    if _lib is None:
        build()
        _lib = C.CDLL(str(SO))
        _lib.orc_num_threads.restype = C.c_int
        import os
        if 'OMP_NUM_THREADS' not in os.environ:
            _lib.orc_set_threads(C.c_int(usable_cpus()))
    return _lib


def _p(a):
    return a.ctypes.data_as(C.c_void_p)


_cube_cache = []


def _yxz(cube):
    """(y,x,z) C-order copies of the fields, cached for the LAST cube (the transposes are not part of the path).  The entry holds the
    source arrays themselves: identity is checked with `is`, which a recycled id() of a freed array cannot fool."""
    if _cube_cache and _cube_cache[0] is cube['wet'] and _cube_cache[1] is cube['hydro']:
        return _cube_cache[2], _cube_cache[3]
    wet = np.ascontiguousarray(np.asarray(cube['wet']).transpose(1, 2, 0)); hyd = np.ascontiguousarray(np.asarray(cube['hydro']).transpose(1, 2, 0))
    if wet.dtype != np.float32:
        wet, hyd = wet.astype(np.float64), hyd.astype(np.float64)
    _cube_cache[:] = [cube['wet'], cube['hydro'], wet, hyd]
    return wet, hyd


def num_threads():
    return lib().orc_num_threads()


def _slice_inputs(cube, xpts, ypts, ht, los, zref):
    xx, yy = np.meshgrid(np.asarray(xpts, float), np.asarray(ypts, float))
    lat = np.ascontiguousarray(yy.ravel()); lon = np.ascontiguousarray(xx.ravel())
    los = np.ascontiguousarray(np.asarray(los, dtype=np.float64).reshape(-1, 3))
    levels = O.ray_levels(cube['zs'], ht, zref)
    lo = np.array([a for a, _ in levels]); hi = np.array([b for _, b in levels])
    return yy.shape, lat, lon, los, lo, hi


def ray_prepass(cube, xpts, ypts, ht, los, zref):
    """Pass 1 alone: (maxlen[K], (all_first_below_zmin, all_last_above_zmax)) of the rays of meshgrid(xpts, ypts).  Blocks of one
    slice combine as element-wise max / logical AND - what the reference's whole-slice reductions (delay.py:283,306-311) do."""
    L = lib()
    _, lat, lon, los, lo, hi = _slice_inputs(cube, xpts, ypts, ht, los, zref)
    zs = np.asarray(cube['zs'], dtype=np.float64)
    maxlen = np.zeros(len(lo)); clamp = (C.c_int * 2)()
    L.orc_prepass(_p(lat), _p(lon), _p(los), C.c_int64(lat.size), C.c_double(ht), _p(lo), _p(hi), C.c_int(len(lo)), C.c_double(zs.min()),
                  C.c_double(zs.max()), _p(maxlen), clamp)
    return maxlen, (int(clamp[0]), int(clamp[1]))


def nparts_of(maxlen, max_seg=1000.0):
    return np.ceil(np.asarray(maxlen) / max_seg).astype(int) + 1          # delay.py:283


def build_cube_ray_slice(cube, xpts, ypts, ht, los, zref, max_seg=1000.0, nparts=None, clamp=None, model_proj=None):
    """One height slice of _build_cube_ray (delay.py:256-323) on meshgrid(xpts, ypts) with look vectors los (ny,nx,3).
    cube: dict(xs, ys, zs, wet, hydro (z,y,x)).  Returns (wet, hydro, nparts).  nparts / clamp: the whole slice's partition and
    z-clamp decisions when these rays are only a block of it (default: this block's own).  model_proj: None (lon/lat cube) or the
    Lambert-conformal-conic parameters of the cube's CRS (dict with lat_1, lat_2, lat_0, lon_0, x_0, y_0, a, es; delay.py:253,295)."""
    L = lib()
    shape, lat, lon, los, lo, hi = _slice_inputs(cube, xpts, ypts, ht, los, zref)
    n = lat.size
    K = len(lo)
    ys, xs, zs = (np.ascontiguousarray(cube[k], dtype=np.float64) for k in ('ys', 'xs', 'zs'))
    wet, hyd = _yxz(cube)
    dtype = 0 if wet.dtype == np.float32 else 1
    if nparts is None or clamp is None:
        maxlen = np.zeros(K); own = (C.c_int * 2)()
        L.orc_prepass(_p(lat), _p(lon), _p(los), C.c_int64(n), C.c_double(ht), _p(lo), _p(hi), C.c_int(K), C.c_double(zs.min()), C.c_double(zs.max()),
                      _p(maxlen), own)
        if nparts is None:
            nparts = nparts_of(maxlen, max_seg)
        if clamp is None:
            clamp = (own[0], own[1])
    np32 = np.ascontiguousarray(nparts, dtype=np.int32)
    ow, oh = np.empty(n), np.empty(n)
    proj = None
    if model_proj is not None:
        mp = dict(x_0=0.0, y_0=0.0, a=6371229.0, es=0.0); mp.update(model_proj)
        proj = np.array([mp['a'], mp['es'], mp['lat_1'], mp['lat_2'], mp['lat_0'], mp['lon_0'], mp['x_0'], mp['y_0']], dtype=np.float64)
    L.orc_march_proj(_p(lat), _p(lon), _p(los), C.c_int64(n), C.c_double(ht), _p(lo), _p(hi), C.c_int(K), _p(np32), C.c_int(clamp[0]), C.c_int(clamp[1]),
                     _p(ys), C.c_int(ys.size), _p(xs), C.c_int(xs.size), _p(zs), C.c_int(zs.size), _p(wet), _p(hyd), C.c_int(dtype),
                     _p(proj) if proj is not None else None, _p(ow), _p(oh))
    return ow.reshape(shape), oh.reshape(shape), np.asarray(nparts)


def _pp_inputs(cube, lat, lon, hts, los):
    shape = np.shape(lat)
    lat = np.ascontiguousarray(np.asarray(lat, dtype=np.float64).ravel()); lon = np.ascontiguousarray(np.asarray(lon, dtype=np.float64).ravel())
    hts = np.ascontiguousarray(np.broadcast_to(np.asarray(hts, dtype=np.float64), shape).ravel())
    los = np.ascontiguousarray(np.asarray(los, dtype=np.float64).reshape(-1, 3))
    return shape, lat, lon, hts, los


def per_pixel_prepass(cube, lat, lon, hts, los, zref):
    """Pass 1 of the per-ray-height rule alone: (maxlen[nz-1] by model interval, (clamp_lo, clamp_hi)).  Blocks of one batch combine
    as element-wise max / logical AND."""
    L = lib()
    _, lat, lon, hts, los = _pp_inputs(cube, lat, lon, hts, los)
    zs = np.ascontiguousarray(cube['zs'], dtype=np.float64)
    maxlen = np.zeros(zs.size - 1); clamp = (C.c_int * 2)(); anyl = C.c_int()
    L.orc_prepass_pp(_p(lat), _p(lon), _p(hts), _p(los), C.c_int64(lat.size), _p(zs), C.c_int(zs.size), C.c_double(zref), C.c_double(zs.min()),
                     C.c_double(zs.max()), _p(maxlen), clamp, C.byref(anyl))
    return maxlen, (int(clamp[0]), int(clamp[1]))


def per_pixel_nparts(maxlen, max_seg=1000.0):
    with np.errstate(invalid='ignore'):
        return np.where(maxlen > 0, np.ceil(maxlen / max_seg) + 1, 0).astype(np.int32)


def build_cube_ray_per_pixel(cube, lat, lon, hts, los, zref, max_seg=1000.0, nparts=None, clamp=None):
    """Rays with their OWN origin heights (no reference semantics; rule in oracle_c.c / DESIGN.md 5c): lat, lon, hts of one shape,
    los (..., 3).  Returns (wet, hydro, nparts[nz-1]) with nparts indexed by model interval (0 where no ray passes).  nparts / clamp:
    the whole batch's when these rays are only a block of it."""
    L = lib()
    shape, lat, lon, hts, los = _pp_inputs(cube, lat, lon, hts, los)
    n = lat.size
    ys, xs, zs = (np.ascontiguousarray(cube[k], dtype=np.float64) for k in ('ys', 'xs', 'zs'))
    wet, hyd = _yxz(cube)
    dtype = 0 if wet.dtype == np.float32 else 1
    if nparts is None or clamp is None:
        M = zs.size - 1
        maxlen = np.zeros(M); own = (C.c_int * 2)(); anyl = C.c_int()
        L.orc_prepass_pp(_p(lat), _p(lon), _p(hts), _p(los), C.c_int64(n), _p(zs), C.c_int(zs.size), C.c_double(zref), C.c_double(zs.min()),
                         C.c_double(zs.max()), _p(maxlen), own, C.byref(anyl))
        if nparts is None:
            if np.isnan(maxlen).any():
                raise ValueError('some ray lengths are NaN: the number of integration parts (delay.py:283) is undefined')
            nparts = per_pixel_nparts(maxlen, max_seg)
        if clamp is None:
            clamp = (own[0], own[1])
    np32 = np.ascontiguousarray(nparts, dtype=np.int32)
    ow, oh = np.empty(n), np.empty(n)
    L.orc_march_pp(_p(lat), _p(lon), _p(hts), _p(los), C.c_int64(n), _p(zs), C.c_double(zref), _p(np32), C.c_int(clamp[0]), C.c_int(clamp[1]),
                   _p(ys), C.c_int(ys.size), _p(xs), C.c_int(xs.size), _p(zs), C.c_int(zs.size), _p(wet), _p(hyd), C.c_int(dtype), _p(ow), _p(oh))
    return ow.reshape(shape), oh.reshape(shape), np.asarray(np32)
