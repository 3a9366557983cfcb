"""Seeded synthetic workloads of SURVEY.md §8(d) (bench / examples).  Pure input generation."""
import numpy as np


def synthetic_cube(ny, nx, nz, seed=0, ztop=41000.0, y0=30.0, y1=36.0, x0=-121.0, x1=-113.0, zs=None):
    """ERA5-like processed weather model: dict(xs, ys, zs, wet, hydro (z,y,x) f32, wet_total,
    hydro_total (z,y,x) f64) laid out as weatherModel.py:685-693 writes it.  zs: the level heights to use (ascending, nz of them;
    e.g. a model's real axis, real_level_heights) instead of the quadratic stand-in."""
    ys = np.linspace(y0, y1, ny)
    xs = np.linspace(x0, x1, nx)
    if zs is None:
        zs = np.round(-100 + ztop * np.linspace(0, 1, nz) ** 2, 3)
    else:
        zs = np.ascontiguousarray(zs, dtype=np.float64)
        if zs.shape != (nz,) or not np.all(np.diff(zs) > 0):
            raise ValueError('synthetic_cube: zs must hold nz ascending heights')
    rng = np.random.default_rng(seed)
    g_h = rng.standard_normal((ny, nx))
    g_w = rng.standard_normal((ny, nx))
    z3 = zs[:, None, None]
    hydro = (270.0 * np.exp(-z3 / 8000.0) * (1 + 0.01 * g_h[None])).astype(np.float32)
    wet = (60.0 * np.exp(-z3 / 2000.0) * (1 + 0.1 * g_w[None])).astype(np.float32)

    def totals(f):   # weatherModel.py:389-403 `_getZTD`
        f = f.astype(np.float64)
        seg = 0.5 * (f[1:] + f[:-1]) * np.diff(zs)[:, None, None]
        cum = np.concatenate([np.cumsum(seg[::-1], axis=0)[::-1], np.zeros((1,) + f.shape[1:])], axis=0)
        return 1e-6 * cum
    return dict(xs=xs, ys=ys, zs=zs, wet=wet, hydro=hydro, wet_total=totals(wet), hydro_total=totals(hydro))


def scene_grid(rows, cols, row0=0, nrows=None, total_rows=None):
    """Query grid of configs 2-4: xpts = linspace(-119.5,-115.5,cols), ypts descending
    linspace(34.5,31.5,total_rows)[row0:row0+nrows] (llreader.py:191), per-pixel incidence
    30+16*col/cols deg (S1-like swath), heading -167.9 deg."""
    total_rows = total_rows or rows
    nrows = nrows or rows
    xpts = np.linspace(-119.5, -115.5, cols)
    ypts = np.linspace(34.5, 31.5, total_rows)[row0:row0 + nrows]
    inc_cols = 30.0 + 16.0 * (np.arange(cols) / float(cols))
    return xpts, ypts, inc_cols, -167.9


def real_level_heights(model):
    """The z axis of a processed model as the reference lays it out (models/model_levels.py, shipped as data under raider_amd/data):
    'era5' - the 145 heights every ECMWF model is resampled to (-500 m .. 80.3 km); 'hrrr' - HRRR's 50 native + 7 padding levels
    (-500 m .. 26.2 km).  Ascending float64."""
    from pathlib import Path
    f = {'era5': 'ecmwf_l137.npz', 'hrrr': 'hrrr_l50.npz'}[model]
    h = np.load(Path(__file__).resolve().parent / 'data' / f)['level_heights']
    return np.sort(np.asarray(h, dtype=np.float64))
