"""Seeded synthetic workloads of SURVEY.md §8(d) (bench / examples).  Pure input generation."""
import numpy as np


def synthetic_cube(ny, nx, nz, seed=0, ztop=41000.0, y0=30.0, y1=36.0, x0=-121.0, x1=-113.0):
    """ERA5-like processed weather model: dict(xs, ys, zs, wet, hydro (z,y,x) f32, wet_total,
    hydro_total (z,y,x) f64) laid out as weatherModel.py:685-693 writes it."""
    ys = np.linspace(y0, y1, ny)
    xs = np.linspace(x0, x1, nx)
    zs = np.round(-100 + ztop * np.linspace(0, 1, nz) ** 2, 3)
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
