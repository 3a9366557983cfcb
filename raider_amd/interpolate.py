"""`RAiDER.interpolate` (tools/bindings/interpolate/src/module.cpp) on the GPU.

Same argument validation and exception types as the pybind11 module; `assume_sorted` and `max_threads`
are accepted for signature compatibility (the GPU kernel always bisects, which gives the same answers
the reference gives for sorted input, and is data-parallel over all 256 CUs instead of <= 8 threads)."""
import ctypes as C

import numpy as np

from ._lib import Context, check, f64, ptr, RDR_HOST


def interpolate(points, values, interp_points, fill_value=None, assume_sorted=False, max_threads=8):
    """module.cpp:26-294: N-D linear interpolation on a rectilinear grid; returns shape (N,) float64."""
    points = [np.asarray(p) for p in points]
    values = np.asarray(values)
    interp_points = np.asarray(interp_points)
    if values.ndim == 0 or interp_points.ndim == 0:
        raise TypeError('Only arrays are supported, not scalar values!')                   # module.cpp:36-38
    for p in points:
        if p.ndim != 1:
            raise TypeError("'points' must be a list of 1D arrays!")                      # module.cpp:40-44
    nd = len(points)
    if nd != values.ndim:
        raise TypeError(f'Dimension mismatch! Grid is {nd}D but values are {values.ndim}D!')   # module.cpp:46-51
    if interp_points.ndim != 2:
        raise TypeError("'interp_points' should have shape (N, ndim).")                   # module.cpp:53-55
    if interp_points.shape[1] != nd:
        raise TypeError(f'Dimension mismatch! Grid is {nd}D but interpolation points are {interp_points.shape[1]}D!')
    if nd > 8:
        raise NotImplementedError('raider_amd.interpolate supports up to 8 dimensions on the device')
    for p, s in zip(points, values.shape):
        if p.size != s:
            raise TypeError('grid axes do not match the shape of values')
    axes = [f64(p) for p in points]
    vals = f64(values)
    q = f64(interp_points)
    n = q.shape[0]
    out = np.empty(n)
    ctx = Context.default()
    ax_ptrs = (C.c_void_p * nd)(*[a.ctypes.data for a in axes])
    ax_len = (C.c_int64 * nd)(*[a.size for a in axes])
    check(ctx.lib.rdr_interp_nd(ctx.handle, nd, ax_ptrs, ax_len, ptr(vals), ptr(q), n, int(fill_value is not None),
                                float(fill_value) if fill_value is not None else 0.0, ptr(out), RDR_HOST), ctx.handle, TypeError)
    return out


def interpolate_along_axis(points, values, interp_points, axis=-1, fill_value=None, assume_sorted=False, max_threads=8):
    """module.cpp:296-493: independent 1-D interpolation along `axis`; output has interp_points' shape."""
    points = np.asarray(points)
    values = np.asarray(values)
    interp_points = np.asarray(interp_points)
    if values.ndim == 0 or interp_points.ndim == 0:
        raise TypeError('Only arrays are supported, not scalar values!')
    if points.ndim != values.ndim or points.ndim != interp_points.ndim:
        raise TypeError("'points', 'values' and 'interp_points' must all have the same number of dimensions!")
    nd = points.ndim
    if points.shape != values.shape:
        raise TypeError("'points' and 'values' must have the same shape!")
    if axis < 0:
        axis += nd
    if axis >= nd or axis < 0:
        raise TypeError("'axis' out of range!")
    if axis == 0 and max_threads > 1:
        raise RuntimeError('Cannot interpolate along axis 0 with multiple threads!')       # module.cpp:332-335
    for i in range(nd):
        if i != axis and interp_points.shape[i] != points.shape[i]:
            raise TypeError(f"Dimension mismatch at axis {i}! 'points' is {points.shape[i]} but interp_points is "
                            f'{interp_points.shape[i]}!')
    P = np.ascontiguousarray(np.moveaxis(points, axis, -1), dtype=np.float64)
    V = np.ascontiguousarray(np.moveaxis(values, axis, -1), dtype=np.float64)
    Q = np.ascontiguousarray(np.moveaxis(interp_points, axis, -1), dtype=np.float64)
    m, mq = P.shape[-1], Q.shape[-1]
    ncol = int(np.prod(P.shape[:-1])) if P.ndim > 1 else 1
    out = np.empty(Q.shape)
    ctx = Context.default()
    check(ctx.lib.rdr_interp_along_axis(ctx.handle, ptr(P), ptr(V), ncol, m, ptr(Q), mq, int(fill_value is not None),
                                        float(fill_value) if fill_value is not None else 0.0, ptr(out), RDR_HOST), ctx.handle, TypeError)
    return np.ascontiguousarray(np.moveaxis(out, -1, axis))
