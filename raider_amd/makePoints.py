"""`RAiDER.makePoints` (tools/bindings/utils/makePoints.pyx:15-148) on the GPU: fixed-step points along
straight rays, output (..., 3, Npts) float64."""
import numpy as np

from ._lib import Context, check, f64, load, ptr, RDR_HOST


def _make(max_len, Rays_SP, Rays_SLV, stepSize, ndim):
    sp = np.asarray(Rays_SP)
    slv = np.asarray(Rays_SLV)
    # Cython typed-buffer checks: ndim and dtype mismatches raise ValueError there
    if sp.ndim != ndim + 1 or slv.ndim != ndim + 1:
        raise ValueError(f'Buffer has wrong number of dimensions (expected {ndim + 1}, got {sp.ndim})')
    if sp.dtype != np.float64 or slv.dtype != np.float64:
        raise ValueError("Buffer dtype mismatch, expected 'double'")
    if sp.shape != slv.shape or sp.shape[-1] != 3:
        raise ValueError('Rays_SP and Rays_SLV must both have shape (..., 3)')
    max_len = float(max_len)
    stepSize = float(stepSize)
    npts = load().rdr_make_points_count(max_len, stepSize)
    if npts < 0:
        raise ZeroDivisionError('float modulo')
    sp2, slv2 = f64(sp).reshape(-1, 3), f64(slv).reshape(-1, 3)
    out = np.empty(sp.shape[:-1] + (3, npts))
    ctx = Context.default()
    check(ctx.lib.rdr_make_points(ctx.handle, max_len, ptr(sp2), ptr(slv2), sp2.shape[0], stepSize, ptr(out), RDR_HOST), ctx.handle)
    return out


def makePoints0D(max_len, Rays_SP, Rays_SLV, stepSize):
    """makePoints.pyx:15-41: (3,) -> (3, Npts)."""
    return _make(max_len, Rays_SP, Rays_SLV, stepSize, 0)


def makePoints1D(max_len, Rays_SP, Rays_SLV, stepSize):
    """makePoints.pyx:45-75: (Nx,3) -> (Nx, 3, Npts)."""
    return _make(max_len, Rays_SP, Rays_SLV, stepSize, 1)


def makePoints2D(max_len, Rays_SP, Rays_SLV, stepSize):
    """makePoints.pyx:79-111: (Nx,Ny,3) -> (Nx, Ny, 3, Npts)."""
    return _make(max_len, Rays_SP, Rays_SLV, stepSize, 2)


def makePoints3D(max_len, Rays_SP, Rays_SLV, stepSize):
    """makePoints.pyx:115-148: (Nx,Ny,Nz,3) -> (Nx, Ny, Nz, 3, Npts)."""
    return _make(max_len, Rays_SP, Rays_SLV, stepSize, 3)
