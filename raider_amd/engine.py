"""Thin object layer over the C ABI: device-resident weather cubes and ray batches.

Everything here only marshals arguments; all arithmetic happens in the HIP kernels
(raider_amd/csrc).  NumPy arrays are staged by the library (RDR_HOST); objects exposing
`data_ptr()` (torch tensors on the GPU) are passed through as device pointers (RDR_DEVICE).
"""
import ctypes as C

import numpy as np

from . import _lib as L
from . import _pinned
from ._lib import Context, check, f64, ptr


def _is_dev(a):
    return hasattr(a, 'data_ptr')


def _dev_f64(t, what, numel=None):
    """A device array the kernels read or write through its raw pointer: float64, contiguous, on a GPU, of the expected size."""
    import torch
    if t.dtype != torch.float64 or not t.is_cuda or not t.is_contiguous():
        raise TypeError(f'{what} must be a contiguous float64 tensor on the GPU (got {t.dtype} on {t.device}, contiguous={t.is_contiguous()})')
    if numel is not None and t.numel() != numel:
        raise ValueError(f'{what} has {t.numel()} elements, {numel} expected')
    return t


class Cube:
    """Both fields of one processed weather model on the GPU, with scipy-RGI semantics.

    Replaces the pair of scipy interpolators getInterpolators returns (delayFcns.py:23-58)."""

    def __init__(self, ys, xs, zs, wet, hydro, order='yxz', ctx=None):
        """wet/hydro: arrays of identical shape/dtype (f32 or f64); order 'yxz' (interpolator order)
        or 'zyx' (file order, weatherModel.py:685-693) - no host transpose is made."""
        self.ctx = ctx or Context.default()
        self.handle = None
        ys, xs, zs = f64(ys), f64(xs), f64(zs)
        dev = _is_dev(wet)
        if not dev:
            wet = np.asarray(wet)
            hydro = np.asarray(hydro)
            # float32 / float64 in EITHER byte order go up as they are (a big-endian NetCDF-3 mapping is swapped on the device while
            # the cube is packed - no host pass over the fields); anything else is converted to float64 first
            if wet.dtype.kind != 'f' or wet.dtype.itemsize not in (4, 8):
                wet = wet.astype(np.float64)
            if hydro.dtype != wet.dtype:
                hydro = hydro.astype(wet.dtype)
            wet = np.ascontiguousarray(wet)
            hydro = np.ascontiguousarray(hydro)
            dt = L.RDR_F32 if wet.dtype.itemsize == 4 else L.RDR_F64
            if not wet.dtype.isnative:
                dt |= L.RDR_BYTESWAPPED
            shape = wet.shape
        else:
            import torch
            if wet.dtype not in (torch.float32, torch.float64) or hydro.dtype != wet.dtype:
                raise TypeError(f'device cubes must be float32 or float64 tensors of one dtype (got {wet.dtype} / {hydro.dtype})')
            if not (wet.is_cuda and hydro.is_cuda and wet.is_contiguous() and hydro.is_contiguous()):
                raise ValueError('device cubes must be contiguous tensors on the GPU')
            dt = L.RDR_F32 if wet.dtype == torch.float32 else L.RDR_F64
            shape = tuple(wet.shape)
            self.ctx.adopt_torch_stream(wet)
        ny, nx, nz = ys.size, xs.size, zs.size
        if order == 'yxz':
            want = (ny, nx, nz)
            sy, sx, sz = nx * nz, nz, 1
        elif order == 'zyx':
            want = (nz, ny, nx)
            sy, sx, sz = nx, 1, ny * nx
        else:
            raise ValueError("order must be 'yxz' or 'zyx'")
        if tuple(shape) != want or tuple(hydro.shape) != want:
            raise ValueError(f'There are {want} points in the grid but values have shape {tuple(shape)}')
        h = C.c_void_p()
        check(self.ctx.lib.rdr_cube_create(self.ctx.handle, ptr(ys), ny, ptr(xs), nx, ptr(zs), nz, ptr(wet), ptr(hydro), dt,
                                           sy, sx, sz, L.RDR_DEVICE if dev else L.RDR_HOST, C.byref(h)), self.ctx.handle)
        self.handle = h
        self.dtype = np.float32 if (dt & 0xff) == L.RDR_F32 else np.float64
        self.shape = (ny, nx, nz)
        gy, gx, gz = np.empty(ny), np.empty(nx), np.empty(nz)
        check(self.ctx.lib.rdr_cube_axes(h, ptr(gy), ptr(gx), ptr(gz)))
        self.grid = (gy, gx, gz)       # ascending, as scipy exposes `.grid` (delay.py:239)
        self.projection = None

    @classmethod
    def _from_handle(cls, ctx, h):
        self = cls.__new__(cls)
        self.ctx, self.handle = ctx, h
        ny, nx, nz, dt = C.c_int64(), C.c_int64(), C.c_int64(), C.c_int()
        check(ctx.lib.rdr_cube_shape(h, C.byref(ny), C.byref(nx), C.byref(nz), C.byref(dt)))
        self.shape = (ny.value, nx.value, nz.value)
        self.dtype = np.float32 if dt.value == L.RDR_F32 else np.float64
        gy, gx, gz = np.empty(ny.value), np.empty(nx.value), np.empty(nz.value)
        check(ctx.lib.rdr_cube_axes(h, ptr(gy), ptr(gx), ptr(gz)))
        self.grid = (gy, gx, gz)
        self.projection = None
        return self

    def set_projection_lcc(self, lat_1, lat_2, lat_0, lon_0, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0):
        """The cube's x/y axes are Lambert-conformal-conic metres (HRRR: models/hrrr.py:248-259, sphere a=6371229).
        build_cube / ray tracing then take geodetic lon/lat queries and project them on the device."""
        p = np.array([a, es, lat_1, lat_2, lat_0, lon_0, x_0, y_0], dtype=np.float64)
        check(self.ctx.lib.rdr_cube_set_projection(self.handle, 1, ptr(p), p.size), self.ctx.handle)
        self.projection = dict(proj='lcc', lat_1=lat_1, lat_2=lat_2, lat_0=lat_0, lon_0=lon_0, x_0=x_0, y_0=y_0, a=a, es=es)
        return self

    def set_projection_stere(self, lat_0=90.0, lat_ts=None, k_0=1.0, lon_0=0.0, x_0=0.0, y_0=0.0, a=6371229.0, es=0.0):
        """The cube's x/y axes are POLAR stereographic metres (HRRR-AK: models/hrrr.py:22-25,359 `+proj=stere +lat_0=90
        +lon_0=225 +lat_ts=60`, sphere a=6371229).  lat_ts=None: scale k_0 at the pole."""
        p = np.array([a, es, lat_0, np.nan if lat_ts is None else lat_ts, k_0, lon_0, x_0, y_0], dtype=np.float64)
        check(self.ctx.lib.rdr_cube_set_projection(self.handle, 2, ptr(p), p.size), self.ctx.handle)
        self.projection = dict(proj='stere', lat_0=lat_0, lat_ts=lat_ts, k_0=k_0, lon_0=lon_0, x_0=x_0, y_0=y_0, a=a, es=es)
        return self

    def clear_projection(self):
        """Back to a lon/lat (EPSG:4326) cube."""
        check(self.ctx.lib.rdr_cube_set_projection(self.handle, 0, None, 0), self.ctx.handle)
        self.projection = None
        return self

    def view(self, projection='same', ctx=None):
        """A second handle on the SAME device buffers (values, axes, corner-quad copy: nothing is copied) that owns only its projection
        (rdr_cube_view) - how a cached cube serves callers with different model-CRS arguments without ever being modified.
        projection: 'same' (this cube's), None (lon/lat) or a dict as `Cube.projection` holds it (proj='lcc' / 'stere' + parameters).
        The view keeps this cube alive; the C side frees the buffers when the source and every view are gone, in any order."""
        ctx = ctx or self.ctx
        if projection == 'same':
            projection = self.projection
        if projection is None:
            kind, p = 0, None
        elif projection.get('proj') == 'lcc':
            d = projection
            kind, p = 1, np.array([d.get('a', 6371229.0), d.get('es', 0.0), d['lat_1'], d['lat_2'], d['lat_0'], d['lon_0'], d.get('x_0', 0.0), d.get('y_0', 0.0)], dtype=np.float64)
        elif projection.get('proj') == 'stere':
            d = projection
            lat_ts = d.get('lat_ts')
            kind, p = 2, np.array([d.get('a', 6371229.0), d.get('es', 0.0), d.get('lat_0', 90.0), np.nan if lat_ts is None else lat_ts, d.get('k_0', 1.0),
                                   d.get('lon_0', 0.0), d.get('x_0', 0.0), d.get('y_0', 0.0)], dtype=np.float64)
        else:
            raise ValueError(f'Cube.view: unknown projection {projection!r}')
        h = C.c_void_p()
        check(ctx.lib.rdr_cube_view(ctx.handle, self.handle, kind, ptr(p), 0 if p is None else p.size, C.byref(h)), ctx.handle)
        v = Cube.__new__(Cube)
        v.ctx, v.handle, v.shape, v.dtype, v.grid = ctx, h, self.shape, self.dtype, self.grid
        v.projection = None if projection is None else dict(projection)
        v._source = self
        return v

    def project(self, lats, lons):
        """EPSG:4326 (lat, lon) -> the cube's (y, x) coordinates (identity for lon/lat cubes)."""
        lats, lons = np.broadcast_arrays(np.asarray(lats, dtype=np.float64), np.asarray(lons, dtype=np.float64))
        la, lo = f64(lats).ravel(), f64(lons).ravel()
        y, x = np.empty(la.size), np.empty(la.size)
        check(self.ctx.lib.rdr_project_points(self.ctx.handle, self.handle, ptr(la), ptr(lo), la.size, ptr(y), ptr(x), L.RDR_HOST), self.ctx.handle)
        return y.reshape(lats.shape), x.reshape(lats.shape)

    def blend(self, w1, other, w2):
        """cli/raider.py:817-819: w1*self + w2*other on the device."""
        h = C.c_void_p()
        check(self.ctx.lib.rdr_cube_blend(self.ctx.handle, self.handle, float(w1), other.handle, float(w2), C.byref(h)), self.ctx.handle)
        out = Cube._from_handle(self.ctx, h)
        out.projection = self.projection       # the C side copies the projection too
        return out

    def has_nan(self):
        """A NaN among the two fields this cube was made from (seen on the device while packing; delayFcns.py:50-52)."""
        return bool(self.ctx.lib.rdr_cube_has_nan(self.handle) == 1)

    def read(self):
        wet = np.empty(self.shape, dtype=self.dtype)
        hydro = np.empty(self.shape, dtype=self.dtype)
        check(self.ctx.lib.rdr_cube_read(self.ctx.handle, self.handle, ptr(wet), ptr(hydro)), self.ctx.handle)
        return wet, hydro

    # ---- zenith / projected ------------------------------------------------------------------
    def point_index(self, build=True):
        """Build (or free) the corner-quad copy that serves large random point sets with one 128 B line per point instead of four
        (rdr_cube_point_index); interp() builds it by itself on the second large call.  Returns the bytes it holds."""
        check(self.ctx.lib.rdr_cube_point_index(self.ctx.handle, self.handle, 1 if build else 0), self.ctx.handle)
        return int(self.ctx.lib.rdr_cube_point_index_bytes(self.handle))

    def interp(self, pts, field=None):
        """scipy RGI __call__ on both fields; pts[...,3] = (y,x,z).  Returns (wet, hydro) f64.
        field=0 / 1 (host arrays): only that field is stored and downloaded, the other entry of the pair is None."""
        if _is_dev(pts):
            import torch
            if pts.shape[-1] != 3:
                raise ValueError(f'The requested sample points xi have dimension {pts.shape[-1]} but this '
                                 'RegularGridInterpolator has dimension 3')
            _dev_f64(pts, 'pts')
            self.ctx.adopt_torch_stream(pts)
            n = pts.numel() // 3
            wet = torch.empty(pts.shape[:-1], dtype=torch.float64, device=pts.device)
            hyd = torch.empty_like(wet)
            check(self.ctx.lib.rdr_interp3(self.ctx.handle, self.handle, ptr(pts), n, ptr(wet), ptr(hyd), L.RDR_DEVICE), self.ctx.handle)
            return wet, hyd
        pts = np.asarray(pts)
        if pts.shape[-1] != 3:
            raise ValueError(f'The requested sample points xi have dimension {pts.shape[-1]} but this '
                             'RegularGridInterpolator has dimension 3')
        p = f64(pts).reshape(-1, 3)
        # (large point sets: recycled page-locked results)
        wet = _pinned.empty((p.shape[0],)) if field in (None, 0) else None
        hyd = _pinned.empty((p.shape[0],)) if field in (None, 1) else None
        check(self.ctx.lib.rdr_interp3(self.ctx.handle, self.handle, ptr(p), p.shape[0], ptr(wet), ptr(hyd), L.RDR_HOST), self.ctx.handle)
        return (None if wet is None else wet.reshape(pts.shape[:-1])), (None if hyd is None else hyd.reshape(pts.shape[:-1]))

    def interp_blend(self, w1, other, w2, pts, via_cube=False):
        """interp() on the temporal blend w1 * self + w2 * other (cli/raider.py:817-819) without handing back a blended cube - bit for bit
        what `self.blend(w1, other, w2).interp(pts)` returns.  via_cube=False: the blend is applied at each point's eight corners
        (rdr_interp3_blend) - cheaper for point sets below ~5 % of the cube's cells (one rank's station block of a multi-GPU job).
        via_cube=True: the blend is made as a cube in the context's scratch, in the layout the gather reads best, and gathered in the same
        call (rdr_interp3_blend_cube) - for point sets large against the cube (distributed.blend_on_the_fly_pays decides)."""
        entry = self.ctx.lib.rdr_interp3_blend_cube if via_cube else self.ctx.lib.rdr_interp3_blend
        if _is_dev(pts):
            import torch
            if pts.shape[-1] != 3:
                raise ValueError(f'The requested sample points xi have dimension {pts.shape[-1]} but this RegularGridInterpolator has dimension 3')
            _dev_f64(pts, 'pts')
            self.ctx.adopt_torch_stream(pts)
            n = pts.numel() // 3
            wet = torch.empty(pts.shape[:-1], dtype=torch.float64, device=pts.device)
            hyd = torch.empty_like(wet)
            check(entry(self.ctx.handle, self.handle, float(w1), other.handle, float(w2), ptr(pts), None, None, n, ptr(wet), ptr(hyd),
                                                 L.RDR_DEVICE), self.ctx.handle)
            return wet, hyd
        pts = np.asarray(pts)
        if pts.shape[-1] != 3:
            raise ValueError(f'The requested sample points xi have dimension {pts.shape[-1]} but this RegularGridInterpolator has dimension 3')
        p = f64(pts).reshape(-1, 3)
        wet = _pinned.empty((p.shape[0],)); hyd = _pinned.empty((p.shape[0],))
        check(entry(self.ctx.handle, self.handle, float(w1), other.handle, float(w2), ptr(p), None, None, p.shape[0], ptr(wet), ptr(hyd),
                                             L.RDR_HOST), self.ctx.handle)
        return wet.reshape(pts.shape[:-1]), hyd.reshape(pts.shape[:-1])

    def interp_project(self, y, x=None, z=None, inc=None, divisor=None):
        """The second stage of tropo_delay's point branch (delay.py:110-128) in one launch: both fields at the points given as three
        arrays y, x, z of one shape (or y = packed pts[..., 3], x = z = None) and, for a projected line of sight, delay / cosd(inc)
        (`inc`: a scalar or an array of the points' shape, degrees) or delay / `divisor` (an array: cos of the look angle) -
        losreader.py:130-133 - before the values leave the device.  Returns (wet, hydro) f64 of the points' shape."""
        if _is_dev(y):
            # device-resident points (three float64 tensors of one shape, or one packed [..., 3]): nothing crosses PCIe, the call is asynchronous
            import torch
            if inc is not None and divisor is not None:
                raise ValueError('give inc= or divisor=, not both')
            self.ctx.adopt_torch_stream(y)
            if x is None:
                if y.shape[-1] != 3:
                    raise ValueError(f'The requested sample points xi have dimension {y.shape[-1]} but this RegularGridInterpolator has dimension 3')
                shape, n = tuple(y.shape[:-1]), y.numel() // 3
                _dev_f64(y, 'pts')
            else:
                shape, n = tuple(y.shape), y.numel()
                for t, what in ((y, 'y'), (x, 'x'), (z, 'z')):
                    _dev_f64(t, what, n)
            mode, parr, inc0 = 0, None, 0.0
            arr = inc if inc is not None else divisor
            if arr is not None:
                if _is_dev(arr):
                    mode, parr = (1 if inc is not None else 3), _dev_f64(arr, 'inc / divisor', n)
                elif np.ndim(arr) == 0 and inc is not None:
                    mode, inc0 = 2, float(arr)
                else:
                    mode = 1 if inc is not None else 3
                    parr = torch.from_numpy(f64(np.broadcast_to(np.asarray(arr, dtype=np.float64), shape))).to(y.device)
            wet = torch.empty(shape, dtype=torch.float64, device=y.device); hyd = torch.empty_like(wet)
            check(self.ctx.lib.rdr_interp3_project(self.ctx.handle, self.handle, ptr(y), ptr(x), ptr(z), n, mode, ptr(parr), inc0, ptr(wet), ptr(hyd),
                                                   L.RDR_DEVICE), self.ctx.handle)
            return wet, hyd
        ya, xa, za, n, shape, mode, parr, inc0 = self._point_args(y, x, z, inc, divisor)
        wet = _pinned.empty((n,)); hyd = _pinned.empty((n,))
        check(self.ctx.lib.rdr_interp3_project(self.ctx.handle, self.handle, ptr(ya), ptr(xa), ptr(za), n, mode, ptr(parr), inc0, ptr(wet), ptr(hyd),
                                               L.RDR_HOST), self.ctx.handle)
        return wet.reshape(shape), hyd.reshape(shape)

    @staticmethod
    def _point_args(y, x, z, inc, divisor):
        """(y, x, z as flat arrays - or y packed (n, 3), x = z = None -, n, shape, proj_mode, proj array, inc0) of a point query."""
        if inc is not None and divisor is not None:
            raise ValueError('give inc= or divisor=, not both')
        if x is None:
            y = np.asarray(y)
            if y.shape[-1] != 3:
                raise ValueError(f'The requested sample points xi have dimension {y.shape[-1]} but this RegularGridInterpolator has dimension 3')
            shape = y.shape[:-1]
            ya, xa, za = f64(y).reshape(-1, 3), None, None
            n = ya.shape[0]
        else:
            y, x, z = np.broadcast_arrays(np.asarray(y), np.asarray(x), np.asarray(z))
            shape = y.shape
            ya, xa, za = f64(y).reshape(-1), f64(x).reshape(-1), f64(z).reshape(-1)
            n = ya.size
        mode, parr, inc0 = 0, None, 0.0
        if inc is not None:
            if np.ndim(inc) == 0:
                mode, inc0 = 2, float(inc)
            else:
                mode, parr = 1, f64(np.broadcast_to(np.asarray(inc, dtype=np.float64), shape)).reshape(-1)
        elif divisor is not None:
            mode, parr = 3, f64(np.broadcast_to(np.asarray(divisor, dtype=np.float64), shape)).reshape(-1)
        return ya, xa, za, n, shape, mode, parr, inc0

    def point_delays(self, xpts, ypts, zpts, y, x=None, z=None, inc=None, divisor=None):
        """tropo_delay's point branch for a zenith / projected line of sight in ONE library call (rdr_point_delays): _build_cube of THIS
        (total-delay) cube on the output grid (xpts, ypts, zpts), the intermediate cube kept in device scratch, both fields gathered
        at the points (y, x, z: three arrays, or y = packed [..., 3]) and divided by cosd(inc) / `divisor` - delay.py:96-128,
        losreader.py:130-133.  Returns (wet, hydro, has_nan): has_nan = the intermediate cube holds a NaN (delay.py:187)."""
        gx, gy, gz = f64(xpts).ravel(), f64(ypts).ravel(), f64(np.atleast_1d(zpts)).ravel()
        ya, xa, za, n, shape, mode, parr, inc0 = self._point_args(y, x, z, inc, divisor)
        wet = _pinned.empty((n,)); hyd = _pinned.empty((n,))
        flag = C.c_int32(0)
        check(self.ctx.lib.rdr_point_delays(self.ctx.handle, self.handle, ptr(gx), gx.size, ptr(gy), gy.size, ptr(gz), gz.size, ptr(ya), ptr(xa), ptr(za), n,
                                            mode, ptr(parr), inc0, ptr(wet), ptr(hyd), C.byref(flag)), self.ctx.handle)
        return wet.reshape(shape), hyd.reshape(shape), bool(flag.value)

    def build_delay_cube(self, xpts, ypts, zpts):
        """_build_cube (delay.py:196-216) whose result stays on the device: a float64 `Cube` with axes (ypts, xpts, zpts) - the
        intermediate delay cube of tropo_delay's point branch (delay.py:96-121), ready for interp_project().  Its has_nan() is the
        np.isnan(...).any() of delay.py:187."""
        x, y, z = f64(xpts).ravel(), f64(ypts).ravel(), f64(np.atleast_1d(zpts)).ravel()
        h = C.c_void_p()
        check(self.ctx.lib.rdr_build_cube_to_cube(self.ctx.handle, self.handle, ptr(x), x.size, ptr(y), y.size, ptr(z), z.size, L.RDR_HOST, C.byref(h)),
              self.ctx.handle)
        return Cube._from_handle(self.ctx, h)

    def build_cube(self, xpts, ypts, zpts, out=None, want_nan=False):
        """_build_cube (delay.py:196-216): (wet, hydro) of shape (nz, ny, nx).  want_nan=True (host arrays): (wet, hydro, has_nan) with
        has_nan = np.isnan(result).any() as scanned on the device (None: not scanned) - part of the RESULT, not state of this object:
        a cached cube serves several threads."""
        if _is_dev(xpts):
            import torch
            self.ctx.adopt_torch_stream(xpts)
            nx, ny, nz = xpts.numel(), ypts.numel(), zpts.numel()
            for t, what in ((xpts, 'xpts'), (ypts, 'ypts'), (zpts, 'zpts')):
                _dev_f64(t, what)
            wet, hyd = out if out is not None else (torch.empty((nz, ny, nx), dtype=torch.float64, device=xpts.device),
                                                    torch.empty((nz, ny, nx), dtype=torch.float64, device=xpts.device))
            _dev_f64(wet, 'out[0]', nx * ny * nz); _dev_f64(hyd, 'out[1]', nx * ny * nz)
            check(self.ctx.lib.rdr_build_cube(self.ctx.handle, self.handle, ptr(xpts), nx, ptr(ypts), ny, ptr(zpts), nz,
                                              ptr(wet), ptr(hyd), L.RDR_DEVICE), self.ctx.handle)
            return (wet, hyd, None) if want_nan else (wet, hyd)       # (device results are not scanned: the call stays asynchronous)
        x, y, z = f64(xpts).ravel(), f64(ypts).ravel(), f64(np.atleast_1d(zpts)).ravel()
        # (large cubes in recycled page-locked memory: the 640 MB of a 1000 x 1000 x 40 zenith cube come down in 12 ms instead of 40-120 ms
        # into freshly mapped pageable pages - _pinned.py)
        wet = _pinned.empty((z.size, y.size, x.size)); hyd = _pinned.empty((z.size, y.size, x.size))
        with self.ctx.lock:          # (the verdict is context state: read it before another thread's build on this context replaces it)
            check(self.ctx.lib.rdr_build_cube(self.ctx.handle, self.handle, ptr(x), x.size, ptr(y), y.size, ptr(z), z.size,
                                              ptr(wet), ptr(hyd), L.RDR_HOST), self.ctx.handle)
            f = self.ctx.lib.rdr_last_nan_output(self.ctx.handle)
        if want_nan:
            return wet, hyd, (None if f < 0 else bool(f))             # np.isnan(result).any(), scanned on the device
        return wet, hyd

    # ---- rays ------------------------------------------------------------------------------------
    def ray_kernel_attributes(self, which):
        """Resources of the light ray kernel (0: pass 1, 1: pass 2) a GRID + look-vector batch on this cube launches, from the loaded
        code object: dict(vgpr, lds_static, lds_dynamic, scratch, max_threads)."""
        v = [C.c_int32() for _ in range(5)]
        check(self.ctx.lib.rdr_ray_kernel_attributes(self.ctx.handle, self.handle, int(which), *[C.byref(x) for x in v]), self.ctx.handle)
        return dict(zip(('vgpr', 'lds_static', 'lds_dynamic', 'scratch', 'max_threads'), (x.value for x in v)))

    def ray_levels(self, ht, zref):
        """(lo, hi, kz) of the contributing model intervals (losreader.py:785-808); raises NoLevels."""
        nz = self.shape[2]
        K = C.c_int32()
        lo, hi, kz = np.empty(nz), np.empty(nz), np.empty(nz, dtype=np.int32)
        check(self.ctx.lib.rdr_ray_levels(self.handle, float(ht), float(zref), C.byref(K), ptr(lo), ptr(hi), ptr(kz)), self.ctx.handle)
        return lo[:K.value].copy(), hi[:K.value].copy(), kz[:K.value].copy()

    def ray_prepass(self, rays, ht, zref):
        rays.adopt_stream(self.ctx)
        ht = rays.table_height(ht)
        K = len(self.ray_levels(ht, zref)[0])
        maxlen = np.zeros(K)
        flags = C.c_int32()
        check(self.ctx.lib.rdr_ray_prepass(self.ctx.handle, self.handle, C.byref(rays.struct), float(ht), float(zref), ptr(maxlen),
                                           C.byref(flags)), self.ctx.handle)
        return maxlen, flags.value

    def _check_partition(self, partition, ht, zref):
        K = len(self.ray_levels(ht, zref)[0])
        _dev_f64(partition, 'partition')
        if partition.numel() < K + 4:
            raise ValueError(f'partition needs K + 4 = {K + 4} elements (got {partition.numel()})')

    def ray_prepass_device(self, rays, ht, zref, partition):
        """Pass 1 with its result left on the device: `partition` = torch float64 tensor of K+4 elements (per-level maxima,
        then the 4 flag bits as 0/1) - ready for an element-wise MAX all-reduce.  Asynchronous."""
        rays.adopt_stream(self.ctx)
        ht = rays.table_height(ht)
        self._check_partition(partition, ht, zref)
        check(self.ctx.lib.rdr_ray_prepass_device(self.ctx.handle, self.handle, C.byref(rays.struct), float(ht), float(zref), ptr(partition)),
              self.ctx.handle)
        return partition

    def ray_march_device(self, rays, ht, zref, partition, max_seg=1000.0, out=None):
        """Pass 2 driven by a device-resident (all-reduced) partition.  Asynchronous."""
        rays.adopt_stream(self.ctx)
        ht = rays.table_height(ht)
        wet, hyd = out if out is not None else rays.empty_outputs()
        rays.check_outputs(wet, hyd)
        self._check_partition(partition, ht, zref)
        check(self.ctx.lib.rdr_ray_march_device(self.ctx.handle, self.handle, C.byref(rays.struct), float(ht), float(zref), float(max_seg),
                                                ptr(partition), ptr(wet), ptr(hyd)), self.ctx.handle)
        return wet, hyd

    def ray_march(self, rays, ht, zref, nparts, flags, out=None):
        rays.adopt_stream(self.ctx)
        ht = rays.table_height(ht)
        nparts = np.ascontiguousarray(nparts, dtype=np.int32)
        wet, hyd = out if out is not None else rays.empty_outputs()
        rays.check_outputs(wet, hyd)
        check(self.ctx.lib.rdr_ray_march(self.ctx.handle, self.handle, C.byref(rays.struct), float(ht), float(zref), ptr(nparts),
                                         int(flags), ptr(wet), ptr(hyd)), self.ctx.handle)
        return wet, hyd

    def raytrace(self, rays, ht, zref, max_seg=1000.0, out=None, want_nparts=True):
        """One slice of _build_cube_ray (delay.py:256-323).  Returns (wet, hydro, nparts, flags);
        nparts/flags are None when want_nparts is False (fully asynchronous for device arrays).
        A batch with per-ray origin heights (Rays.grid(..., hts=...)) takes ht=None; nparts then has one entry per level of the
        table built for the LOWEST ray (ray_levels(rays.ht_min, zref))."""
        rays.adopt_stream(self.ctx)
        ht = rays.table_height(ht)
        wet, hyd = out if out is not None else rays.empty_outputs()
        rays.check_outputs(wet, hyd)
        if want_nparts:
            K = len(self.ray_levels(ht, zref)[0])
            nparts = np.zeros(K, dtype=np.int32)
            flags = C.c_int32()
            check(self.ctx.lib.rdr_raytrace(self.ctx.handle, self.handle, C.byref(rays.struct), float(ht), float(zref), float(max_seg),
                                            ptr(wet), ptr(hyd), ptr(nparts), C.byref(flags)), self.ctx.handle)
            return wet, hyd, nparts, flags.value
        check(self.ctx.lib.rdr_raytrace(self.ctx.handle, self.handle, C.byref(rays.struct), float(ht), float(zref), float(max_seg),
                                        ptr(wet), ptr(hyd), None, None), self.ctx.handle)
        return wet, hyd, None, None

    def raytrace_slices(self, rays, hts, zref, max_seg=1000.0, out=None, want_partition=True, want_nan=False):
        """The height loop of _build_cube_ray (delay.py:256-323) in one launch pair: every slice hts[s] of the same origins is
        integrated exactly as raytrace() would integrate it alone (own level table, slice maxima, nParts, z-clamp decision).
        `rays`: a GRID / LLH batch; when built with slices=S its look-vector / incidence arrays hold one block per slice.
        Returns (wet[S,...], hydro[S,...], K[S], nparts[S, nz-1], flags[S]); the last three are None when want_partition is
        False (fully asynchronous for device arrays).  K[s] == 0: build_ray -> None for that slice (its delays are 0).
        want_nan=True: a sixth element, bool[S] = np.isnan(result[s]).any() as scanned on the device before the download."""
        rays.adopt_stream(self.ctx)
        if rays.ht_min is not None:
            raise ValueError('a batch with per-ray heights is ONE slice: use raytrace()')
        hts = f64(np.atleast_1d(hts)).ravel()
        S = hts.size
        if rays.slices not in (0, S):
            raise ValueError(f'the ray batch carries look vectors for {rays.slices} slices, {S} heights were given')
        if out is not None:
            wet, hyd = out
        elif rays._torch_device is not None:
            import torch
            wet = torch.empty((S,) + tuple(rays.shape), dtype=torch.float64, device=rays._torch_device)
            hyd = torch.empty_like(wet)
        else:
            wet = _pinned.empty((S,) + tuple(rays.shape)); hyd = _pinned.empty((S,) + tuple(rays.shape))
        rays.check_outputs(wet, hyd, slices=S)
        ld = self.shape[2] - 1
        if want_partition:
            K = np.zeros(S, dtype=np.int32); nparts = np.zeros((S, ld), dtype=np.int32); flags = np.zeros(S, dtype=np.int32)
            check(self.ctx.lib.rdr_raytrace_slices(self.ctx.handle, self.handle, C.byref(rays.struct), ptr(hts), S, int(rays.slices > 0), float(zref),
                                                   float(max_seg), ptr(wet), ptr(hyd), ptr(K), ptr(nparts), ld, ptr(flags)), self.ctx.handle)
            # RDR_FLAG_NAN_OUTPUT (np.isnan(result).any() per slice, scanned on the device before the download) is reported apart
            # from the partition flags, which stay what rdr_raytrace returns for the slice
            nan_out = (flags & L.FLAG_NAN_OUTPUT) != 0
            flags &= ~np.int32(L.FLAG_NAN_OUTPUT)
            return (wet, hyd, K, nparts, flags, nan_out) if want_nan else (wet, hyd, K, nparts, flags)
        check(self.ctx.lib.rdr_raytrace_slices(self.ctx.handle, self.handle, C.byref(rays.struct), ptr(hts), S, int(rays.slices > 0), float(zref),
                                               float(max_seg), ptr(wet), ptr(hyd), None, None, ld, None), self.ctx.handle)
        return wet, hyd, None, None, None

    def raytrace_slices_to_cube(self, rays, hts, zref, max_seg=1000.0):
        """raytrace_slices() whose delays stay on the device: (Cube with axes (ypts, xpts, hts), K[S], nparts[S, nz-1], flags[S]) -
        the intermediate cube of tropo_delay's point branch for a ray-traced line of sight.  GRID batches only."""
        rays.adopt_stream(self.ctx)
        if rays.ht_min is not None:
            raise ValueError('a batch with per-ray heights is ONE slice: use raytrace()')
        hts = f64(np.atleast_1d(hts)).ravel()
        S = hts.size
        if rays.slices not in (0, S):
            raise ValueError(f'the ray batch carries look vectors for {rays.slices} slices, {S} heights were given')
        ld = self.shape[2] - 1
        K = np.zeros(S, dtype=np.int32); nparts = np.zeros((S, ld), dtype=np.int32); flags = np.zeros(S, dtype=np.int32)
        h = C.c_void_p()
        check(self.ctx.lib.rdr_raytrace_slices_to_cube(self.ctx.handle, self.handle, C.byref(rays.struct), ptr(hts), S, int(rays.slices > 0), float(zref),
                                                       float(max_seg), ptr(K), ptr(nparts), ld, ptr(flags), C.byref(h)), self.ctx.handle)
        flags &= ~np.int32(L.FLAG_NAN_OUTPUT)
        return Cube._from_handle(self.ctx, h), K, nparts, flags

    def __del__(self):
        try:
            if getattr(self, 'handle', None):
                self.ctx.lib.rdr_cube_destroy(self.handle)
                self.handle = None
        except Exception:
            pass


class Rays:
    """One ray batch = one (ny,nx) slice at one height (delay.py:256-273).  Keeps references to the
    arrays it points at."""

    def __init__(self):
        self.struct = L.RdrRays()
        self._keep = []
        self.shape = ()
        self._torch_device = None
        self._keep_tensor = None
        self._has_host = False
        self.slices = 0          # > 0: the look-vector / incidence / heading arrays hold one block of n rays per height slice
        self.ht_min = None       # per-ray origin heights (hts=...): their minimum = the height the batch's level table is built for

    def _set(self, field, arr):
        """Every array of a batch lives in ONE place (the C struct has a single `loc`): all NumPy, or all tensors on one GPU.
        A mixed batch gets its NumPy arrays uploaded to the tensors' device."""
        if arr is None:
            return
        if _is_dev(arr):
            import torch
            if arr.dtype != torch.float64:
                raise TypeError(f'ray arrays must be float64 (got {arr.dtype} for {field})')
            if not arr.is_cuda:
                arr = arr.numpy()                      # a CPU tensor is a host array (uploaded below when the batch lives on a GPU)
            elif self._torch_device is not None and arr.device != self._torch_device:
                raise ValueError(f'ray arrays live on different devices ({self._torch_device} and {arr.device})')
            else:
                if self._has_host:
                    self._upload_host_fields(arr.device)
                self._torch_device = arr.device
                self._keep_tensor = arr
                if not arr.is_contiguous():
                    arr = arr.contiguous()
                self._keep.append(arr)
                setattr(self.struct, field, arr.data_ptr())
                return
        a = f64(arr)
        if self._torch_device is not None:
            import torch
            t = torch.from_numpy(a).to(self._torch_device)
            self._keep.append(t)
            setattr(self.struct, field, t.data_ptr())
            return
        self._has_host = True
        self._keep.append((field, a))
        setattr(self.struct, field, a.ctypes.data)

    def _upload_host_fields(self, device):
        import torch
        kept = []
        for item in self._keep:
            if isinstance(item, tuple):
                field, a = item
                t = torch.from_numpy(a).to(device)
                setattr(self.struct, field, t.data_ptr())
                kept.append(t)
            else:
                kept.append(item)
        self._keep = kept
        self._has_host = False

    def check_outputs(self, *outs, slices=1):
        """Output arrays must live where the batch lives (a NumPy output handed to a device batch would be written through a
        host pointer by the kernels) and hold one float64 per ray (and slice)."""
        want = int(self.struct.n) * int(slices)
        for o in outs:
            if o is None:
                continue
            have = o.numel() if _is_dev(o) else np.size(o)
            if have != want:
                raise ValueError(f'output arrays must hold {want} values for this ray batch (got {have})')
            dev = _is_dev(o) and getattr(o, 'is_cuda', False)
            if dev != (self._torch_device is not None):
                raise ValueError('output arrays must be ' + ('tensors on ' + str(self._torch_device) if self._torch_device is not None else 'NumPy arrays') +
                                 ' for this ray batch')
            if dev and (o.device != self._torch_device or not o.is_contiguous() or str(o.dtype) != 'torch.float64'):
                raise ValueError('output tensors must be contiguous float64 and on the device of the ray batch')
            if not dev and not (isinstance(o, np.ndarray) and o.dtype == np.float64 and o.flags.c_contiguous):
                raise ValueError('output arrays must be C-contiguous float64 NumPy arrays')

    @classmethod
    def grid(cls, xpts, ypts, los=None, inc=None, hd=None, zenith=False, slices=0, hts=None):
        """Origins on meshgrid(xpts, ypts) (delay.py:242).  LOS: `los` (ny,nx,3) ECEF unit vectors, or
        inc/hd (scalars or (ny,nx) arrays, degrees), or zenith.  slices=S: los / inc / hd carry a leading slice axis of
        length S (look vectors that depend on the slice height; Cube.raytrace_slices).
        hts=(ny,nx): PER-PIXEL origin heights (a scene on a DEM) instead of one slice height - no reference semantics, the rule
        is in include/raider_hip.h (rdr_rays.hts); then pass ht=None to the ray-tracing calls."""
        r = cls()
        r.slices = int(slices)
        nx = xpts.numel() if _is_dev(xpts) else np.size(xpts)
        ny = ypts.numel() if _is_dev(ypts) else np.size(ypts)
        r.struct.origin_mode = L.ORIGIN_GRID
        r.struct.nx, r.struct.ny, r.struct.n = nx, ny, nx * ny
        r._set('xpts', xpts); r._set('ypts', ypts)
        r.shape = (ny, nx)
        r._set_heights(hts)
        r._set_los(los, inc, hd, zenith)
        return r

    @classmethod
    def points(cls, lat=None, lon=None, xyz=None, los=None, inc=None, hd=None, zenith=False, hts=None):
        """Arbitrary ray list: lat/lon (deg) at the slice height (or at per-ray heights hts[n]), or ECEF xyz (n,3)."""
        r = cls()
        if xyz is not None:
            r.struct.origin_mode = L.ORIGIN_XYZ
            n = (xyz.numel() if _is_dev(xyz) else np.size(xyz)) // 3
            r.shape = tuple(xyz.shape[:-1])
            r._set('xyz', xyz)
            r._set('lat', lat); r._set('lon', lon)
        else:
            r.struct.origin_mode = L.ORIGIN_LLH
            n = lat.numel() if _is_dev(lat) else np.size(lat)
            r.shape = tuple(lat.shape) if hasattr(lat, 'shape') else (n,)
            r._set('lat', lat); r._set('lon', lon)
        r.struct.n = n
        r._set_heights(hts)
        r._set_los(los, inc, hd, zenith)
        return r

    def _set_heights(self, hts):
        """Per-ray origin heights: one float64 per ray; their minimum (one reduction, and for device arrays one read-back, at
        construction) is the height the batch's level table is built for."""
        if hts is None:
            return
        cnt = hts.numel() if _is_dev(hts) else np.size(hts)
        if cnt != self.struct.n:
            raise ValueError(f'per-ray heights must have shape {tuple(self.shape)} (got {cnt} values for {self.struct.n} rays)')
        if self.slices:
            raise ValueError('per-ray heights and height slices exclude each other')
        self.ht_min = float(hts.min()) if cnt else 0.0
        if self.ht_min != self.ht_min:
            raise ValueError('per-ray heights contain NaN')
        self._set('hts', hts)

    def table_height(self, ht):
        """The `ht` argument of the C entry points for this batch: the slice height, or min(hts) for per-ray heights (an explicit
        ht below it is allowed: the same rays in a taller level table)."""
        if self.ht_min is None:
            if ht is None:
                raise ValueError('this ray batch has no per-ray heights: a slice height ht is needed')
            return float(ht)
        if ht is None:
            return self.ht_min
        if float(ht) > self.ht_min:
            raise ValueError(f'ht = {ht} lies above the lowest per-ray height {self.ht_min}')
        return float(ht)

    def _set_los(self, los, inc, hd, zenith):
        n = self.struct.n
        mult = max(self.slices, 1)
        if los is not None:
            cnt = los.numel() if _is_dev(los) else np.size(los)
            if cnt != 3 * n * mult:
                raise ValueError(f'look vectors must have shape {((mult,) if self.slices else ()) + tuple(self.shape) + (3,)}')
            self.struct.los_mode = L.LOS_VEC
            self._set('los', los)
        elif zenith:
            self.struct.los_mode = L.LOS_ZENITH
        elif inc is not None:
            if np.ndim(inc) == 0 and np.ndim(hd) == 0 and not _is_dev(inc):
                if float(inc) < 0:
                    raise ValueError('inc_hd_to_enu: Incidence angle cannot be less than 0')
                self.struct.los_mode = L.LOS_INC_HD_SCALAR
                self.struct.inc0, self.struct.hd0 = float(inc), float(hd)
            else:
                one_heading = np.ndim(hd) == 0 and not _is_dev(hd)          # incidence raster + one heading: no heading array
                full = ((mult,) if self.slices else ()) + tuple(self.shape)
                if not _is_dev(inc):
                    inc = np.broadcast_to(np.asarray(inc, dtype=np.float64), full)
                    if not one_heading and not _is_dev(hd):
                        hd = np.broadcast_to(np.asarray(hd, dtype=np.float64), full)
                    if inc.min() < 0:
                        raise ValueError('inc_hd_to_enu: Incidence angle cannot be less than 0')
                self.struct.los_mode = L.LOS_INC_HD
                self._set('inc', inc)
                if one_heading:
                    self.struct.hd0 = float(hd)
                else:
                    self._set('hd', hd)
        else:
            raise ValueError('a ray batch needs look vectors, inc/heading, or zenith=True')
        if self._torch_device is not None and self._has_host:
            self._upload_host_fields(self._torch_device)
        self.struct.loc = L.RDR_DEVICE if self._torch_device is not None else L.RDR_HOST

    def adopt_stream(self, ctx):
        """Device-resident batches run on torch's current stream (so they are ordered with the producer / consumer of
        the tensors); host batches run on the context's private stream."""
        if self._keep_tensor is not None:
            ctx.adopt_torch_stream(self._keep_tensor)
        else:
            ctx.set_stream(-1)

    def empty_outputs(self):
        if self._torch_device is not None:
            import torch
            return (torch.empty(self.shape, dtype=torch.float64, device=self._torch_device),
                    torch.empty(self.shape, dtype=torch.float64, device=self._torch_device))
        return _pinned.empty(tuple(self.shape)), _pinned.empty(tuple(self.shape))       # (large batches: recycled page-locked results)

    def look_vectors(self, ctx=None):
        ctx = ctx or Context.default()
        self.adopt_stream(ctx)
        if self._torch_device is not None:
            import torch
            out = torch.empty(self.shape + (3,), dtype=torch.float64, device=self._torch_device)
        else:
            out = np.empty(self.shape + (3,))
        check(ctx.lib.rdr_look_vectors(ctx.handle, C.byref(self.struct), 0.0, ptr(out)), ctx.handle)
        return out


def nparts_from_maxlen(maxlen, max_seg=1000.0):
    """delay.py:283."""
    maxlen = f64(maxlen)
    out = np.empty(maxlen.size, dtype=np.int32)
    check(L.load().rdr_nparts(ptr(maxlen), maxlen.size, float(max_seg), ptr(out)))
    return out


def torch_device_or_none():
    """The torch device of the default context when torch with a GPU is importable, else None (NumPy-only callers never need torch)."""
    try:
        import torch
    except ImportError:
        return None
    if not torch.cuda.is_available():
        return None
    import os
    idx = int(os.environ.get('RAIDER_HIP_DEVICE', os.environ.get('LOCAL_RANK', '-1')))
    return torch.device('cuda', idx if idx >= 0 else torch.cuda.current_device())


def lla2ecef_device(lat, lon, h, ctx=None):
    """utilFcns.lla2ecef on device tensors (same kernel, no host round trip): broadcastable float64 tensors -> (..., 3) ECEF."""
    import torch
    ctx = ctx or Context.default()
    lat, lon, h = torch.broadcast_tensors(lat, lon, h)
    lat, lon, h = (_dev_f64(t.contiguous(), 'lla2ecef_device input') for t in (lat, lon, h))
    ctx.adopt_torch_stream(lat)
    out = torch.empty(tuple(lat.shape) + (3,), dtype=torch.float64, device=lat.device)
    check(ctx.lib.rdr_lla2ecef(ctx.handle, ptr(lat), ptr(lon), ptr(h), lat.numel(), ptr(out), L.RDR_DEVICE), ctx.handle)
    return out
