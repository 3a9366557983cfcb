"""GUNW radian conversion of delay cubes (reference: tools/RAiDER/aria/calcGUNW.py:26-108).

`compute_delays_slc` mirrors the reference function of the same name for the part that is arithmetic: the delays of the
reference and the secondary date are multiplied by `phase2range = -4 pi / wavelength` (calcGUNW.py:54-59) - on the GPU
through `rdr_delays_to_phase` - renamed to the GUNW layer names, given the GUNW attributes and float32 coordinates
(calcGUNW.py:75-108).  Writing the result INTO a GUNW product (`update_gunw_slc`, HDF5 through h5py/netCDF4) is outside
the hot path and not provided.
"""
import datetime as dt
from pathlib import Path

import numpy as np

from . import _lib as L
from ._lib import Context, check, ptr

TROPO_GROUP = 'science/grids/corrections/external/troposphere'
TROPO_NAMES = ['troposphereWet', 'troposphereHydrostatic']
DIM_NAMES = ['heightsMeta', 'latitudeMeta', 'longitudeMeta']


def delays_to_phase(wet, hydro, wavelength, ctx=None):
    """(wet, hydro) delays [m] -> phase [rad]: each times -4 pi / wavelength (calcGUNW.py:54-59), dtype preserved
    (float32 stays float32 as with a Python-float factor in NumPy, everything else is computed in float64).
    NumPy arrays in -> NumPy arrays out; torch tensors on the GPU in -> new tensors on the same device."""
    ctx = ctx or Context.default()
    if hasattr(wet, 'data_ptr'):
        import torch
        dtype = wet.dtype if wet.dtype == torch.float32 else torch.float64
        w = wet.to(dtype).contiguous(); h = hydro.to(dtype).contiguous()
        if w.shape != h.shape:
            raise ValueError('wet and hydro must have the same shape')
        ctx.adopt_torch_stream(w)
        ow, oh = torch.empty_like(w), torch.empty_like(h)
        check(ctx.lib.rdr_delays_to_phase(ctx.handle, ptr(w), ptr(h), w.numel(), L.RDR_F32 if dtype == torch.float32 else L.RDR_F64,
                                          float(wavelength), ptr(ow), ptr(oh), L.RDR_DEVICE), ctx.handle)
        return ow, oh
    wet = np.asarray(wet); hydro = np.asarray(hydro)
    dtype = np.float32 if wet.dtype == np.float32 and hydro.dtype == np.float32 else np.float64
    w = np.ascontiguousarray(wet, dtype=dtype); h = np.ascontiguousarray(hydro, dtype=dtype)
    if w.shape != h.shape:
        raise ValueError('wet and hydro must have the same shape')
    ow, oh = np.empty_like(w), np.empty_like(h)
    check(ctx.lib.rdr_delays_to_phase(ctx.handle, ptr(w), ptr(h), w.size, L.RDR_F32 if dtype == np.float32 else L.RDR_F64,
                                      float(wavelength), ptr(ow), ptr(oh), L.RDR_HOST), ctx.handle)
    return ow, oh


def _file_attrs(path):
    """Global attributes of a delay-cube file (NetCDF-3 through scipy, NetCDF-4 through h5lite)."""
    with open(path, 'rb') as fh:
        magic = fh.read(4)
    if magic[:3] == b'CDF':
        from scipy.io import netcdf_file
        with netcdf_file(path, 'r', mmap=False) as f:
            return {k: (v.decode() if isinstance(v, bytes) else v) for k, v in f._attributes.items()}
    from . import h5lite
    return dict(h5lite.File(path).attrs.items())


def _open_cube(item):
    """A delay cube as (variables, attrs): a path to a file written by DelayCube.to_netcdf / the reference, or an object
    with `.variables` / `.attrs` (DelayCube, xarray.Dataset)."""
    if isinstance(item, (str, Path)):
        from .delayFcns import _read_cube_file
        return _read_cube_file(str(item)), _file_attrs(str(item))
    return item.variables, dict(getattr(item, 'attrs', {}))


def compute_delays_slc(cube_paths, wavelength, ctx=None):
    """calcGUNW.compute_delays_slc: `cube_paths` = the two delay-cube files of a GUNW's reference and secondary dates, named
    `<model>_tropo_<YYYYmmddTHHMMSS>_...nc` (the date is the third `_` field, calcGUNW.py:44-46); the LATER date is the
    reference (calcGUNW.py:48: `sec, ref = sorted(...)`).  Alternatively a dict {datetime: cube or path}.
    Returns a DelayCube-like object whose variables are `reference_/secondary_troposphereWet/Hydrostatic` [rad] on float32
    coordinates `heightsMeta, latitudeMeta, longitudeMeta`, with the reference's per-layer attributes."""
    from .delay import DelayCube
    if isinstance(cube_paths, dict):
        dct = dict(cube_paths)
        model = None
    else:
        dct = {}
        for path in cube_paths:
            path = Path(path)
            dct[dt.datetime.strptime(path.name.split('_')[2], '%Y%m%dT%H%M%S')] = path
        model = Path(list(cube_paths)[-1]).name.split('_')[0]
    if len(dct) != 2:
        raise ValueError('compute_delays_slc needs the delay cubes of exactly two dates')
    sec, ref = sorted(dct.keys())

    out, layer_attrs = {}, {}
    coords = None
    for key, date in (('reference', ref), ('secondary', sec)):
        v, attrs = _open_cube(dct[date])
        wet, hyd = delays_to_phase(np.asarray(v['wet'][:]), np.asarray(v['hydro'][:]), wavelength, ctx=ctx)
        out[f'{key}_{TROPO_NAMES[0]}'] = wet
        out[f'{key}_{TROPO_NAMES[1]}'] = hyd
        for name in TROPO_NAMES:
            # `name.lstrip('troposphere')` strips CHARACTERS, not the prefix (calcGUNW.py:91): 'Wet' -> 'Wet', 'Hydrostatic' -> 'Hydrostatic'
            layer_attrs[f'{key}_{name}'] = {
                'units': 'radians', 'grid_mapping': 'crs',
                'description': f"Delay due to {name.lstrip('troposphere')} component of troposphere",
                'long_name': name, 'standard_name': name,
                'model_times_used': attrs.get('model_times_used'),
                'scene_center_time': attrs.get('reference_time'),
                'time_interpolation_method': attrs.get('interpolation_method'),
            }
        # the reference copies the coordinates of the LAST file opened (the secondary), calcGUNW.py:69
        coords = {d: np.asarray(v[c][:], dtype=np.float32) for d, c in zip(DIM_NAMES, ('z', 'y', 'x'))}   # calcGUNW.py:104-108
    out.update(coords)
    ds = DelayCube(out, {'model': model, 'method': 'ray tracing'})
    ds.layer_attrs = layer_attrs
    return ds
