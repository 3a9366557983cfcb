"""Flat-binary rasters without GDAL: what `utilFcns.rio_open` (utilFcns.py:164-202) reads for the delay path.

The reference opens its line-of-sight rasters (ISCE `los.rdr`: band 1 incidence, band 2 heading; losreader.py:116-121) and the
lat / lon / height rasters of a radar-geometry AOI through rasterio = GDAL.  Every such file in the reference's tests and in
ISCE's products is a headerless binary array described by a side-car:

  * `<file>.vrt`  GDAL virtual raster: `VRTRawRasterBand` (ImageOffset / PixelOffset / LineOffset / ByteOrder on a raw source
                  file) or a plain band with a `SimpleSource` pointing at a raw file that has its own ENVI header / VRT;
  * `<file>.hdr` / `<stem>.hdr`  ENVI header: samples, lines, bands, data type, interleave (bsq / bil / bip), byte order,
                  header offset.

`rio_open` prefers `<file>.vrt` when it exists; so does this reader.  GeoTIFF, NetCDF sub-datasets and everything else GDAL
reads are NOT handled here (rasterio is used for them when it is installed).  Returns what rasterio returns: `read()` gives
(bands, rows, cols) in the file's element type; `profile` carries width / height / count / dtype / nodata."""
import re
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

_GDAL_TYPES = dict(Byte='u1', Int8='i1', UInt16='u2', Int16='i2', UInt32='u4', Int32='i4', UInt64='u8', Int64='i8', Float32='f4', Float64='f8',
                   CFloat32='c8', CFloat64='c16')
_ENVI_TYPES = {1: 'u1', 2: 'i2', 3: 'i4', 4: 'f4', 5: 'f8', 6: 'c8', 9: 'c16', 12: 'u2', 13: 'u4', 14: 'i8', 15: 'u8'}


class NotARawRaster(OSError):
    """The file has no VRT / ENVI description this reader understands (an OSError, like rasterio's RasterioIOError: the reference's
    callers catch OSError to fall back to other interpretations of the file, losreader.py:122)."""


def _envi_header(path):
    txt = Path(path).read_text(errors='replace')
    if not txt.lstrip().upper().startswith('ENVI'):
        raise NotARawRaster(f'{path}: not an ENVI header')
    txt = re.sub(r'\{[^}]*\}', lambda m: m.group(0).replace('\n', ' '), txt)          # multi-line {...} values on one line
    kv = {}
    for line in txt.splitlines()[1:]:
        if '=' in line:
            k, v = line.split('=', 1)
            kv[k.strip().lower()] = v.strip()
    try:
        meta = dict(width=int(kv['samples']), height=int(kv['lines']), count=int(kv.get('bands', 1)), offset=int(kv.get('header offset', 0)),
                    interleave=kv.get('interleave', 'bsq').lower(), dtype=_ENVI_TYPES[int(kv['data type'])],
                    order='>' if int(kv.get('byte order', 0)) == 1 else '<')
    except (KeyError, ValueError) as e:
        raise NotARawRaster(f'{path}: incomplete ENVI header ({e})')
    nd = kv.get('data ignore value')
    meta['nodata'] = float(nd) if nd not in (None, '') else None
    return meta


def _read_envi(data_path, hdr_path):
    m = _envi_header(hdr_path)
    dt = np.dtype(m['order'] + m['dtype'])
    w, h, n = m['width'], m['height'], m['count']
    raw = np.fromfile(data_path, dtype=dt, count=w * h * n, offset=m['offset'])
    if raw.size != w * h * n:
        raise NotARawRaster(f'{data_path}: {raw.size} elements, the header describes {n} x {h} x {w}')
    if m['interleave'] == 'bsq':
        data = raw.reshape(n, h, w)
    elif m['interleave'] == 'bil':
        data = raw.reshape(h, n, w).transpose(1, 0, 2)
    elif m['interleave'] == 'bip':
        data = raw.reshape(h, w, n).transpose(2, 0, 1)
    else:
        raise NotARawRaster(f'{hdr_path}: unknown interleave {m["interleave"]!r}')
    return np.ascontiguousarray(data).astype(dt.newbyteorder('='), copy=False), [m['nodata']] * n


def _find_envi_header(path):
    path = Path(path)
    for cand in (Path(str(path) + '.hdr'), path.with_suffix('.hdr')):
        if cand.exists():
            return cand
    return None


def _read_vrt(vrt_path):
    vrt_path = Path(vrt_path)
    try:
        root = ET.parse(vrt_path).getroot()
    except ET.ParseError as e:
        raise NotARawRaster(f'{vrt_path}: not a VRT ({e})')
    if root.tag != 'VRTDataset':
        raise NotARawRaster(f'{vrt_path}: not a VRT')
    w, h = int(root.get('rasterXSize')), int(root.get('rasterYSize'))
    bands, nodata = [], []
    for b in root.findall('VRTRasterBand'):
        dt_name = b.get('dataType', 'Byte')
        if dt_name not in _GDAL_TYPES:
            raise NotARawRaster(f'{vrt_path}: unsupported dataType {dt_name}')
        nd = b.findtext('NoDataValue')
        nodata.append(float(nd) if nd not in (None, '') else None)

        def source(el):
            name = el.findtext('SourceFilename')
            rel = el.find('SourceFilename').get('relativeToVRT', '0') == '1'
            return (vrt_path.parent / name) if rel else Path(name)
        if b.get('subClass') == 'VRTRawRasterBand':
            src = source(b)
            order = '>' if (b.findtext('ByteOrder') or 'LSB').upper() == 'MSB' else '<'
            dt = np.dtype(order + _GDAL_TYPES[dt_name])
            off = int(b.findtext('ImageOffset') or 0)
            px = int(b.findtext('PixelOffset') or dt.itemsize)
            ln = int(b.findtext('LineOffset') or px * w)
            need = off + (h - 1) * ln + (w - 1) * px + dt.itemsize
            buf = np.fromfile(src, dtype=np.uint8)
            if buf.size < need:
                raise NotARawRaster(f'{src}: {buf.size} bytes, band {b.get("band")} of {vrt_path.name} needs {need}')
            arr = np.lib.stride_tricks.as_strided(buf[off:], shape=(h, w, dt.itemsize), strides=(ln, px, 1))
            bands.append(np.ascontiguousarray(arr).view(dt)[..., 0].astype(dt.newbyteorder('='), copy=False))
            continue
        ss = b.find('SimpleSource')
        if ss is None:
            ss = b.find('ComplexSource')
        if ss is None:
            raise NotARawRaster(f'{vrt_path}: band without a raw / simple source')
        src = source(ss)
        sb = int(ss.findtext('SourceBand') or 1)
        for rect in ('SrcRect', 'DstRect'):
            r = ss.find(rect)
            if r is not None and (float(r.get('xOff', 0)) != 0 or float(r.get('yOff', 0)) != 0 or float(r.get('xSize', w)) != w or float(r.get('ySize', h)) != h):
                raise NotARawRaster(f'{vrt_path}: windowed / resampled sources are not supported')
        inner_vrt = Path(str(src) + '.vrt')
        hdr = _find_envi_header(src)
        if hdr is not None:
            data, _ = _read_envi(src, hdr)
        elif inner_vrt.exists() and inner_vrt.resolve() != vrt_path.resolve():
            data, _ = _read_vrt(inner_vrt)
        else:                                            # headerless: the SourceProperties / band type describe a one-band BSQ file
            sp = ss.find('SourceProperties')
            dt = np.dtype('<' + _GDAL_TYPES[(sp.get('DataType') if sp is not None else None) or dt_name])
            raw = np.fromfile(src, dtype=dt)
            if raw.size % (w * h) != 0 or raw.size == 0:
                raise NotARawRaster(f'{src}: size does not match {h} x {w}')
            data = raw.reshape(-1, h, w)
        if sb > data.shape[0] or data.shape[1:] != (h, w):
            raise NotARawRaster(f'{vrt_path}: source {src.name} is {data.shape}, band {sb} of {h} x {w} wanted')
        bands.append(data[sb - 1].astype(np.dtype(_GDAL_TYPES[dt_name]), copy=False))
    if not bands:
        raise NotARawRaster(f'{vrt_path}: no bands')
    return np.stack(bands, axis=0), nodata


class RawRaster:
    """The part of a rasterio dataset the delay path uses: `read()` / `read(band)`, `profile`, `nodatavals`, context manager."""

    def __init__(self, path):
        path = Path(path)
        if not path.exists():
            raise FileNotFoundError(f'{path}: No such file or directory')
        if path.suffix.lower() == '.vrt':
            self._data, self.nodatavals = _read_vrt(path)
        elif Path(str(path) + '.vrt').exists():
            self._data, self.nodatavals = _read_vrt(Path(str(path) + '.vrt'))
        else:
            hdr = _find_envi_header(path)
            if hdr is None or path.suffix.lower() == '.hdr':
                raise NotARawRaster(f'{path}: no .vrt or ENVI .hdr beside it (GeoTIFF / NetCDF rasters need rasterio)')
            self._data, self.nodatavals = _read_envi(path, hdr)
        self.nodatavals = tuple(self.nodatavals)
        n, h, w = self._data.shape
        self.profile = dict(driver='RAW', dtype=str(self._data.dtype), nodata=self.nodatavals[0], width=w, height=h, count=n, crs=None, transform=None)
        self.count, self.height, self.width = n, h, w

    def read(self, band=None):
        return self._data.copy() if band is None else self._data[band - 1].copy()

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


def open_raster(path):
    """rasterio.open when rasterio is installed (the reference's reader), else the built-in raw-raster reader."""
    try:
        import rasterio
    except ImportError:
        return RawRaster(path)
    return rasterio.open(path)


def rio_open(path, userNDV=None, band=None):
    """utilFcns.py:164-202: (data, profile); `<path>.vrt` is preferred when it exists; all bands squeezed unless `band` is given.
    (The reference's nodataToNan works on a private copy - utilFcns.py:205-210, `inarr.astype(float)` - so no-data values come
    back unchanged there; they do here too.)"""
    path = Path(path)
    vrt = path.with_suffix(path.suffix + '.vrt')
    if vrt.exists():
        path = vrt
    with open_raster(path) as src:
        profile = src.profile
        data = src.read(band).squeeze() if band is not None else src.read().squeeze()
    return np.array(data), profile


_ENVI_CODE = {'uint8': 1, 'int16': 2, 'int32': 3, 'float32': 4, 'float64': 5, 'complex64': 6, 'complex128': 9, 'uint16': 12, 'uint32': 13, 'int64': 14, 'uint64': 15}


def _envi_map_info(proj, x0, y0, dx, dy):
    """The header's `map info` for the raster's CRS: geographic WGS 84 (None / EPSG:4326), a WGS 84 UTM zone (EPSG:326xx / 327xx),
    or None when the CRS has no ENVI spelling known here (the header then says nothing rather than something wrong)."""
    code = None
    if proj is None:
        code = 4326
    else:
        txt = str(getattr(proj, 'to_epsg', lambda: None)() or proj).strip().upper().replace('EPSG:', '')
        if txt.isdigit():
            code = int(txt)
        elif 'WGS 84' in txt and 'UTM' not in txt and 'PROJ' not in txt.replace('PROJCS', 'PROJ') and ('GEOGCS' in txt or 'GEOGCRS' in txt):
            code = 4326
    tie = f'1, 1, {x0:.15g}, {y0:.15g}, {abs(dx):.15g}, {abs(dy):.15g}'
    if code == 4326:
        return f'map info = {{Geographic Lat/Lon, {tie}, WGS-84}}'
    if code is not None and (32601 <= code <= 32660 or 32701 <= code <= 32760):
        return f'map info = {{UTM, {tie}, {code % 100}, {"North" if code < 32700 else "South"}, WGS-84}}'
    return None


def write_envi(array, path, nodata=None, geotransform=None, description=None, proj=None):
    """One band as an ENVI raster: `<path>` flat binary (native little-endian) + `<path with .hdr>` - the `fmt='ENVI'` product of
    utilFcns.writeArrayToRaster (utilFcns.py:257-304) without GDAL.  geotransform: GDAL's 6 numbers (x0, dx, 0, y0, 0, dy); written as
    the header's `map info` (pixel-corner tie point, as GDAL writes it)."""
    path = Path(path)
    a = np.ascontiguousarray(array)
    if a.ndim != 2:
        raise RuntimeError(f'writeArrayToRaster: cannot write an array of shape {np.shape(array)} to a raster image')
    a = a.astype(a.dtype.newbyteorder('<'), copy=False)
    if str(a.dtype) not in _ENVI_CODE:
        raise TypeError(f'no ENVI data type for {a.dtype}')
    a.tofile(path)
    hdr = ['ENVI', f'description = {{{description or path.name}}}', f'samples = {a.shape[1]}', f'lines   = {a.shape[0]}', 'bands   = 1', 'header offset = 0',
           'file type = ENVI Standard', f'data type = {_ENVI_CODE[str(a.dtype)]}', 'interleave = bsq', 'byte order = 0']
    if geotransform is not None:
        x0, dx, rx, y0, ry, dy = (float(v) for v in geotransform)
        if rx != 0.0 or ry != 0.0:
            raise ValueError('rotated geotransforms have no ENVI map info')
        mi = _envi_map_info(proj, x0, y0, dx, dy)
        if mi is not None:
            hdr.append(mi)
        elif isinstance(proj, str) and ('[' in proj):                    # a WKT: GDAL reads it back from this key
            hdr.append(f'coordinate system string = {{{proj}}}')
    if nodata is not None:
        hdr.append(f'data ignore value = {float(nodata):.15g}')
    hdr_path = path.with_suffix('.hdr') if path.suffix else Path(str(path) + '.hdr')
    hdr_path.write_text('\n'.join(hdr) + '\n')
    return str(path)
