"""raider_amd.rawraster: the flat-binary rasters `rio_open` reads for the delay path (ISCE los.rdr / lat.rdr / lon.rdr with a GDAL
.vrt or an ENVI .hdr side-car) without GDAL.  Pinned on the radar-geometry rasters the reference's tests hold
(test/scenario_4/lat.rdr, lon.rdr: the statistics GDAL itself wrote into their .vrt) and on every layout written here."""
from pathlib import Path

import numpy as np
import pytest

from raider_amd import rawraster
from raider_amd.rawraster import NotARawRaster, RawRaster, rio_open

S4 = Path(__file__).parent / 'golden' / 'ref_files' / 'scenario_4'


def test_reference_geometry_rasters_match_the_statistics_gdal_recorded():
    """lat.rdr.vrt / lon.rdr.vrt carry STATISTICS_MINIMUM / MAXIMUM / MEAN / STDDEV computed by GDAL over the valid pixels (NoDataValue
    0): an independent pin of shape, element type, byte order and band selection."""
    import xml.etree.ElementTree as ET
    for name in ('lat.rdr', 'lon.rdr'):
        data, prof = rio_open(S4 / name)
        assert data.shape == (45, 226) and data.dtype == np.float64 and prof['width'] == 226 and prof['height'] == 45 and prof['count'] == 1
        assert prof['nodata'] == 0.0
        md = {m.get('key'): float(m.text) for m in ET.parse(S4 / (name + '.vrt')).getroot().iter('MDI') if m.get('key', '').startswith('STATISTICS')}
        valid = data[data != 0.0]
        assert abs(valid.min() - md['STATISTICS_MINIMUM']) < 1e-9 and abs(valid.max() - md['STATISTICS_MAXIMUM']) < 1e-9
        assert abs(valid.mean() - md['STATISTICS_MEAN']) < 1e-9 and abs(valid.std() - md['STATISTICS_STDDEV']) < 1e-9
        # the ENVI header beside it describes the same array
        hdr_only, _ = rawraster._read_envi(S4 / name, S4 / name.replace('.rdr', '.hdr'))
        assert np.array_equal(hdr_only[0], data)


@pytest.mark.parametrize('interleave', ['bsq', 'bil', 'bip'])
@pytest.mark.parametrize('order', ['<', '>'])
def test_envi_layouts(tmp_path, interleave, order):
    rng = np.random.default_rng(0)
    a = rng.uniform(20, 45, (2, 7, 11)).astype(np.float32)
    a[1] = rng.uniform(-180, 180, (7, 11))
    disk = {'bsq': a, 'bil': a.transpose(1, 0, 2), 'bip': a.transpose(1, 2, 0)}[interleave]
    p = tmp_path / 'los.rdr'
    with open(p, 'wb') as f:
        f.write(b'\0' * 16)                                               # header offset
        f.write(np.ascontiguousarray(disk).astype(order + 'f4').tobytes())
    (tmp_path / 'los.rdr.hdr').write_text(f'ENVI\ndescription = {{\n  two lines}}\nsamples = 11\nlines   = 7\nbands   = 2\nheader offset = 16\n'
                                          f'file type = ENVI Standard\ndata type = 4\ninterleave = {interleave}\nbyte order = {1 if order == ">" else 0}\n')
    data, prof = rio_open(p)
    assert data.dtype == np.float32 and np.array_equal(data, a) and prof['count'] == 2
    one, _ = rio_open(p, band=2)
    assert np.array_equal(one, a[1])


def test_vrt_raw_bands_and_simple_sources(tmp_path):
    rng = np.random.default_rng(1)
    a = rng.uniform(20, 45, (2, 5, 9)).astype(np.float32)
    (tmp_path / 'los.rdr').write_bytes(np.ascontiguousarray(a.transpose(1, 2, 0)).tobytes())       # pixel-interleaved, as ISCE writes los.rdr
    (tmp_path / 'los.rdr.vrt').write_text('''<VRTDataset rasterXSize="9" rasterYSize="5">
    <VRTRasterBand band="1" dataType="Float32" subClass="VRTRawRasterBand">
        <SourceFilename relativeToVRT="1">los.rdr</SourceFilename><ByteOrder>LSB</ByteOrder>
        <ImageOffset>0</ImageOffset><PixelOffset>8</PixelOffset><LineOffset>72</LineOffset>
    </VRTRasterBand>
    <VRTRasterBand band="2" dataType="Float32" subClass="VRTRawRasterBand">
        <SourceFilename relativeToVRT="1">los.rdr</SourceFilename><ByteOrder>LSB</ByteOrder>
        <ImageOffset>4</ImageOffset><PixelOffset>8</PixelOffset><LineOffset>72</LineOffset>
    </VRTRasterBand>
</VRTDataset>''')
    # a stale ENVI header beside it must NOT win: rio_open prefers <file>.vrt (utilFcns.py:171-173)
    (tmp_path / 'los.hdr').write_text('ENVI\nsamples = 9\nlines = 5\nbands = 2\ndata type = 4\ninterleave = bsq\nbyte order = 0\n')
    data, _ = rio_open(tmp_path / 'los.rdr')
    assert np.array_equal(data, a)
    # a VRT whose band is a SimpleSource on a headerless file (SourceProperties give the type)
    b = rng.uniform(0, 3000, (5, 9))
    (tmp_path / 'hgt.rdr').write_bytes(b.tobytes())
    (tmp_path / 'hgt.rdr.vrt').write_text('''<VRTDataset rasterXSize="9" rasterYSize="5"><VRTRasterBand dataType="Float64" band="1">
      <NoDataValue>-32768</NoDataValue><SimpleSource><SourceFilename relativeToVRT="1">hgt.rdr</SourceFilename><SourceBand>1</SourceBand>
      <SourceProperties RasterXSize="9" RasterYSize="5" DataType="Float64" BlockXSize="9" BlockYSize="1" />
      <SrcRect xOff="0" yOff="0" xSize="9" ySize="5" /><DstRect xOff="0" yOff="0" xSize="9" ySize="5" /></SimpleSource></VRTRasterBand></VRTDataset>''')
    data, prof = rio_open(tmp_path / 'hgt.rdr')
    assert np.array_equal(data, b) and prof['nodata'] == -32768.0
    # truncated file / no side-car / not a raster at all: OSError (what the reference's callers catch)
    (tmp_path / 'short.rdr').write_bytes(b'\0' * 10)
    (tmp_path / 'short.rdr.vrt').write_text((tmp_path / 'los.rdr.vrt').read_text().replace('los.rdr', 'short.rdr'))
    with pytest.raises(OSError):
        rio_open(tmp_path / 'short.rdr')
    (tmp_path / 'orbit.txt').write_text('2020-01-01T00:00:00 1 2 3 4 5 6\n')
    with pytest.raises(NotARawRaster):
        rio_open(tmp_path / 'orbit.txt')
    with pytest.raises(OSError):
        rio_open(tmp_path / 'missing.rdr')


def _los_raster(tmp_path, rng):
    inc = rng.uniform(30, 45, (6, 8)).astype(np.float32); hd = rng.uniform(-170, -165, (6, 8)).astype(np.float32)
    (tmp_path / 'los.rdr').write_bytes(np.ascontiguousarray(np.stack([inc, hd], -1)).tobytes())
    (tmp_path / 'los.rdr.vrt').write_text('''<VRTDataset rasterXSize="8" rasterYSize="6">
    <VRTRasterBand band="1" dataType="Float32" subClass="VRTRawRasterBand"><SourceFilename relativeToVRT="1">los.rdr</SourceFilename>
        <ByteOrder>LSB</ByteOrder><ImageOffset>0</ImageOffset><PixelOffset>8</PixelOffset><LineOffset>64</LineOffset></VRTRasterBand>
    <VRTRasterBand band="2" dataType="Float32" subClass="VRTRawRasterBand"><SourceFilename relativeToVRT="1">los.rdr</SourceFilename>
        <ByteOrder>LSB</ByteOrder><ImageOffset>4</ImageOffset><PixelOffset>8</PixelOffset><LineOffset>64</LineOffset></VRTRasterBand>
</VRTDataset>''')
    return inc, hd


@pytest.mark.gpu
def test_conventional_los_from_a_raster_file(tmp_path):
    """losreader.py:110-133 with an ISCE-style 2-band LOS raster: delays / cos(incidence) - the reference reads the file through
    rasterio; here through the built-in reader, and the division runs on the device (rdr_project_cosinc; round 3 divided on the
    host).  A 2-D delay array is divided by the up component; an (.., 3)-shaped one by the ENU vector itself (the reference's
    shape rule).  Tolerance 2 ulp: the device's cos against libm's."""
    from raider_amd.losreader import Conventional, inc_hd_to_enu
    rng = np.random.default_rng(2)
    inc, hd = _los_raster(tmp_path, rng)
    los = Conventional(filename=str(tmp_path / 'los.rdr'))
    lats = rng.uniform(30, 31, (6, 8)); lons = rng.uniform(-118, -117, (6, 8))
    los.setPoints(lats, lons, np.zeros((6, 8)))
    ztd = rng.uniform(2.0, 2.5, (6, 8))
    keep = ztd.copy()
    out = los(ztd)
    assert np.array_equal(ztd, keep) and out.shape == ztd.shape            # (the caller's array is not divided in place)
    np.testing.assert_allclose(out, ztd / inc_hd_to_enu(inc, hd)[..., -1], rtol=5e-16, atol=0)
    enu = inc_hd_to_enu(inc, hd)
    d3 = rng.uniform(2.0, 2.5, (6, 8, 3))
    assert np.array_equal(los(d3), d3 / enu)                               # losreader.py:130-131: same shape -> delays / LOS_enu


def test_conventional_los_from_an_unreadable_file(tmp_path):
    from raider_amd.losreader import Conventional
    rng = np.random.default_rng(2)
    lats = rng.uniform(30, 31, (6, 8)); lons = rng.uniform(-118, -117, (6, 8))
    ztd = rng.uniform(2.0, 2.5, (6, 8))
    # a file that is neither a raster nor an orbit file: one error naming both attempts
    bad = tmp_path / 'junk.bin'
    bad.write_bytes(b'\x01\x02\x03')
    los2 = Conventional(filename=str(bad)); los2.setPoints(lats, lons, np.zeros((6, 8)))
    with pytest.raises(ValueError, match='line-of-sight raster'):
        los2(ztd)


def test_write_delays_station_csv_and_envi_rasters(tmp_path):
    """utilFcns.writeDelays (utilFcns.py:431-464), the output side of the point branch in cli/raider.py: the station CSV gains the three
    delay columns (duplicates dropped, NaN -> no-data value IN PLACE); raster AOIs get two ENVI rasters this package reads back."""
    import pandas as pd
    from raider_amd.utilFcns import rio_open, writeArrayToRaster, writeDelays
    csv = tmp_path / 'stations.csv'
    csv.write_text('ID,Lat,Lon,Hgt_m\nA,33.1,-117.2,10\nB,33.5,-117.9,200\nB2,33.5,-117.9,200\nC,34.0,-118.4,1500\n')
    aoi = type('A', (), dict(type=lambda self: 'station_file', _filename=str(csv)))()
    wet = np.array([0.10, np.nan, 0.05]); hyd = np.array([2.3, 2.2, 1.9])
    writeDelays(aoi, wet, hyd, tmp_path / 'out.csv')
    df = pd.read_csv(tmp_path / 'out.csv')
    assert list(df['ID']) == ['A', 'B', 'C'] and list(df['wetDelay']) == [0.10, 0.0, 0.05] and np.allclose(df['totalDelay'], [2.4, 2.2, 1.95]) and wet[1] == 0.0
    rng = np.random.default_rng(0)
    w2 = rng.uniform(0, 0.3, (7, 9)); h2 = rng.uniform(1.8, 2.4, (7, 9)); w2[2, 3] = np.nan
    raoi = type('R', (), dict(type=lambda self: 'radar_rasters', projection=lambda self: None, geotransform=lambda self: (-118.0, 0.01, 0.0, 34.0, 0.0, -0.01)))()
    with pytest.raises(ValueError, match='Hydro delay file path'):
        writeDelays(raoi, w2.copy(), h2.copy(), tmp_path / 'wet.envi')
    writeDelays(raoi, w2, h2, tmp_path / 'wet.envi', tmp_path / 'hydro.envi', outformat='ENVI', ndv=-9999.0)
    back, prof = rio_open(tmp_path / 'wet.envi')
    assert back.dtype == np.float32 and back.shape == (7, 9) and back[2, 3] == -9999.0 and prof['nodata'] == -9999.0
    assert np.array_equal(back, w2.astype(np.float32)) and np.array_equal(rio_open(tmp_path / 'hydro.envi')[0], h2.astype(np.float32))
    assert 'map info = {Geographic Lat/Lon, 1, 1, -118, 34, 0.01, 0.01, WGS-84}' in (tmp_path / 'wet.hdr').read_text()
    with pytest.raises(RuntimeError, match='cannot write an array of shape'):
        writeArrayToRaster(np.zeros((2, 3, 4)), tmp_path / 'x.envi')
    try:
        import rasterio  # noqa: F401
    except ImportError:
        with pytest.raises(ImportError, match='rasterio'):
            writeArrayToRaster(np.zeros((2, 3)), tmp_path / 'x.tif', fmt='GTiff')
