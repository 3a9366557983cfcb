"""raider_amd.crs: the model CRS of a weather-model / delay-cube file without pyproj - CF grid-mapping attributes (what
CRS.to_cf() writes, models/weatherModel.py:711-715) and WKT (delay.py:66-73 reads `crs_wkt`)."""
from pathlib import Path

import numpy as np
import pytest

from raider_amd import crs, h5lite

REF = Path(__file__).parent / 'golden' / 'ref_files'
DEG = 'ANGLEUNIT["degree",0.0174532925199433]'


def test_files_written_by_the_reference():
    """`proj` of a processed ERA-5 cube and `crs` of a delay cube, both written by the real RAiDER through pyproj's CRS.to_cf():
    WKT2:2019 with a datum ENSEMBLE, plus the CF attributes - EPSG:4326 by either route."""
    for fn, var in (('ERA-5_2020_01_30_T13_52_45_32N_35N_120W_115W.nc', 'proj'), ('HRRR_tropo_20200101T120000_ztd.nc', 'crs')):
        at = h5lite.File(REF / fn)[var].attrs
        assert crs.crs_from_cf(at) == 4326 and crs.crs_from_wkt(at['crs_wkt']) == 4326 and crs.crs_from_proj_var(at) == 4326
        root = crs.parse_wkt(at['crs_wkt'])
        assert root[0] == 'GEOGCRS' and root[1][0] == 'WGS 84' and crs._find(root, 'ELLIPSOID')[1][:3] == ['WGS 84', 6378137.0, 298.257223563]


def test_hrrr_and_hrrr_ak_roundtrip_through_wkt_and_cf():
    hrrr = dict(proj='lcc', lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, x_0=0.0, y_0=0.0, a=6371229.0, b=6371229.0)      # models/hrrr.py:248-259
    ak = dict(proj='stere', lat_0=90.0, lat_ts=60.0, lon_0=225.0, x_0=0.0, y_0=0.0, a=6371229.0, b=6371229.0)                 # models/hrrr.py:22-25
    pole = dict(proj='stere', lat_0=-90.0, k_0=0.994, lon_0=0.0, x_0=2.0e6, y_0=2.0e6, a=6378137.0, rf=298.257223563)
    two = dict(proj='lcc', lat_1=33.0, lat_2=45.0, lat_0=23.0, lon_0=-96.0, x_0=1.5e6, y_0=-2.5e5, a=6378206.4, rf=294.978698213898)
    for d in (hrrr, ak, pole, two):
        wkt, cf = crs.cf_from_crs(d)
        back = crs.crs_from_wkt(wkt)
        assert set(back) == set(d) and all(back[k] == d[k] if isinstance(d[k], str) else abs(back[k] - d[k]) < 1e-9 * max(1.0, abs(d[k])) for k in d), (d, back)
        back = crs.crs_from_cf(cf)
        assert set(back) == set(d) and all(back[k] == d[k] if isinstance(d[k], str) else abs(back[k] - d[k]) < 1e-9 * max(1.0, abs(d[k])) for k in d), (d, back)
        assert crs.crs_from_proj_var(cf) == crs.crs_from_cf(cf)                                  # CF attributes win over the WKT text
        assert crs.crs_from_proj_var(dict(crs_wkt=wkt)) == crs.crs_from_wkt(wkt)                 # a file with the WKT only
    assert cf['grid_mapping_name'] == 'lambert_conformal_conic' and list(cf['standard_parallel']) == [33.0, 45.0]


def test_older_and_foreign_wkt_spellings():
    w1 = ('PROJCS["unnamed",GEOGCS["unnamed ellipse",DATUM["unknown",SPHEROID["unnamed",6371229,0]],PRIMEM["Greenwich",0],UNIT["degree",0.0174532925199433]],'
          'PROJECTION["Lambert_Conformal_Conic_2SP"],PARAMETER["standard_parallel_1",38.5],PARAMETER["standard_parallel_2",38.5],PARAMETER["latitude_of_origin",38.5],'
          'PARAMETER["central_meridian",262.5],PARAMETER["false_easting",0],PARAMETER["false_northing",0],UNIT["Meter",1]]')
    assert crs.crs_from_wkt(w1) == dict(proj='lcc', lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, x_0=0.0, y_0=0.0, a=6371229.0, b=6371229.0)
    ft = w1.replace('PARAMETER["false_easting",0]', 'PARAMETER["false_easting",2000000]').replace('UNIT["Meter",1]', 'UNIT["US survey foot",0.3048006096012192]')
    assert abs(crs.crs_from_wkt(ft)['x_0'] - 2000000 * 0.3048006096012192) < 1e-6                 # WKT1: linear parameters in the CRS's unit
    ps1 = ('PROJCS["NSIDC Sea Ice Polar Stereographic North",GEOGCS["Hughes 1980",DATUM["Hughes_1980",SPHEROID["Hughes 1980",6378273,298.279411123064]],PRIMEM["Greenwich",0],'
           'UNIT["degree",0.0174532925199433]],PROJECTION["Polar_Stereographic"],PARAMETER["latitude_of_origin",70],PARAMETER["central_meridian",-45],'
           'PARAMETER["false_easting",0],PARAMETER["false_northing",0],UNIT["metre",1]]')
    d = crs.crs_from_wkt(ps1)
    assert d['proj'] == 'stere' and d['lat_0'] == 90.0 and d['lat_ts'] == 70.0 and d['lon_0'] == -45.0 and abs(d['rf'] - 298.279411123064) < 1e-9
    va = ('PROJCRS["x",BASEGEOGCRS["WGS 84",DATUM["World Geodetic System 1984",ELLIPSOID["WGS 84",6378137,298.257223563,LENGTHUNIT["metre",1]]]],'
          f'CONVERSION["UPS North",METHOD["Polar Stereographic (variant A)",ID["EPSG",9810]],PARAMETER["Latitude of natural origin",90,{DEG}],'
          f'PARAMETER["Longitude of natural origin",0,{DEG}],PARAMETER["Scale factor at natural origin",0.994,SCALEUNIT["unity",1]],'
          'PARAMETER["False easting",2000000,LENGTHUNIT["metre",1]],PARAMETER["False northing",2000000,LENGTHUNIT["metre",1]]],CS[Cartesian,2]]')
    d = crs.crs_from_wkt(va)
    assert d['proj'] == 'stere' and d['lat_0'] == 90.0 and d['k_0'] == 0.994 and d['x_0'] == 2.0e6 and 'lat_ts' not in d
    sp1 = ('PROJCRS["x",BASEGEOGCRS["y",DATUM["z",ELLIPSOID["GRS 1980",6378137,298.257222101,LENGTHUNIT["metre",1]]]],CONVERSION["c",METHOD["Lambert Conic Conformal (1SP)"],'
           f'PARAMETER["Latitude of natural origin",46.5,{DEG}],PARAMETER["Longitude of natural origin",3,{DEG}],PARAMETER["Scale factor at natural origin",0.99987742,SCALEUNIT["unity",1]],'
           'PARAMETER["False easting",700,LENGTHUNIT["kilometre",1000]],PARAMETER["False northing",6600000,LENGTHUNIT["metre",1]]],CS[Cartesian,2]]')
    d = crs.crs_from_wkt(sp1)
    assert d['proj'] == 'lcc' and d['lat_1'] == d['lat_2'] == d['lat_0'] == 46.5 and d['k_0'] == 0.99987742 and d['x_0'] == 700000.0
    grad = sp1.replace(f'PARAMETER["Latitude of natural origin",46.5,{DEG}]', 'PARAMETER["Latitude of natural origin",51.666666666666667,ANGLEUNIT["grad",0.015707963267949]]')
    assert abs(crs.crs_from_wkt(grad)['lat_0'] - 46.5) < 1e-9                                     # angles in other units come out in degrees
    utm = ('PROJCRS["WGS 84 / UTM zone 11N",BASEGEOGCRS["WGS 84",DATUM["World Geodetic System 1984",ELLIPSOID["WGS 84",6378137,298.257223563,LENGTHUNIT["metre",1]]],'
           f'PRIMEM["Greenwich",0,{DEG}],ID["EPSG",4326]],CONVERSION["UTM zone 11N",METHOD["Transverse Mercator",ID["EPSG",9807]],PARAMETER["Latitude of natural origin",0,{DEG}],'
           f'PARAMETER["Longitude of natural origin",-117,{DEG}],PARAMETER["Scale factor at natural origin",0.9996,SCALEUNIT["unity",1]],PARAMETER["False easting",500000,LENGTHUNIT["metre",1]],'
           'PARAMETER["False northing",0,LENGTHUNIT["metre",1]]],CS[Cartesian,2],AXIS["(E)",east,ORDER[1],LENGTHUNIT["metre",1]],AXIS["(N)",north,ORDER[2],LENGTHUNIT["metre",1]],ID["EPSG",32611]]')
    assert crs.crs_from_wkt(utm) == 32611                                                          # the CRS's own EPSG identifier wins (not the base CRS's 4326)
    d = crs.crs_from_wkt(utm.replace(',ID["EPSG",32611]]', ']'))
    assert d == dict(proj='tmerc', lat_0=0.0, lon_0=-117.0, k_0=0.9996, x_0=500000.0, y_0=0.0, a=6378137.0, rf=298.257223563)
    with pytest.raises(ValueError, match='not built in'):
        crs.crs_from_wkt(utm.replace(',ID["EPSG",32611]]', ']').replace('Transverse Mercator', 'Mercator (variant A)'))
    with pytest.raises(ValueError):
        crs.crs_from_wkt('PROJCRS["broken",BASEGEOGCRS["x"')
    with pytest.raises(KeyError):
        crs.crs_from_proj_var({})


def test_cf_only_grid_mappings():
    d = crs.crs_from_cf(dict(grid_mapping_name='lambert_conformal_conic', standard_parallel=np.array([25.0]), longitude_of_central_meridian=265.0,
                             latitude_of_projection_origin=25.0, earth_radius=6371200.0))
    assert d == dict(proj='lcc', lat_1=25.0, lat_2=25.0, lat_0=25.0, lon_0=265.0, x_0=0.0, y_0=0.0, a=6371200.0, b=6371200.0)
    d = crs.crs_from_cf(dict(grid_mapping_name='polar_stereographic', straight_vertical_longitude_from_pole=-45.0, latitude_of_projection_origin=90.0,
                             standard_parallel=70.0, semi_major_axis=6378137.0, inverse_flattening=298.257223563, false_easting=0.0, false_northing=0.0))
    assert d['proj'] == 'stere' and d['lat_ts'] == 70.0 and d['lon_0'] == -45.0 and d['rf'] == 298.257223563
    d = crs.crs_from_cf(dict(grid_mapping_name='transverse_mercator', scale_factor_at_central_meridian=0.9996, longitude_of_central_meridian=-117.0,
                             latitude_of_projection_origin=0.0, false_easting=500000.0, false_northing=0.0, semi_major_axis=6378137.0, semi_minor_axis=6356752.314245179))
    assert d['proj'] == 'tmerc' and abs(d['rf'] - 298.257223563) < 1e-6
    assert crs.crs_from_cf(dict(grid_mapping_name='latitude_longitude')) == 4326
    assert crs.crs_from_cf(dict(grid_mapping_name='rotated_latitude_longitude')) is None and crs.crs_from_cf({}) is None


def test_delay_cube_on_a_conic_output_grid_carries_its_crs(tmp_path):
    """writeResultsToXarray (delay.py:329-401) with out_proj = the model's own conic CRS: the `crs` grid-mapping variable of the
    written NetCDF-4 file describes it (CF attributes + WKT), and reads back to the same parameters."""
    import datetime
    pytest.importorskip('scipy')
    try:
        import xarray  # noqa: F401
        pytest.skip('xarray installed: the Dataset branch writes the CRS through pyproj')
    except ImportError:
        pass
    from raider_amd.delay import writeResultsToXarray
    hr = '+proj=lcc +lat_1=38.5 +lat_2=38.5 +lat_0=38.5 +lon_0=262.5 +x_0=0 +y_0=0 +a=6371229 +b=6371229 +units=m +no_defs'
    ds = writeResultsToXarray(datetime.datetime(2020, 1, 1), np.arange(4.0) * 3000, np.arange(3.0) * 3000, np.array([0.0, 100.0]), hr, np.zeros((2, 3, 4)),
                              np.ones((2, 3, 4)), 'x.nc', 'zenith')
    ds.to_netcdf(tmp_path / 'out.nc')
    f = h5lite.File(tmp_path / 'out.nc')
    at = f['crs'].attrs
    assert at['grid_mapping_name'] == 'lambert_conformal_conic' and f['wet'].attrs['grid_mapping'] == 'crs' and f['x'].attrs['standard_name'] == 'projection_x_coordinate'
    assert crs.crs_from_proj_var(at) == dict(proj='lcc', lat_1=38.5, lat_2=38.5, lat_0=38.5, lon_0=262.5, x_0=0.0, y_0=0.0, a=6371229.0, b=6371229.0)
