#!/usr/bin/env python3
"""Fixtures cut out of the DATA FILES the reference's own tests hold (test infrastructure; run in the build container only).

  tests/golden/ref_files/ERA-5_2019_11_17_T20_51_58_5S_2S_41W_37W.nc
      a processed weather-model cube written by the real RAiDER (NetCDF-4/HDF5; /root/reference/test/weather_files/),
      copied verbatim: exercises the built-in HDF5 reader and pins the refractivity and ZTD stages on real output;
  tests/golden/ref_files/ERA-5_2020_01_30_T13_52_45_32N_35N_120W_115W.nc, scenario_6_stations.csv
      the inputs of the reference's test/test_intersect.py::test_gnss_intersect (golden total zenith delay at station TORP);
  tests/golden/g12_gmao_time_interp.npz
      a 145 x 5 x 6 block of the three GMAO cubes of /root/reference/test/gunw_test_data/weather_files/ (12:00, 15:00 and
      the reference's own `timeInterp` product for 13:52:44): pins the two-epoch temporal blend (cli/raider.py:817-819,
      877-888) on a file the reference itself produced.
  tests/golden/ref_files/HRRR_tropo_20200101T120000_ztd.nc (scenario_1/golden_data: a delay cube written by the reference)
  tests/golden/ref_files/scenario_4/{lat,lon}.rdr (+ .vrt, .hdr): radar-geometry rasters of the reference's test/scenario_4
  tests/golden/ref_files/ERA-5_2019_11_17_T20_51_58.nc, ERA-5_2022_08_29_T17_00_01.nc (+ the latter's processed cube
      ERA-5_2022_08_29_T17_00_01_69N_73N_159W_152W.nc)
      RAW ERA-5 model-level files (NetCDF-3, packed int16 z / t / q / lnsp on 137 levels) whose processed counterparts sit
      beside them: the whole producer chain raw file -> processed cube is replayed against what the real RAiDER wrote.
  raider_amd/data/ecmwf_l137.npz, raider_amd/data/hrrr_l50.npz
      ECMWF's published L137 hybrid-level coefficients a, b (138 each) and the 145 output heights every ECMWF model is
      resampled to, as DATA (read off the reference's models/model_levels.py tables A_137_HRES, B_137_HRES,
      LEVELS_137_HEIGHTS by importing it).
"""
import shutil
import sys
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(REPO))
from raider_amd import h5lite  # noqa: E402

REF_TEST = Path('/root/reference/test')
OUT = REPO / 'tests' / 'golden'


def main():
    (OUT / 'ref_files').mkdir(parents=True, exist_ok=True)
    src = REF_TEST / 'weather_files' / 'ERA-5_2019_11_17_T20_51_58_5S_2S_41W_37W.nc'
    shutil.copyfile(src, OUT / 'ref_files' / src.name)
    # test/test_intersect.py::test_gnss_intersect: the processed cube it runs on and its station list (golden ZTD 2.34514 m at TORP)
    src2 = REF_TEST / 'weather_files' / 'ERA-5_2020_01_30_T13_52_45_32N_35N_120W_115W.nc'
    shutil.copyfile(src2, OUT / 'ref_files' / src2.name)
    shutil.copyfile(REF_TEST / 'scenario_6' / 'stations.csv', OUT / 'ref_files' / 'scenario_6_stations.csv')
    # radar-geometry rasters (flat binary + GDAL .vrt with the statistics GDAL recorded + ENVI .hdr): pins raider_amd.rawraster
    (OUT / 'ref_files' / 'scenario_4').mkdir(parents=True, exist_ok=True)
    for fn in ('lat.rdr', 'lat.rdr.vrt', 'lat.hdr', 'lon.rdr', 'lon.rdr.vrt', 'lon.hdr'):
        shutil.copyfile(REF_TEST / 'scenario_4' / fn, OUT / 'ref_files' / 'scenario_4' / fn)
    # the delay-cube product the reference's tests load as `wmdata` (test/test_delayFcns.py:30-45)
    shutil.copyfile(REF_TEST / 'scenario_1' / 'golden_data' / 'HRRR_tropo_20200101T120000_ztd.nc', OUT / 'ref_files' / 'HRRR_tropo_20200101T120000_ztd.nc')
    d = REF_TEST / 'gunw_test_data' / 'weather_files'
    names = dict(t12='GMAO_2020_01_30_T12_00_00_32N_36N_121W_114W.nc', t15='GMAO_2020_01_30_T15_00_00_32N_36N_121W_114W.nc',
                 interp='GMAO_2020_01_30T13_52_44_timeInterp_32N_36N_121W_114W.nc')
    out = {}
    blk = (slice(None), slice(5, 10), slice(6, 12))
    for tag, fn in names.items():
        f = h5lite.File(d / fn)
        for v in ('wet', 'hydro', 'wet_total', 'hydro_total'):
            out[f'{tag}_{v}'] = f[v].read()[blk]
        out[f'{tag}_datetime'] = np.array(f.attrs['datetime'])
    f = h5lite.File(d / names['t12'])
    out['x'], out['y'], out['z'] = f['x'].read()[blk[2]], f['y'].read()[blk[1]], f['z'].read()
    out['query_time'] = np.array('2020-01-30T13:52:44')
    np.savez_compressed(OUT / 'g12_gmao_time_interp.npz', **out)
    for fn in ('ERA-5_2019_11_17_T20_51_58.nc', 'ERA-5_2022_08_29_T17_00_01.nc', 'ERA-5_2022_08_29_T17_00_01_69N_73N_159W_152W.nc'):
        shutil.copyfile(REF_TEST / 'weather_files' / fn, OUT / 'ref_files' / fn)
    import importlib.util
    spec = importlib.util.spec_from_file_location('model_levels', '/root/reference/tools/RAiDER/models/model_levels.py')
    ml = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(ml)
    (REPO / 'raider_amd' / 'data').mkdir(exist_ok=True)
    np.savez(REPO / 'raider_amd' / 'data' / 'ecmwf_l137.npz', a=np.array(ml.A_137_HRES, dtype=np.float64), b=np.array(ml.B_137_HRES, dtype=np.float64),
             level_heights=np.array(ml.LEVELS_137_HEIGHTS, dtype=np.float64))
    # HRRR's 50 native levels + the 7 padding levels below the surface (model_levels.py:517-533): the z axis of a processed HRRR model
    np.savez(REPO / 'raider_amd' / 'data' / 'hrrr_l50.npz', level_heights=np.array(ml.LEVELS_50_HEIGHTS, dtype=np.float64))
    print('g12_gmao_time_interp.npz', (OUT / 'g12_gmao_time_interp.npz').stat().st_size // 1024, 'KiB;', src.name, src.stat().st_size // 1024, 'KiB')


if __name__ == '__main__':
    main()
