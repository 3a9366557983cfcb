#!/usr/bin/env python3
"""Wall time of tropo_delay on the reference test suite's own processed ERA-5 cube (145 levels, NetCDF-4 read through
raider_amd.h5lite) with the default 0.02-degree output grid of a bounding-box AOI and 4 height levels, ray-traced."""
import datetime as dt
import json
import sys
import time
from pathlib import Path

import numpy as np

ROOT = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(ROOT))
from raider_amd.delay import GridAOI, tropo_delay      # noqa: E402
from raider_amd.losreader import Raytracing             # noqa: E402

cube = ROOT / 'tests' / 'golden' / 'ref_files' / 'ERA-5_2020_01_30_T13_52_45_32N_35N_120W_115W.nc'
x = np.arange(-119.5, -115.5 + 1e-9, 0.02); y = np.arange(34.5, 32.5 - 1e-9, -0.02)
inc = np.broadcast_to(30.0 + 14.0 * np.arange(x.size) / x.size, (y.size, x.size)).copy()
out = {}
for rep in range(4):
    t0 = time.perf_counter()
    ds, _ = tropo_delay(dt.datetime(2020, 1, 30, 13, 52, 45), str(cube), GridAOI(x, y), Raytracing(inc=inc, heading=-167.9),
                        [0.0, 500.0, 1500.0, 3000.0], 4326, None)
    out[f'run{rep}_ms'] = (time.perf_counter() - t0) * 1e3
h = np.asarray(ds['hydro'][:])
out.update(grid=[int(y.size), int(x.size)], rays=int(4 * x.size * y.size), nan_share=float(np.isnan(h).mean()), mean_hydro=float(np.nanmean(h)))
print(json.dumps(out))
if len(sys.argv) > 1 and sys.argv[1] == 'profile':
    import cProfile
    import pstats
    pr = cProfile.Profile(); pr.enable()
    for _ in range(10):
        tropo_delay(dt.datetime(2020, 1, 30, 13, 52, 45), str(cube), GridAOI(x, y), Raytracing(inc=inc, heading=-167.9), [0.0, 500.0, 1500.0, 3000.0], 4326, None)
    pr.disable()
    pstats.Stats(pr).sort_stats('tottime').print_stats(18)
