#!/usr/bin/env python3
"""Replay one dumped fuzz trial (FUZZ_DUMP_TRIAL / FUZZ_DUMP_PATH of tools/fuzz_parity.py) on the GPU and compare it with the oracle values in
the dump:  python tools/fuzz_replay.py gpurun_out/fuzz_dump166.npz"""
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R  # noqa: E402

d = np.load(sys.argv[1])
cube = R.Cube(d['ys'], d['xs'], d['zs'], d['cw'], d['ch'], order='zyx')
wet, hyd, nparts, flags = cube.raytrace(R.Rays.grid(d['xpts'], d['ypts'], los=np.ascontiguousarray(d['los'])), float(d['ht']), float(d['zref']), float(d['max_seg']))
for name, g, o in (('wet', wet, d['ow']), ('hydro', hyd, d['oh'])):
    m = np.isnan(g) != np.isnan(o)
    print(name, 'gpu nan', int(np.isnan(g).sum()), 'oracle nan', int(np.isnan(o).sum()), 'mismatching pixels', np.argwhere(m).tolist(), 'max |d|', float(np.nanmax(np.abs(g - o))) if np.isfinite(g - o).any() else None)
print('nparts equal', np.array_equal(nparts, d['nparts']), 'generic rays', cube.ctx.generic_ray_count())
