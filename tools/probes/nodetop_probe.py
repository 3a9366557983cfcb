"""Driver of tools/probes/nodetop_probe.c: error of evaluating node-top samples AT the node (VERDICT r5 task 1), per scene."""
import ctypes as C, json, sys
from pathlib import Path
import numpy as np
ROOT = Path(__file__).resolve().parents[2]
sys.path.insert(0, str(ROOT))
from oracle import raider_oracle as O, oracle_c as OC
from raider_amd.synthetic import synthetic_cube, scene_grid

L = C.CDLL(str(Path(__file__).with_name('nodetop_probe.so')))
p = lambda a: a.ctypes.data_as(C.c_void_p)


def run(cube, xpts, ypts, inc, hd, ht=0.0):
    zref = float(cube['zs'].max() - 1.0)
    xx, yy = np.meshgrid(xpts, ypts)
    los = O.look_vectors_from_inc_hd(np.broadcast_to(inc, yy.shape), np.full(yy.shape, hd), yy, xx, ht)
    maxlen, _ = OC.ray_prepass(cube, xpts, ypts, ht, los, zref)
    nparts = np.ascontiguousarray(OC.nparts_of(maxlen), dtype=np.int32)
    shape, lat, lon, los, lo, hi = OC._slice_inputs(cube, xpts, ypts, ht, los, zref)
    ys, xs, zs = (np.ascontiguousarray(cube[k], dtype=np.float64) for k in ('ys', 'xs', 'zs'))
    wet, hyd = OC._yxz(cube)
    out = np.zeros((lat.size, 4))
    L.probe_nodetop(p(lat), p(lon), p(los), C.c_int64(lat.size), C.c_double(ht), p(lo), p(hi), C.c_int(len(lo)), p(nparts),
                    p(ys), C.c_int(ys.size), p(xs), C.c_int(xs.size), p(zs), C.c_int(zs.size), p(wet), p(hyd), C.c_int(0 if wet.dtype == np.float32 else 1), p(out))
    return dict(rays=int(lat.size), K=len(lo), S=int(nparts.sum()), max_abs_dhydro_m=float(np.nanmax(np.abs(out[:, 0]))), max_abs_dwet_m=float(np.nanmax(np.abs(out[:, 1]))),
                max_abs_delta_m=float(np.nanmax(out[:, 2])), delta_last_min=float(np.nanmin(out[:, 3])), delta_last_max=float(np.nanmax(out[:, 3])))


res = {}
cube = synthetic_cube(300, 300, 80, seed=0)
xpts, ypts, inc_cols, hd = scene_grid(4000, 4000)
sel_c = np.arange(0, 4000, 40); sel_r = np.arange(0, 4000, 400)
res['c3_bench_scene_sampled(inc 30-46)'] = run(cube, xpts[sel_c], ypts[sel_r], inc_cols[sel_c][None, :], hd)
for inc in (20.0, 45.0, 60.0, 70.0, 80.0):
    res[f'bench_cube_inc{inc:.0f}'] = run(cube, xpts[sel_c][::4], ypts[sel_r][::2], inc, hd)
# the reference's real ERA5 137-level heights are not here; a 145-level axis to 80 km like model_levels.py's (quadratic spacing stand-in)
c2 = synthetic_cube(60, 60, 145, seed=1, ztop=80400.0)
for inc in (35.0, 45.0, 60.0):
    res[f'145_levels_80km_inc{inc:.0f}'] = run(c2, xpts[sel_c][::4], ypts[sel_r][::2], inc, hd)
print(json.dumps(res, indent=1))
