#!/usr/bin/env python3
"""gpurun_out/<dir> with the host-API timings of a round -> profiles/<round>_e2e.json (what DESIGN.md section 4 cites).
Inputs (each optional): e2e_c2.json / e2e_c5.json (tools/e2e_points.py c2 / c5), e2e_tropo.json (tools/e2e_tropo_delay.py), e2e_zenith.json
(tools/e2e_zenith.py), bench_c5.json (bench.py --workload c5), bench.json (bench.py).
usage: e2e_digest.py gpurun_out/<dir> [round-prefix, default r04]"""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))
from raider_amd import _lib       # noqa: E402

src = Path(sys.argv[1]); rnd = sys.argv[2] if len(sys.argv) > 2 else 'r06'


def last_json(name):
    f = src / name
    if not f.exists():
        return None
    lines = [ln for ln in f.read_text().splitlines() if ln.startswith('{')]
    return json.loads(lines[-1]) if lines else None


out = {'source_hash': _lib.source_hash(), 'from': str(src)}
c2 = last_json('e2e_c2.json')
if c2:
    n = c2['points']
    out['points_c2'] = {
        'what': f'tropo_delay(datetime, processed 300x300x80 cube file on disk, {n} query points with their own heights (PointsAOI), LOS) -> wet / hydro at the '
                'points, NumPy in / NumPy out (tools/e2e_points.py c2; best of 5 warm calls: the opened file and its device cube are cached by file identity, '
                'the AOI object serves every call as in cli/raider.py)',
        'points': n, 'intermediate_grid_zyx': c2.get('intermediate_grid'),
        'zenith_ms': c2['zenith_ms'], 'zenith_points_per_s': c2['zenith_points_per_s'], 'zenith_first_call_ms': c2['zenith_first_call_ms'],
        'conventional_ms': c2['conventional_ms'], 'conventional_points_per_s': c2['conventional_points_per_s'],
        'conventional_what': 'Conventional(inc=<per-point incidence raster>, heading): the division by cos(inc) runs in the gather kernel (rdr_point_delays)',
        'conventional_fresh_aoi_ms': c2.get('conventional_fresh_aoi_ms'), 'conventional_fresh_aoi_points_per_s': c2.get('conventional_fresh_aoi_points_per_s'),
        'fresh_aoi_note': 'a NEW PointsAOI per call as well: four min / max passes over the points for their bounding box (host work of the AOI provider)',
        'file_cache_off_ms': c2.get('zenith_nocache_ms'), 'file_cache_off_points_per_s': c2.get('zenith_nocache_points_per_s'),
        'file_cache_off_note': 'RAIDER_HIP_FILE_CACHE=0: every call opens the file and uploads the 115 MB of f64 totals again',
        'pieces_ms': {'build_delay_cube': c2.get('piece_build_delay_cube_ms'), 'interp_project_with_inc': c2.get('piece_interp_project_ms'), 'interp_project': c2.get('piece_interp_ms')},
        'conventional_vs_zenith_over_cos_max_rel': c2.get('conventional_vs_zenith_over_cos_max_rel'),
        'round3_for_comparison': {'zenith_points_per_s': 68.7e6, 'ms': 14.6, 'source': 'VERDICT r3 (gpurun_out/.last_call.json of round 3)'}}
c5 = last_json('e2e_c5.json')
if c5:
    out['points_c5'] = {
        'what': 'tropo_delay(datetime, device-resident blend(0.25, 0.75) of two HRRR-sized 1000x1000x50 epochs on the 3-km LCC grid (ProcessedModel), '
                f'{c5["points"]} stations (PointsAOI with a 0.03 deg lon/lat output grid), Zenith()) -> wet / hydro at the stations (tools/e2e_points.py c5)',
        'points': c5['points'], 'intermediate_grid_zyx': c5.get('intermediate_grid'), 'zenith_ms': c5['zenith_ms'], 'zenith_points_per_s': c5['zenith_points_per_s'],
        'zenith_first_call_ms': c5['zenith_first_call_ms'], 'blend_ms': c5.get('blend_ms'), 'nan': c5.get('zenith_nan')}
tr = last_json('e2e_tropo.json')
if tr:
    out['tropo_2000x2000x8'] = {'what': 'tropo_delay(datetime, processed-cube NetCDF on disk, grid AOI, Raytracing(inc raster, heading), heights) -> NumPy delay cubes '
                                        '(tools/e2e_tropo_delay.py; best warm call)', 'rays': tr['rays'], 'rays_per_s': tr['rays_per_s'],
                                'ms': 1e3 * min(v for k, v in tr.items() if k.startswith('run') and k != 'run0_s'), 'first_call_ms': 1e3 * tr['run0_s']}
ze = last_json('e2e_zenith.json')
if ze:
    out['zenith_1000x1000x40'] = ze
b5 = last_json('bench_c5.json')
if b5:
    out['bench_c5'] = {k: b5[k] for k in ('metric', 'value', 'unit', 'ms_per_step', 'n_gpus') if k in b5}
    out['bench_c5']['roofline'] = {k: b5['roofline'][k] for k in ('achieved', 'peak', 'unit', 'frac', 'blend_ms', 'interp_ms_per_step') if k in b5['roofline']}
    out['bench_c5']['cpu_baseline'] = b5.get('cpu_baseline')
b = last_json('bench.json')
if b:
    out['bench'] = {k: b[k] for k in ('value', 'unit', 'ms_per_step') if k in b}
    out['bench']['end_to_end'] = b.get('end_to_end')
(REPO / 'profiles' / f'{rnd}_e2e.json').write_text(json.dumps(out, indent=1) + '\n')
print(json.dumps(out, indent=1)[:3000])
