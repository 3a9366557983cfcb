#!/usr/bin/env python3
"""gpurun_out/gather_{c2,c5} (tools/profile_gather.sh) -> profiles/<round>_{c2,c5}_counters.json + <round>_{c2,c5}_kernel_stats.txt:
per kernel of the step the rocprofv3 average duration and the HBM bytes per launch (FETCH_SIZE in KiB x 2 on gfx950 - the correction
measured on the 1 GiB calibration copy of tools/profile_round.sh -, WRITE_SIZE x 1), at the source hash that ran.
usage: gather_digest.py [round prefix, default r05]"""
import csv
import glob
import json
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent


def newest(pattern):
    """gpurun MERGES a call's files into the local gpurun_out/: a pass directory may hold the CSVs of earlier runs beside the last one's
    (rocprofv3 names them by pid).  Only the newest file per directory is this run's."""
    import glob as _g
    import os as _o
    best = {}
    for f in _g.glob(pattern, recursive=True):
        d = _o.path.dirname(f)
        if d not in best or _o.path.getmtime(f) > _o.path.getmtime(best[d]):
            best[d] = f
    return list(best.values())

RND = sys.argv[1] if len(sys.argv) > 1 else 'r06'
# kernels one step of the workload launches once each (bench.py c2_measure / run_c5)
STEP = {'c2': ('build_cube_setup_kernel', 'build_cube_kernel', 'pack_cube_kernel', 'pack_cube_xfast_kernel', 'interp_points_kernel'),
        'c5': ('blend_kernel', 'blend_pair_kernel', 'interp_points_kernel', 'interp_points_pair_kernel', 'interp_points_blend_kernel', 'interp_points_quad_kernel')}


def short(name):
    for k in ('build_cube_setup_kernel', 'build_cube_kernel', 'pack_cube_xfast_kernel', 'pack_cube_kernel', 'interp_points_blend_kernel', 'interp_points_quad_kernel', 'interp_points_pair_kernel',
              'interp_points_kernel', 'blend_pair_kernel', 'blend_kernel', 'quad_build_kernel', 'nan_scan_kernel'):
        if k in name:
            return k
    return None


def pmc(d, counter):
    per = defaultdict(list)
    for f in newest(str(d) + '/**/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            k = short(r['Kernel_Name'])
            if k and r['Counter_Name'] == counter:
                per[k].append(float(r['Counter_Value']))
    return per


for wl in ('c2', 'c5'):
    src = REPO / 'gpurun_out' / f'gather_{wl}'
    if not (src / 'info.json').exists():
        continue
    info = json.loads((src / 'info.json').read_text())
    res = dict(info, kernels={}, method='rocprofv3 --kernel-trace --stats / --pmc FETCH_SIZE / --pmc WRITE_SIZE, separate passes over `bench.py --workload '
                                        f'{wl}` (tools/profile_gather.sh); FETCH_SIZE KiB x 2 (gfx950), WRITE_SIZE KiB x 1; median launch')
    stats = sorted(glob.glob(str(src / 'kt') + '/**/*kernel_stats.csv', recursive=True), key=lambda f: Path(f).stat().st_mtime, reverse=True)
    if stats:
        subprocess.run([sys.executable, str(REPO / 'tools' / 'rocprof_summary.py'), stats[0], str(REPO / 'profiles' / f'{RND}_{wl}_kernel_stats.txt'),
                        f'bench.py --workload {wl}'], check=True, stdout=subprocess.DEVNULL)
        for r in csv.DictReader(open(stats[0])):
            k = short(r['Name'])
            if k:
                e = res['kernels'].setdefault(k, {})
                if float(r['TotalDurationNs']) > e.get('_tot', 0):
                    e.update(rocprof_avg_us=float(r['AverageNs']) / 1e3, rocprof_calls=int(r['Calls']), rocprof_kernel=r['Name'][:110], _tot=float(r['TotalDurationNs']))
    fetch, write = pmc(src / 'fetch', 'FETCH_SIZE'), pmc(src / 'write', 'WRITE_SIZE')
    for k in sorted(set(fetch) | set(write)):
        e = res['kernels'].setdefault(k, {})
        f, w = sorted(fetch.get(k, [0.0])), sorted(write.get(k, [0.0]))
        e['hbm_read_bytes'] = f[len(f) // 2] * 2.0 * 1024
        e['hbm_write_bytes'] = w[len(w) // 2] * 1024
        e['per_step'] = k in STEP[wl]
    res['kernels'] = {k: e for k, e in res['kernels'].items() if 'rocprof_avg_us' in e}      # (gpurun MERGES into gpurun_out: counters of kernels an earlier build launched are not this run's)
    for e in res['kernels'].values():
        e.pop('_tot', None)
    res['step_traffic_bytes'] = sum(e.get('hbm_read_bytes', 0) + e.get('hbm_write_bytes', 0) for e in res['kernels'].values() if e.get('per_step'))
    if (src / 'bench.json').exists():
        line = [ln for ln in (src / 'bench.json').read_text().splitlines() if ln.startswith('{')]
        if line:
            res['bench_line_under_rocprof'] = json.loads(line[-1])
    (REPO / 'profiles' / f'{RND}_{wl}_counters.json').write_text(json.dumps(res, indent=1, sort_keys=True) + '\n')
    print(wl, 'step traffic', res['step_traffic_bytes'] / 1e6, 'MB;', {k: (round(e.get('rocprof_avg_us', 0), 1), round((e.get('hbm_read_bytes', 0) + e.get('hbm_write_bytes', 0)) / 1e6, 1)) for k, e in res['kernels'].items()})
