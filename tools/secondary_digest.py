#!/usr/bin/env python3
"""gpurun_out/secondary (tools/profile_secondary.sh) -> profiles/r03_secondary.json + r03_secondary_kernel_stats.txt: per secondary
kernel the rocprofv3 average duration, the SURVEY 8(d) algorithmic bytes and the resulting rate against the 8 TB/s HBM peak."""
import csv
import glob
import json
import subprocess
import sys
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

TAG = sys.argv[1] if len(sys.argv) > 1 else 'r06'        # round prefix of the files written under profiles/
src = REPO / 'gpurun_out' / 'secondary'
line = [ln for ln in (src / 'bench.json').read_text().splitlines() if ln.startswith('{')][-1]
res = json.loads(line)
stats = max(glob.glob(str(src / 'kt') + '/**/*kernel_stats.csv', recursive=True), key=lambda f: Path(f).stat().st_mtime)   # (gpurun merges runs: newest)
subprocess.run([sys.executable, str(REPO / 'tools' / 'rocprof_summary.py'), stats, str(REPO / 'profiles' / f'{TAG}_secondary_kernel_stats.txt'),
                'tools/secondary_bench.py: configs 2 / 5 sizes, producer, orbit look vectors'], check=True, stdout=subprocess.DEVNULL)
rows = list(csv.DictReader(open(stats)))
for k, d in res.items():
    if not isinstance(d, dict) or 'units' not in d:
        continue
    cand = [r for r in rows if k in r['Name']]
    if not cand:
        continue
    r = max(cand, key=lambda r: float(r['TotalDurationNs']))
    avg = float(r['AverageNs']) * 1e-9
    d['rocprof_kernel'] = r['Name'][:120]; d['rocprof_calls'] = int(r['Calls']); d['rocprof_avg_us'] = avg * 1e6
    d['units_per_s'] = d['units'] / avg
    d['algorithmic_GBps'] = d['units'] * d['bytes_per_unit'] / avg / 1e9
    d['frac_of_hbm_peak_8TBps'] = d['algorithmic_GBps'] / 8000.0
# measured HBM bytes per launch: FETCH_SIZE (KiB; gfx950: x2, checked on the calibration copy of tools/profile_round.sh) and WRITE_SIZE (x1)
def pmc(sub, counter):
    out = {}
    for f in newest(str(src / sub) + '/**/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] == counter:
                out.setdefault(r['Kernel_Name'], []).append(float(r['Counter_Value']))
    return out


fetch, write = pmc('fetch', 'FETCH_SIZE'), pmc('write', 'WRITE_SIZE')
for k, d in res.items():
    if not isinstance(d, dict) or 'units' not in d:
        continue
    fr = [v for name, vals in fetch.items() if k in name for v in vals]
    wr = [v for name, vals in write.items() if k in name for v in vals]
    if fr and wr and 'rocprof_avg_us' in d:
        fr, wr = sorted(fr)[len(fr) // 2], sorted(wr)[len(wr) // 2]                 # median launch
        d['hbm_read_bytes'] = fr * 2.0 * 1024; d['hbm_write_bytes'] = wr * 1024
        d['hbm_measured_GBps'] = (d['hbm_read_bytes'] + d['hbm_write_bytes']) / (d['rocprof_avg_us'] * 1e-6) / 1e9
        d['hbm_measured_frac'] = d['hbm_measured_GBps'] / 8000.0
# issue side (tools/profile_secondary.sh, two SQ passes): VALU instructions per launch against the issue peak of 256 CUs x 4 SIMDs x 2.4 GHz / 4
# cycles per wave64 instruction; VALU-busy = SQ_ACTIVE_INST_VALU x 4 / SQ_BUSY_CYCLES-normalised wave cycles is not defined across kernels
# of different occupancy, so the two robust figures are kept: instructions issued / issue slots of the launch, and instructions per unit
VALU_PEAK = 256 * 4 * 2.4e9 / 4


def sq(sub):
    out = {}
    for f in newest(str(src / sub) + '/**/*counter_collection.csv'):
        for r in csv.DictReader(open(f)):
            out.setdefault((r['Kernel_Name'], r['Counter_Name']), []).append(float(r['Counter_Value']))
    return out


sqc = {}
for sub in ('sq1', 'sq2'):
    for (name, cn), vals in sq(sub).items():
        sqc.setdefault(name, {})[cn] = sorted(vals)[len(vals) // 2]
for k, d in res.items():
    if not isinstance(d, dict) or 'units' not in d:
        continue
    hit = [v for name, v in sqc.items() if k in name]
    if not hit or 'rocprof_avg_us' not in d:
        continue
    v = hit[0]
    t = d['rocprof_avg_us'] * 1e-6
    if 'SQ_INSTS_VALU' in v:
        d['valu_instr_per_launch'] = v['SQ_INSTS_VALU']
        d['valu_instr_per_unit'] = v['SQ_INSTS_VALU'] * 64.0 / d['units']          # lane-instructions per unit
        d['valu_issue_frac'] = v['SQ_INSTS_VALU'] / t / VALU_PEAK
    if 'SQ_ACTIVE_INST_VALU' in v and 'SQ_BUSY_CYCLES' in v and v['SQ_BUSY_CYCLES'] > 0:
        d['valu_active_cycles_over_busy_cycles'] = v['SQ_ACTIVE_INST_VALU'] / v['SQ_BUSY_CYCLES']
    for cn, key in (('SQ_INSTS_VMEM_RD', 'vmem_rd_instr_per_launch'), ('SQ_INSTS_VMEM_WR', 'vmem_wr_instr_per_launch'), ('SQ_INSTS_SALU', 'salu_instr_per_launch'),
                    ('SQ_WAVES', 'waves_per_launch'), ('SQ_WAVE_CYCLES', 'wave_cycles_per_launch')):
        if cn in v:
            d[key] = v[cn]
    if 'hbm_read_bytes' in d:
        d['hbm_bytes_per_unit'] = (d['hbm_read_bytes'] + d['hbm_write_bytes']) / d['units']
sys.path.insert(0, str(REPO))
from raider_amd import _lib                                   # noqa: E402
res['source_hash'] = _lib.source_hash()
(REPO / 'profiles' / f'{TAG}_secondary.json').write_text(json.dumps(res, indent=1) + '\n')
for k, d in res.items():
    if not isinstance(d, dict) or 'unit' not in d:
        continue
    print(f"{k:24s} {d.get('rocprof_avg_us', float('nan')):10.1f} us  {d.get('units_per_s', 0)/1e9:8.2f} G {d['unit']}/s  {d.get('algorithmic_GBps', 0):9.1f} GB/s  = {d.get('frac_of_hbm_peak_8TBps', 0):.3f} of 8 TB/s;  measured HBM {d.get('hbm_measured_GBps', float('nan')):8.1f} GB/s = {d.get('hbm_measured_frac', float('nan')):.3f} ({d.get('hbm_bytes_per_unit', float('nan')):.1f} B/{d['unit'][:-1]});  VALU issue {d.get('valu_issue_frac', float('nan')):.3f} ({d.get('valu_instr_per_unit', float('nan')):.0f} lane-instr/{d['unit'][:-1]})")
