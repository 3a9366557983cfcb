#!/usr/bin/env python3
"""Digest one tools/profile_round.sh output directory (gpurun_out/<tag>) into the tracked files bench.py and DESIGN.md cite:

    profiles/r02_<tag>_counters.json     machine-readable: per kernel VALU instr / 64-ray wave, VALU-busy fraction, VGPR, scratch,
                                         HBM read / write bytes per launch (PMC, gfx950-corrected), rocprof average duration;
                                         keyed by the sha256 of the kernel sources that ran (source_hash.txt written on the GPU box)
    profiles/r02_<tag>_kernel_stats.txt  rocprofv3 --kernel-trace --stats summary
    profiles/r02_<tag>_hbm_traffic_pmc.txt, r02_<tag>_sq_counters_per_raywave.txt, r02_<tag>_bench.json

usage: profile_digest.py gpurun_out/<tag> [round-prefix, default r02]
"""
import csv
import glob
import json
import shutil
import subprocess
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
WAVES_PER_SIMD = 4          # amdgpu_waves_per_eu of the two light ray kernels


def short(name):
    for k in ('march_kernel', 'crossings_kernel'):
        if k in name:
            slow = '<HIP_vector_type<float, 2u>, true' in name or '<HIP_vector_type<double, 2u>, true' in name
            return k + ('_generic' if slow else '')
    return None


def sq(dirs, nwaves):
    acc = defaultdict(lambda: defaultdict(float)); launches = defaultdict(lambda: defaultdict(set)); meta = {}
    for d in dirs:
        for f in glob.glob(str(d) + '/**/*counter_collection.csv', recursive=True):
            for r in csv.DictReader(open(f)):
                k = short(r['Kernel_Name'])
                if not k:
                    continue
                acc[k][r['Counter_Name']] += float(r['Counter_Value']); launches[k][r['Counter_Name']].add(r['Dispatch_Id'])
                meta[k] = dict(vgpr=float(r['VGPR_Count']), scratch=float(r['Scratch_Size']), lds=float(r.get('LDS_Block_Size', 0) or 0))
    out = {}
    for k, d in acc.items():
        out[k] = {c: v / max(1, len(launches[k][c])) / nwaves for c, v in d.items()}
        out[k].update(meta[k])
    return out


def pmc(dirpath, counter):
    per = defaultdict(list); cal = None
    for f in glob.glob(str(dirpath) + '/**/*counter_collection.csv', recursive=True):
        for r in csv.DictReader(open(f)):
            if r['Counter_Name'] != counter:
                continue
            v = float(r['Counter_Value'])
            k = short(r['Kernel_Name'])
            if k:
                per[k].append(v)
            elif ('copy' in r['Kernel_Name'].lower() or 'elementwise' in r['Kernel_Name'].lower()) and v > 1e5:
                cal = max(cal or 0, v)
    return per, cal


def main():
    src = Path(sys.argv[1]); rnd = sys.argv[2] if len(sys.argv) > 2 else 'r02'
    tag = src.name
    prof = REPO / 'profiles'
    info = json.loads((src / 'info.json').read_text())
    res = dict(source_hash=info['source_hash'], tag=tag, cube=info['cube'], sq_scene=info['sq_scene'], hbm_scene=info['hbm_scene'],
               device=info.get('device'), waves_per_simd=WAVES_PER_SIMD,
               method='rocprofv3 --pmc in separate passes (tools/profile_round.sh); FETCH_SIZE corrected by the factor measured on a 1 GiB '
                      'calibration copy in the same pass (gfx950: x2), WRITE_SIZE likewise (x1); SQ counters divided by launches and 64-ray waves',
               kernels={})
    nw = info['sq_scene'][0] * info['sq_scene'][1] / 64.0
    s = sq([src / 'sq1', src / 'sq2'], nw)
    lines = [f'# rocprofv3 --pmc SQ passes on bench.py --rows {info["sq_scene"][0]} --cols {info["sq_scene"][1]} --cube {info["cube"]}; per launch and per 64-ray wave',
             f'# VALU busy while a wave is resident = {WAVES_PER_SIMD} waves/SIMD x SQ_ACTIVE_INST_VALU / SQ_WAVE_CYCLES']
    for k, d in sorted(s.items()):
        e = res['kernels'].setdefault(k, {})
        e['valu_per_raywave'] = round(d.get('SQ_INSTS_VALU', 0.0), 1)
        e['salu_per_raywave'] = round(d.get('SQ_INSTS_SALU', 0.0), 1)
        e['vmem_rd_per_raywave'] = round(d.get('SQ_INSTS_VMEM_RD', 0.0), 1)
        e['lds_per_raywave'] = round(d.get('SQ_INSTS_LDS', 0.0), 1)
        if d.get('SQ_WAVE_CYCLES'):
            e['valu_busy_frac'] = round(WAVES_PER_SIMD * d.get('SQ_ACTIVE_INST_VALU', 0.0) / d['SQ_WAVE_CYCLES'], 4)
            e['wait_inst_any_frac'] = round(d.get('SQ_WAIT_INST_ANY', 0.0) / d['SQ_WAVE_CYCLES'], 4)
        # (registers / LDS / scratch are NOT taken from rocprof's metadata columns - VGPR_Count reads 64 for a 128-register kernel here -
        # bench.py asks the loaded code object: rdr_ray_kernel_attributes)
        lines.append(k + ': ' + json.dumps({c: round(v, 1) for c, v in sorted(d.items())}))
    (prof / f'{rnd}_{tag}_sq_counters_per_raywave.txt').write_text('\n'.join(lines) + '\n')

    fetch, cal_f = pmc(src / 'fetch', 'FETCH_SIZE')
    write, cal_w = pmc(src / 'write', 'WRITE_SIZE')
    kib = float(1 << 20)
    ff = kib / cal_f if cal_f else None
    fw = kib / cal_w if cal_w else None
    res['fetch_correction'] = ff; res['write_correction'] = fw
    lines = ['# HBM bytes per launch: rocprofv3 --pmc FETCH_SIZE and --pmc WRITE_SIZE in separate passes over tools/pmc_probe.py',
             f'# calibration copy of 1 GiB: FETCH_SIZE {cal_f} KiB -> factor {ff}; WRITE_SIZE {cal_w} KiB -> factor {fw}']
    for k in sorted(set(fetch) | set(write)):
        e = res['kernels'].setdefault(k, {})
        f = fetch.get(k, [0.0]); w = write.get(k, [0.0])
        if ff and fw:
            e['hbm_read_bytes'] = sum(f) / len(f) * ff * 1024
            e['hbm_write_bytes'] = sum(w) / len(w) * fw * 1024
            lines.append(f'{k}: launches {len(f)}  read {e["hbm_read_bytes"]/1e9:.3f} GB  write {e["hbm_write_bytes"]/1e9:.3f} GB per launch')
    (prof / f'{rnd}_{tag}_hbm_traffic_pmc.txt').write_text('\n'.join(lines) + '\n')

    stats = sorted(glob.glob(str(src / 'kt') + '/**/*kernel_stats.csv', recursive=True), key=lambda f: Path(f).stat().st_mtime, reverse=True)
    if stats:
        subprocess.run([sys.executable, str(REPO / 'tools' / 'rocprof_summary.py'), stats[0], str(prof / f'{rnd}_{tag}_kernel_stats.txt'), tag], check=True,
                       stdout=subprocess.DEVNULL)
        for r in csv.DictReader(open(stats[0])):
            k = short(r['Name'])
            if k:
                res['kernels'].setdefault(k, {})['rocprof_avg_us'] = float(r['AverageNs']) / 1e3
                res['kernels'][k]['rocprof_calls'] = int(r['Calls'])
    if (src / 'bench.json').exists() and (src / 'bench.json').stat().st_size:
        shutil.copy(src / 'bench.json', prof / f'{rnd}_{tag}_bench.json')
    (prof / f'{rnd}_{tag}_counters.json').write_text(json.dumps(res, indent=1, sort_keys=True) + '\n')
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
