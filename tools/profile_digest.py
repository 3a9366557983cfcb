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
        for f in newest(str(d) + '/**/*counter_collection.csv'):
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


def instr_mix(d):
    """VALU instructions per 64-ray wave by class, from the SQ class counters of gfx950 (rocprofv3 -L), and what they cost.
    A wave64 instruction issues over 2 cycles on CDNA4's SIMD-32 (MI355X_MICROARCH.md "Wave scheduling"); fp64 arithmetic runs at half
    that rate (78.6 TFLOP/s = 256 CUs x 4 SIMDs x 16 lanes x 2 flop x 2.4 GHz), i.e. 4 cycles.  Conversions are priced at the fp64 rate
    (this kernel's are all f32 <-> f64 or f64 <-> i32: v_cvt_f64_f32 of the gathered values, v_cvt_i32_f64 / v_cvt_f64_i32 of the
    cell search) - an assumption the measured busy cycles check: 4 x SQ_ACTIVE_INST_VALU (quad-cycles) per wave against the model."""
    g = lambda k: float(d.get(k, 0.0))
    fma, add, mul, trans = g('SQ_INSTS_VALU_FMA_F64'), g('SQ_INSTS_VALU_ADD_F64'), g('SQ_INSTS_VALU_MUL_F64'), g('SQ_INSTS_VALU_TRANS_F64')
    f32 = g('SQ_INSTS_VALU_FMA_F32') + g('SQ_INSTS_VALU_ADD_F32') + g('SQ_INSTS_VALU_MUL_F32') + g('SQ_INSTS_VALU_TRANS_F32')
    cvt, i32, i64, valu = g('SQ_INSTS_VALU_CVT'), g('SQ_INSTS_VALU_INT32'), g('SQ_INSTS_VALU_INT64'), g('SQ_INSTS_VALU')
    fp64 = fma + add + mul + trans
    other = valu - (fp64 + f32 + cvt + i32 + i64)
    m = dict(valu=round(valu, 1), fp64=round(fp64, 1), fp64_fma=round(fma, 1), fp64_add=round(add, 1), fp64_mul=round(mul, 1), fp64_trans=round(trans, 1),
             cvt=round(cvt, 1), int32=round(i32, 1), int64=round(i64, 1), f32=round(f32, 1), other=round(other, 1),
             fp64_flops_per_lane=round(2 * fma + add + mul + trans, 1),
             flops_fp64_counter_per_wave=round(g('SQ_INSTS_VALU_FLOPS_FP64'), 1),
             issue_cycles_model=round(4 * (fp64 + cvt) + 2 * (f32 + i32 + i64 + max(other, 0.0)), 1),
             valu_busy_cycles_measured=round(4 * g('SQ_ACTIVE_INST_VALU'), 1),
             vmem_wr=round(g('SQ_INSTS_VMEM_WR'), 2), smem=round(g('SQ_INSTS_SMEM'), 1), branch=round(g('SQ_INSTS_BRANCH'), 1))
    return m


def pmc(dirpath, counter):
    per = defaultdict(list); cal = None
    for f in newest(str(dirpath) + '/**/*counter_collection.csv'):
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
    s = sq(sorted(p_ for p_ in src.glob('sq*') if p_.is_dir()), nw)
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
        if 'SQ_INSTS_VALU_FMA_F64' in d:
            e['instr_mix'] = instr_mix(d)
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
    mix = {k: v['instr_mix'] for k, v in res['kernels'].items() if 'instr_mix' in v}
    if mix:
        (prof / f'{rnd}_instr_mix.json').write_text(json.dumps(dict(
            source_hash=res['source_hash'], tag=tag, cube=res['cube'], scene=res['sq_scene'], unit='VALU instructions per 64-ray wave and launch (rocprofv3 --pmc SQ_INSTS_VALU_* '
            'class counters, tools/profile_round.sh passes sq3-sq6; per-lane flops = 2 x FMA + ADD + MUL + TRANS)',
            pricing='wave64 instruction = 2 cycles on the SIMD-32 (f32 / integer), 4 cycles for fp64 arithmetic and f32<->f64 conversions; valu_busy_cycles_measured = 4 x '
                    'SQ_ACTIVE_INST_VALU (quad-cycles) per wave is the check of that model',
            kernels=mix), indent=1, sort_keys=True) + '\n')
    print(json.dumps(res, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
