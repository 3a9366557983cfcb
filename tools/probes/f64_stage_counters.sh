# SQ counters of the f64 marcher with and without the LDS staging:  gpurun -- 'bash tools/probes/f64_stage_counters.sh'
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for st in 0 1; do
  O=$R/gpurun_out/f64stage_$st; rm -rf $O; mkdir -p $O
  RAIDER_HIP_F64_STAGE=$st timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python $R/tools/probes/f64_stage_probe.py 2000 2000 > $O/sq1.log 2>&1
  RAIDER_HIP_F64_STAGE=$st timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/sq2 -- python $R/tools/probes/f64_stage_probe.py 2000 2000 > $O/sq2.log 2>&1
  RAIDER_HIP_F64_STAGE=$st timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INST_CYCLES_VMEM --output-format csv -d $O/sq3 -- python $R/tools/probes/f64_stage_probe.py 2000 2000 > $O/sq3.log 2>&1
  python - $O $st <<'PY'
import csv, glob, sys
from collections import defaultdict
O, st = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(set)); dur = defaultdict(list)
for f in glob.glob(O + '/sq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'march_kernel' not in k or 'double' not in k or 'true,' in k.split('(')[0]: continue
        k = 'march_f64'
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']].add(r['Dispatch_Id'])
        if r['Counter_Name'] in ('SQ_INSTS_VALU',): dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, d in sorted(acc.items()):
    g = lambda c: d[c] / max(1, len(n[k][c])) / 62500.0
    t = sum(dur[k]) / max(1, len(dur[k]))
    print(f"stage={st} {k}: per 64-ray wave: VALU {g('SQ_INSTS_VALU'):.0f} SALU {g('SQ_INSTS_SALU'):.0f} LDS {g('SQ_INSTS_LDS'):.0f} VMEM {g('SQ_INSTS_VMEM_RD'):.0f} busy {4*g('SQ_ACTIVE_INST_VALU')/max(1,g('SQ_WAVE_CYCLES')):.3f} "
          f"wait_any {g('SQ_WAIT_INST_ANY')/max(1,g('SQ_WAVE_CYCLES')):.3f} wait_lds {g('SQ_WAIT_INST_LDS')/max(1,g('SQ_WAVE_CYCLES')):.3f} bank_conflict_cycles {g('SQ_LDS_BANK_CONFLICT'):.0f} lds_idx_active {g('SQ_LDS_IDX_ACTIVE'):.0f} "
          f"vmem_cycles {g('SQ_INST_CYCLES_VMEM'):.0f} wave_cycles {g('SQ_WAVE_CYCLES'):.0f} t_ms(4M rays, counters on) {t:.3f}")
PY
  tail -2 $O/sq3.log
done
