# SQ counters of one or more builds on one box:  gpurun -- 'bash tools/sq_quick.sh raider_amd/libA.so [raider_amd/libB.so ...]'
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for lib in "$@"; do
  O=$R/gpurun_out/sqq_$(basename $lib .so); rm -rf $O; mkdir -p $O
  RAIDER_HIP_LIB=$R/$lib timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python $R/bench.py --rows 2000 --cols 2000 --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-secondary > $O/sq1.log 2>&1
  RAIDER_HIP_LIB=$R/$lib timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/sq2 -- python $R/bench.py --rows 2000 --cols 2000 --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-secondary > $O/sq2.log 2>&1
  python - $O $lib <<'PY'
import csv, glob, sys
from collections import defaultdict
O, lib = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(set)); dur = defaultdict(list)
for f in glob.glob(O + '/sq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        if 'true,' in k.split('(')[0] and 'false, true' not in k.split('(')[0]: continue
        k = 'march' if 'march_kernel' in k else ('crossings' if 'crossings_kernel' in k else None)
        if not k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']].add(r['Dispatch_Id'])
        if r['Counter_Name'] in ('SQ_INSTS_VALU',): dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
        acc[k]['_vgpr'] = float(r['VGPR_Count']); acc[k]['_scratch'] = float(r['Scratch_Size'])
for k, d in sorted(acc.items()):
    g = lambda c: d[c] / max(1, len(n[k][c])) / 62500.0
    t = sum(dur[k]) / len(dur[k])
    print(f"{lib} {k}: VALU {g('SQ_INSTS_VALU'):.0f} SALU {g('SQ_INSTS_SALU'):.0f} LDS {g('SQ_INSTS_LDS'):.0f} VMEM {g('SQ_INSTS_VMEM_RD'):.0f} busy {4*g('SQ_ACTIVE_INST_VALU')/g('SQ_WAVE_CYCLES'):.3f} "
          f"wait_any {g('SQ_WAIT_INST_ANY')/g('SQ_WAVE_CYCLES'):.3f} clk_GHz {g('SQ_BUSY_CYCLES')*62500/32/(t*1e-3)/1e9:.3f} t_ms(4M rays, counters on) {t:.3f} scratch {d['_scratch']:.0f}")
PY
done
