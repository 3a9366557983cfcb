# SQ counters of ONE library under an environment switch, on one box:  gpurun -- 'bash tools/sq_env.sh RAIDER_HIP_F32_TILES 0 1'
R=$GRAFT_REPO_ROOT
V=$1; shift
cd /tmp && export TMPDIR=/tmp
for val in "$@"; do
  O=$R/gpurun_out/sqe_${V}_$val; rm -rf $O; mkdir -p $O
  B="python $R/bench.py --rows ${ROWS:-2000} --cols ${ROWS:-2000} --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-secondary"
  env $V=$val timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- $B > $O/sq1.log 2>&1
  env $V=$val timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/sq2 -- $B > $O/sq2.log 2>&1
  env $V=$val timeout 600 rocprofv3 --pmc SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_WAIT_ANY --output-format csv -d $O/sq3 -- $B > $O/sq3.log 2>&1
  env $V=$val timeout 600 rocprofv3 --pmc SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_ANY --output-format csv -d $O/sq4 -- $B > $O/sq4.log 2>&1
  python - $O "$V=$val" <<'PY'
import csv, glob, sys
from collections import defaultdict
O, lib = sys.argv[1], sys.argv[2]
acc = defaultdict(lambda: defaultdict(float)); n = defaultdict(lambda: defaultdict(set)); dur = defaultdict(list)
for f in glob.glob(O + '/sq*/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        k = r['Kernel_Name']
        k = 'march' if ('march_kernel' in k and 'false, 1' in k.replace('(bool)0', 'false').replace('(bool)1', 'true')) else ('march_other' if 'march_kernel' in k else ('crossings' if 'crossings_kernel' in k else None))
        if not k: continue
        acc[k][r['Counter_Name']] += float(r['Counter_Value']); n[k][r['Counter_Name']].add(r['Dispatch_Id'])
        if r['Counter_Name'] in ('SQ_INSTS_VALU',): dur[k].append((int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6)
for k, d in sorted(acc.items()):
    g = lambda c: d[c] / max(1, len(n[k][c])) / (float(__import__("os").environ.get("ROWS", "2000")) ** 2 / 64.0)
    if not dur[k]: continue
    t = sum(dur[k]) / len(dur[k]); wc = g('SQ_WAVE_CYCLES') or 1
    print(f"{lib} {k}: VALU {g('SQ_INSTS_VALU'):.0f} SALU {g('SQ_INSTS_SALU'):.0f} LDS {g('SQ_INSTS_LDS'):.0f} VMEM {g('SQ_INSTS_VMEM_RD'):.0f} | per wave-cycle: valu_busy {4*g('SQ_ACTIVE_INST_VALU')/wc:.3f} "
          f"lds_active {g('SQ_ACTIVE_INST_LDS')/wc:.3f} vmem_active {g('SQ_ACTIVE_INST_VMEM')/wc:.3f} any_active {g('SQ_ACTIVE_INST_ANY')/wc:.3f} wait_inst_any {g('SQ_WAIT_INST_ANY')/wc:.3f} wait_inst_lds {g('SQ_WAIT_INST_LDS')/wc:.3f} wait_any {g('SQ_WAIT_ANY')/wc:.3f} "
          f"| lds_idx_active {g('SQ_LDS_IDX_ACTIVE'):.0f} bank_conflict {g('SQ_LDS_BANK_CONFLICT'):.0f} wave_cycles {wc:.0f} t_ms {t:.3f} launches {len(dur[k])}")
PY
done
