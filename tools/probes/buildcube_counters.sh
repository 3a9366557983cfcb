R=$GRAFT_REPO_ROOT; cd /tmp; export TMPDIR=/tmp
O=$R/gpurun_out/bcc; rm -rf $O; mkdir -p $O
i=0
for set in "SQ_INSTS_VALU SQ_INSTS_LDS SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT" "SQ_WAVE_CYCLES SQ_WAIT_INST_ANY SQ_LDS_IDX_ACTIVE SQ_ACTIVE_INST_LDS" "SQ_BUSY_CYCLES SQ_WAVES SQ_INSTS_SALU SQ_INSTS_VMEM_RD"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --output-format csv -d $O/s$i -- python $R/tools/probes/buildcube_run.py > $O/s$i.log 2>&1
done
python - <<PY
import csv, glob
from collections import defaultdict
acc = defaultdict(list)
for f in glob.glob('$O/**/*counter_collection.csv', recursive=True):
    for r in csv.DictReader(open(f)):
        if 'build_cube' in r['Kernel_Name']:
            acc[r['Counter_Name']].append(float(r['Counter_Value']))
for k, v in sorted(acc.items()):
    print(f'{k:40s} {sorted(v)[len(v)//2]:16.0f}   ({len(v)} launches)')
PY
