# gpurun -- 'bash tools/profile_gather.sh'  then (here)  python tools/gather_digest.py r05
# The two gather workloads of BASELINE.json as bench.py runs them (--workload c2 / c5): kernel durations and HBM bytes per launch
# (rocprofv3 --kernel-trace --stats, --pmc FETCH_SIZE, --pmc WRITE_SIZE in separate passes) -> profiles/r05_c2_counters.json, r05_c5_counters.json,
# which the `roofline.traffic` of those lines cites at the current source hash.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
for W in c2 c5; do
  O=$R/gpurun_out/gather_$W; rm -rf $O; mkdir -p $O
  B="python $R/bench.py --workload $W --steps 4 --warmup 1 --cpu-sample 0 --no-e2e"
  python - > $O/info.json <<PY
import json, sys
sys.path.insert(0, '$R')
import bench
print(json.dumps(dict(source_hash=bench.kernel_source_hash(), workload='$W', steps=4, warmup=1)))
PY
  timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- $B > $O/bench.json 2> $O/kt.err
  timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- $B > $O/fetch.log 2>&1
  timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- $B > $O/write.log 2>&1
  tail -c 400 $O/bench.json
done
