# gpurun -- 'bash tools/round6_final_lines.sh'  AFTER tools/round6_digest.sh was run here and its profiles/r06_* committed: the bench lines of the final tree with
# counters / mix / parity record filled (the digests are keyed by the source hash) + the rocprofv3 --kernel-trace --stats summary of the driver's default command
cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/final6
python bench.py > gpurun_out/final6/bench.json 2> gpurun_out/final6/bench.err
python bench.py --workload c2 > gpurun_out/final6/bench_c2.json 2> gpurun_out/final6/bench_c2.err
python bench.py --workload c5 > gpurun_out/final6/bench_c5.json 2> gpurun_out/final6/bench_c5.err
python bench.py --per-pixel-ht --no-e2e > gpurun_out/final6/bench_c3b.json 2> gpurun_out/final6/bench_c3b.err
(cd /tmp && TMPDIR=/tmp timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/final6/kt -- python $GRAFT_REPO_ROOT/bench.py > $GRAFT_REPO_ROOT/gpurun_out/final6/bench_under_rocprof.json 2> $GRAFT_REPO_ROOT/gpurun_out/final6/kt.err)
python -c "
import json
d=json.load(open('gpurun_out/final6/bench.json')); r=d['roofline']
print('value', d['value']/1e9, 'ms', d['ms_per_step'], 'march', r['march_ms_per_step'], 'cross', r['crossings_ms_per_step'], 'frac', r['frac'], 'busy', r['valu_busy_frac'], 'clock', r['clock_GHz_measured'])
print({k: r[k] for k in ('executed_fp64_flops_frac','useful_flops_frac','survey_flops_frac_step','survey_bytes_over_hbm_peak_step','frac_class_priced','frac_at_measured_clock','valu_per_evaluated_sample','frac_hbm_measured','traffic_over_compulsory')})
print(d['parity'])
print({k:(v.get('value'),v.get('ms_per_step')) for k,v in d['secondary'].items() if k in ('c2','c5')}, {k:(v['rays_per_s'], v['valu_per_evaluated_sample']) for k,v in d['secondary']['real_levels'].items()})
"
