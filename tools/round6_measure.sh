# ONE gpurun call = every record the round-6 bench lines and DESIGN.md cite, at the CURRENT source hash:
#   gpurun --timeout 3000 -- 'bash tools/round6_measure.sh'
# then (here):  python tools/profile_digest.py gpurun_out/v40 r06 ; python tools/secondary_digest.py r06 ; python tools/gather_digest.py r06 ;
#               python tools/world8_digest.py r06 ; cp gpurun_out/parity/<x>.json profiles/r06_full_scene_parity_<x>.json ...
set -x
cd $GRAFT_REPO_ROOT
rm -rf gpurun_out/v40 gpurun_out/secondary gpurun_out/parity gpurun_out/world8 gpurun_out/gather_c2 gpurun_out/gather_c5
PROFILE_TAG=v40 bash tools/profile_round.sh > gpurun_out/v40_round.log 2>&1
bash tools/profile_secondary.sh > gpurun_out/secondary_round.log 2>&1
bash tools/profile_gather.sh > gpurun_out/gather_round.log 2>&1
bash tools/world8_dryrun.sh > gpurun_out/world8_script.log 2>&1
# the march kernel on the REAL level axes (secondary.real_levels of the default line): SQ counters per model
rm -rf gpurun_out/real_levels
for M in era5 hrrr; do
  mkdir -p gpurun_out/real_levels/$M
  python tools/real_levels_run.py $M 2000 > gpurun_out/real_levels/$M/info.json 2> gpurun_out/real_levels/$M/info.err
  (cd /tmp && TMPDIR=/tmp timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $GRAFT_REPO_ROOT/gpurun_out/real_levels/$M/sq1 -- python $GRAFT_REPO_ROOT/tools/real_levels_run.py $M 2000 > $GRAFT_REPO_ROOT/gpurun_out/real_levels/$M/sq1.log 2>&1)
done
mkdir -p gpurun_out/parity
python tools/full_scene_parity.py 4000 4000 gpurun_out/parity/c3.json > gpurun_out/parity/c3.log 2>&1
python tools/full_scene_parity.py 4000 4000 gpurun_out/parity/c3b.json c3b > gpurun_out/parity/c3b.log 2>&1
python tools/full_scene_parity.py 4000 4000 gpurun_out/parity/c5.json c5 > gpurun_out/parity/c5.log 2>&1
python tools/full_scene_parity.py 10000 10000 gpurun_out/parity/c4.json > gpurun_out/parity/c4.log 2>&1
python tools/full_batch_parity_points.py gpurun_out/parity/points.json > gpurun_out/parity/points.log 2>&1
python tools/fuzz_parity.py 2000 5 > gpurun_out/fuzz_parity.txt 2>&1
python tools/fuzz_natives.py 1500 5 > gpurun_out/fuzz_natives.txt 2>&1
mkdir -p gpurun_out/e2e6
python tools/e2e_points.py c2 > gpurun_out/e2e6/e2e_c2.json 2> gpurun_out/e2e6/e2e_c2.err
python tools/e2e_points.py c5 > gpurun_out/e2e6/e2e_c5.json 2> gpurun_out/e2e6/e2e_c5.err
python tools/e2e_tropo_delay.py > gpurun_out/e2e6/e2e_tropo.json 2> gpurun_out/e2e6/e2e_tropo.err
python tools/e2e_zenith.py 1000 1000 40 > gpurun_out/e2e6/e2e_zenith.json 2> gpurun_out/e2e6/e2e_zenith.err
python bench.py --workload c5 > gpurun_out/e2e6/bench_c5.json 2> gpurun_out/e2e6/bench_c5.err
python bench.py --workload c2 > gpurun_out/e2e6/bench_c2.json 2> gpurun_out/e2e6/bench_c2.err
python bench.py > gpurun_out/e2e6/bench.json 2> gpurun_out/e2e6/bench.err
python tools/probes/cold_path_breakdown.py > gpurun_out/e2e6/cold_path.json 2>/dev/null
for f in gpurun_out/parity/*.log gpurun_out/fuzz_*.txt; do tail -n 2 $f; done
