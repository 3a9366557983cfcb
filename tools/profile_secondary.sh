# gpurun -- 'bash tools/profile_secondary.sh'   then (here)   python tools/secondary_digest.py
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/secondary; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/secondary_bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; tail -3 $O/bench.err
