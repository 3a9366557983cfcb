# gpurun -- 'bash tools/profile_secondary.sh'   then (here)   python tools/secondary_digest.py
# NOTE: delete the local gpurun_out/secondary first - gpurun MERGES new files into it and the digest would mix runs
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/secondary; rm -rf $O; mkdir -p $O
timeout 900 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/tools/secondary_bench.py > $O/bench.json 2> $O/bench.err
tail -c 1500 $O/bench.json; tail -3 $O/bench.err
# HBM bytes per launch (separate PMC passes, as tools/profile_round.sh does for the ray kernels)
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/tools/secondary_bench.py > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/tools/secondary_bench.py > $O/write.log 2>&1
# issue side: VALU instructions, VALU-busy cycles, wave cycles, vector-memory reads per launch (two passes: four counters each)
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python $R/tools/secondary_bench.py > $O/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR SQ_INSTS_SALU SQ_WAVES --output-format csv -d $O/sq2 -- python $R/tools/secondary_bench.py > $O/sq2.log 2>&1
