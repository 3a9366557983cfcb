# One gpurun call = one full profile set of the current kernels:  gpurun -- 'PROFILE_TAG=v13 bash tools/profile_round.sh'
# then (here)  python tools/profile_digest.py gpurun_out/v13   ->  profiles/r02_v13_*   (bench.py reads r02_v13_counters.json)
set -x
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/${PROFILE_TAG:-v13}; mkdir -p $O
CUBE=${PROFILE_CUBE:-300x300x80}
python - > $O/info.json <<PY
import json, sys
sys.path.insert(0, '$R')
import bench, torch
print(json.dumps(dict(source_hash=bench.kernel_source_hash(), cube='$CUBE', sq_scene=[2000, 2000], hbm_scene=[4000, 4000],
                      device=torch.cuda.get_device_name(0))))
PY
python $R/bench.py --cube $CUBE > $O/bench.json 2> $O/bench.err
timeout 600 rocprofv3 --kernel-trace --stats --output-format csv -d $O/kt -- python $R/bench.py --cube $CUBE --steps 10 --warmup 3 --cpu-sample 0 --no-e2e --no-secondary > $O/kt.log 2>&1
timeout 600 rocprofv3 --pmc FETCH_SIZE --output-format csv -d $O/fetch -- python $R/tools/pmc_probe.py $CUBE > $O/fetch.log 2>&1
timeout 600 rocprofv3 --pmc WRITE_SIZE --output-format csv -d $O/write -- python $R/tools/pmc_probe.py $CUBE > $O/write.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES --output-format csv -d $O/sq1 -- python $R/bench.py --cube $CUBE --rows 2000 --cols 2000 --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-secondary > $O/sq1.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_SALU SQ_INSTS_VMEM_RD SQ_INSTS_LDS SQ_WAIT_INST_ANY --output-format csv -d $O/sq2 -- python $R/bench.py --cube $CUBE --rows 2000 --cols 2000 --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-secondary > $O/sq2.log 2>&1
# instruction mix by class (round 5: measured, not read off the ISA): four more SQ passes of four counters each on the same 2000 x 2000 scene
SQB="python $R/bench.py --cube $CUBE --rows 2000 --cols 2000 --steps 2 --warmup 1 --cpu-sample 0 --no-e2e --no-secondary"
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F64 SQ_INSTS_VALU_ADD_F64 SQ_INSTS_VALU_MUL_F64 SQ_INSTS_VALU_TRANS_F64 --output-format csv -d $O/sq3 -- $SQB > $O/sq3.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_CVT SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_INT64 SQ_INSTS_VALU_FLOPS_FP64 --output-format csv -d $O/sq4 -- $SQB > $O/sq4.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32 SQ_INSTS_VALU_MUL_F32 SQ_INSTS_VALU_TRANS_F32 --output-format csv -d $O/sq5 -- $SQB > $O/sq5.log 2>&1
timeout 600 rocprofv3 --pmc SQ_INSTS_VMEM_WR SQ_INSTS_SMEM SQ_INSTS_BRANCH SQ_WAVES --output-format csv -d $O/sq6 -- $SQB > $O/sq6.log 2>&1
# the vector-L1 return path (second roof).  TA_* / GRBM_* sets hung for ~100 s each on this pool in round 3: ONE TCP set, own short timeout, last
[ -n "$PROFILE_TCP" ] && timeout 240 rocprofv3 --pmc TCP_TOTAL_CACHE_ACCESSES_sum TCP_TCC_READ_REQ_sum --output-format csv -d $O/tcp -- $SQB > $O/tcp.log 2>&1
find $O -name "*.csv" | head -40
cat $O/bench.json
