# gpurun -- 'bash tools/world8_dryrun.sh'  ->  gpurun_out/world8/*.json (then: python tools/world8_digest.py -> profiles/r05_world8_dryrun.json)
# Eight ranks on ONE GPU (gloo, device-resident collective tensors: the path `--backend auto` picks when ranks > devices), launched exactly as
# the driver launches a scaling run (torch.distributed.run, --nproc-per-node 8, 127.0.0.1): configs[3] at full size (10000 x 10000, strong
# scaling), the c5 and c2 gather workloads, each beside its one-rank run.  Not a scaling measurement (the ranks share one device): a rehearsal
# of every branch an 8-GPU node will take - rendezvous, cube broadcast, MAX all-reduce per step, max-over-ranks timing, one JSON line.
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
O=$R/gpurun_out/world8; rm -rf $O; mkdir -p $O
run8() { tag=$1; shift; python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29517 $R/bench.py --gpus 8 "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; tail -c 600 $O/$tag.json; }
run1() { tag=$1; shift; python $R/bench.py --gpus 1 "$@" > $O/$tag.json 2> $O/$tag.err; echo "$tag rc=$?"; }
run8 rays8 --steps 5 --warmup 2 --cpu-sample 0 --no-e2e
run1 rays1 --steps 5 --warmup 2 --cpu-sample 0 --no-e2e --no-secondary --rows 10000 --cols 10000
run8 c5_8 --workload c5 --steps 5 --warmup 2 --cpu-sample 0
run1 c5_1 --workload c5 --steps 5 --warmup 2 --cpu-sample 0
run8 c2_8 --workload c2 --steps 5 --warmup 2 --cpu-sample 0 --no-e2e
run1 c2_1 --workload c2 --steps 5 --warmup 2 --cpu-sample 0 --no-e2e
# the nccl preflight: one message, exit code != 0, no torchrun stack
python $R/bench.py --gpus 8 --backend nccl --steps 1 --warmup 0 > $O/nccl_preflight.out 2> $O/nccl_preflight.err; echo "preflight rc=$?" | tee $O/nccl_preflight.rc
cat $O/nccl_preflight.err | head -5
