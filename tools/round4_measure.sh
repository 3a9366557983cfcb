set -x
cd $GRAFT_REPO_ROOT
PROFILE_TAG=v29 bash tools/profile_round.sh > gpurun_out/v29_round.log 2>&1
bash tools/profile_secondary.sh > gpurun_out/secondary_round.log 2>&1
mkdir -p gpurun_out/parity
python tools/full_scene_parity.py 4000 4000 gpurun_out/parity/c3.json > gpurun_out/parity/c3.log 2>&1
python tools/full_scene_parity.py 4000 4000 gpurun_out/parity/c3b.json c3b > gpurun_out/parity/c3b.log 2>&1
python tools/full_scene_parity.py 4000 4000 gpurun_out/parity/c5.json c5 > gpurun_out/parity/c5.log 2>&1
python tools/full_scene_parity.py 10000 10000 gpurun_out/parity/c4.json > gpurun_out/parity/c4.log 2>&1
python tools/full_batch_parity_points.py gpurun_out/parity/points.json > gpurun_out/parity/points.log 2>&1
for f in gpurun_out/parity/*.log; do tail -n 2 $f; done
