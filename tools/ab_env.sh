# A/B of one library under an environment switch (e.g. RAIDER_HIP_F32_TILES) on ONE box:  gpurun -- 'bash tools/ab_env.sh RAIDER_HIP_F32_TILES [bench args]'
V=$1; shift
mkdir -p gpurun_out
for v in 0 1; do
  env $V=$v python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-e2e --no-secondary "$@" --dump gpurun_out/env$v >/dev/null 2>gpurun_out/env$v.err || tail -5 gpurun_out/env$v.err
done
python - <<'PY'
import numpy as np
a = np.load('gpurun_out/env0.rank0.npz'); b = np.load('gpurun_out/env1.rank0.npz')
for k in ('wet', 'hydro'):
    print(f'identity {k}: bit-identical={np.array_equal(a[k], b[k], equal_nan=True)} max|d|={np.nanmax(np.abs(a[k] - b[k])):.3e} nan={int(np.isnan(a[k]).sum())}/{int(np.isnan(b[k]).sum())}')
print('nparts equal:', np.array_equal(a['nparts'], b['nparts']))
PY
rm -f gpurun_out/env0.rank0.npz gpurun_out/env1.rank0.npz
for rep in 1 2 3; do
  for v in 0 1; do
    env $V=$v python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-e2e --no-secondary "$@" 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$V=$v', round(d['value']/1e9,4), 'G rays/s  step', round(d['ms_per_step'],3), 'march', round(d['roofline']['march_ms_per_step'],3), 'crossings', round(d['roofline']['crossings_ms_per_step'],3))"
  done
done
