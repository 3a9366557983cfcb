# Are two builds of libraider_hip.so bit-identical on the bench scene (and on the DEM variant)?   gpurun -- 'bash tools/ab_identity.sh raider_amd/libA.so raider_amd/libB.so'
mkdir -p gpurun_out
for v in "" "--per-pixel-ht"; do
  RAIDER_HIP_LIB=$PWD/$1 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-e2e --no-secondary $v --dump gpurun_out/idA >/dev/null 2>&1
  RAIDER_HIP_LIB=$PWD/$2 python bench.py --steps 1 --warmup 0 --cpu-sample 0 --no-e2e --no-secondary $v --dump gpurun_out/idB >/dev/null 2>&1
  python - "$v" <<'PY'
import numpy as np, sys
a = np.load('gpurun_out/idA.rank0.npz'); b = np.load('gpurun_out/idB.rank0.npz')
for k in ('wet', 'hydro'):
    same = np.array_equal(a[k], b[k], equal_nan=True)
    d = np.nanmax(np.abs(a[k] - b[k]))
    print(f'identity[{sys.argv[1] or "c3"}] {k}: bit-identical={same} max|d|={d:.3e} nan={int(np.isnan(a[k]).sum())}/{int(np.isnan(b[k]).sum())}')
print('nparts equal:', np.array_equal(a['nparts'], b['nparts']))
PY
done
rm -f gpurun_out/idA.rank0.npz gpurun_out/idB.rank0.npz
