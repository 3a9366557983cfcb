# A/B of two builds of libraider_hip.so on ONE box (box-to-box spread is +-3 %):  gpurun -- 'bash tools/ab_bench.sh raider_amd/libA.so raider_amd/libB.so'
for rep in 1 2 3; do
  for lib in "$@"; do
    RAIDER_HIP_LIB=$PWD/$lib python bench.py --steps 20 --warmup 5 --cpu-sample 0 --no-e2e --no-secondary 2>/dev/null | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('$lib', round(d['value']/1e9,4), 'G rays/s  step', round(d['ms_per_step'],3), 'march', round(d['roofline']['march_ms_per_step'],3), 'crossings', round(d['roofline']['crossings_ms_per_step'],3))"
  done
done
