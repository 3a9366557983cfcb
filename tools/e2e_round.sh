# gpurun -- 'bash tools/e2e_round.sh'  ->  gpurun_out/e2e/*.json  (copy into profiles/r03_e2e.json with tools/e2e_digest.py)
R=$GRAFT_REPO_ROOT
O=$R/gpurun_out/e2e; rm -rf $O; mkdir -p $O
python $R/tools/e2e_tropo_delay.py 2000 2000 8 2>/dev/null | tail -1 > $O/tropo_2000x2000x8.json
python $R/tools/e2e_tropo_delay.py 4000 4000 4 2>/dev/null | tail -1 > $O/tropo_4000x4000x4.json
python $R/tools/e2e_tropo_delay.py 316 316 20 2>/dev/null | tail -1 > $O/tropo_316x316x20.json
python $R/tools/e2e_orbit.py 2>/dev/null | tail -1 > $O/orbit_1000x1000x8.json
python $R/tools/e2e_zenith.py 1000 1000 40 2>/dev/null | tail -1 > $O/zenith_1000x1000x40.json
python $R/tools/bench_slices.py 2>/dev/null | tail -2 > $O/bench_slices.txt
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -w $R/tools/probes/write_probe.hip -o /tmp/wp && /tmp/wp > $O/write_probe.txt
python $R/tools/probes/pin_probe.py > $O/pin_probe.txt 2>/dev/null
python $R/bench.py > $O/bench.json 2> $O/bench.err
python $R/bench.py --per-pixel-ht > $O/bench_per_pixel.json 2>> $O/bench.err
head -c 600 $O/*.json; cat $O/bench_slices.txt
