# here, after `gpurun -- 'bash tools/round6_measure.sh'`: every digest of the round at the current source hash
python tools/profile_digest.py gpurun_out/v40 r06 > /dev/null
python tools/secondary_digest.py r06 | tail -9 | cut -c1-160
python tools/gather_digest.py r06
python tools/real_levels_digest.py r06 | cut -c1-300
python tools/world8_digest.py r06 | head -4
python tools/e2e_digest.py gpurun_out/e2e6 r06 > /dev/null
for t in c3 c3b c4 c5; do cp gpurun_out/parity/$t.json profiles/r06_full_scene_parity_$t.json; done
cp gpurun_out/parity/points.json profiles/r06_full_batch_parity_points.json
python - <<'PY'
import json
from pathlib import Path
from raider_amd import _lib
h = _lib.source_hash()
fp = [l for l in Path('gpurun_out/fuzz_parity.txt').read_text().splitlines() if l.startswith('{')][-1]
fn = [l for l in Path('gpurun_out/fuzz_natives.txt').read_text().splitlines() if l.startswith('{')][-1]
Path('profiles/r06_fuzz.txt').write_text(f"Randomised sweeps of the round-6 binary (source hash {h}) on the MI355X, inside tools/round6_measure.sh:\n\n"
    f"tools/fuzz_parity.py 2000 5 (ray tracer vs NumPy oracle; DEM trials vs the C oracle)\n{fp}\n\n"
    f"tools/fuzz_natives.py 1500 5 (zenith cube, station queries, interpolate 1-5 D, interpolate_along_axis, makePoints - vs the oracle AND the reference's own compiled extensions)\n{fn}\n")
for t in ('c3', 'c3b', 'c4', 'c5'):
    d = json.load(open(f'profiles/r06_full_scene_parity_{t}.json')); print(t, d.get('source_hash'), d.get('max_abs_hydro_m'), d.get('nparts_equal'), d.get('nan_mask_mismatches'))
c = json.load(open('profiles/r06_v40_counters.json')); print('counters', c['source_hash'], 'tree', h)
PY
