#!/usr/bin/env python3
"""gpurun_out/real_levels/<model>/{sq1,info.json} (tools/round6_measure.sh) -> profiles/<round>_real_levels_counters.json: VALU instructions per
64-ray wave, VALU busy and the sample counts of the march kernel on the REAL level axes (ERA5 145, HRRR 57), at the source hash that ran.
bench.py's secondary.real_levels cites it (valu_per_evaluated_sample).  usage: real_levels_digest.py [round prefix, default r06]"""
import csv
import glob
import json
import os
import re
import sys
from collections import defaultdict
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
RND = sys.argv[1] if len(sys.argv) > 1 else 'r06'
out = {}
for model in ('era5', 'hrrr'):
    src = REPO / 'gpurun_out' / 'real_levels' / model
    if not (src / 'info.json').exists():
        continue
    info = json.loads([ln for ln in (src / 'info.json').read_text().splitlines() if ln.startswith('{')][-1])
    out.setdefault('source_hash', info['source_hash'])
    files = sorted(glob.glob(str(src / 'sq1') + '/**/*counter_collection.csv', recursive=True), key=os.path.getmtime)
    acc = defaultdict(float); n = defaultdict(set)
    for r in csv.DictReader(open(files[-1])):
        k = r['Kernel_Name']
        if 'march_kernel' not in k or re.search(r'march_kernel<HIP_vector_type<\w+, 2u>, (true|\(bool\)1)', k):      # (the generic mop-up launch returns at once)
            continue
        acc[r['Counter_Name']] += float(r['Counter_Value']); n[r['Counter_Name']].add(r['Dispatch_Id'])
    waves = info['rows'] * info['rows'] / 64.0
    g = lambda c: acc[c] / max(1, len(n[c])) / waves
    ev = info['evaluated_samples_per_ray']
    out[model] = dict(levels=info['levels'], S=info['S'], K=info['K'], evaluated_samples_per_ray=ev, rows=info['rows'], valu_per_raywave=g('SQ_INSTS_VALU'),
                      valu_per_evaluated_sample=g('SQ_INSTS_VALU') / ev, valu_busy_frac=4 * g('SQ_ACTIVE_INST_VALU') / g('SQ_WAVE_CYCLES'), march_launches=len(n['SQ_INSTS_VALU']),
                      method='rocprofv3 --pmc SQ_INSTS_VALU SQ_ACTIVE_INST_VALU SQ_WAVE_CYCLES SQ_BUSY_CYCLES over tools/real_levels_run.py (light march launches only)')
(REPO / 'profiles' / f'{RND}_real_levels_counters.json').write_text(json.dumps(out, indent=1) + '\n')
print(json.dumps(out))
