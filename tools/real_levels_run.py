#!/usr/bin/env python3
"""One model's real-levels scene (bench.real_levels_measure) for a profiler pass:  rocprofv3 --pmc ... -- python tools/real_levels_run.py era5|hrrr [rows]"""
import json
import sys
from pathlib import Path

import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import bench  # noqa: E402
import raider_amd as R  # noqa: E402

model = sys.argv[1]
rows = int(sys.argv[2]) if len(sys.argv) > 2 else 2000
ctx = R.Context.default()
r = bench.real_levels_measure(ctx, torch.device('cuda', 0), model, rows, rows, steps=2, block=0)
r.update(model=model, rows=rows, source_hash=bench.kernel_source_hash())
print(json.dumps(r))
