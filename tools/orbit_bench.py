#!/usr/bin/env python3
"""Throughput of the zero-Doppler look-vector solver (orbit_los_kernel) on device-resident targets: a 4000 x 4000 scene at
ht = 0 against a synthetic 25-state-vector orbit arc (10 s spacing), as Raytracing(filename=...) drives it per height slice."""
import datetime as dt
import json
import sys
import time
from pathlib import Path

import numpy as np
import torch

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R                          # noqa: E402
from raider_amd.orbits import Orbit             # noqa: E402
from raider_amd.utilFcns import lla2ecef        # noqa: E402

n = int(sys.argv[1]) if len(sys.argv) > 1 else 4000
t = np.arange(-120.0, 121.0, 10.0)
r, w = 7.07e6, 2 * np.pi / 5900.0
lat0, lon0 = np.radians(33.0), np.radians(-100.0)
pos = np.stack([r * np.cos(lat0 + w * t) * np.cos(lon0), r * np.cos(lat0 + w * t) * np.sin(lon0), r * np.sin(lat0 + w * t)], -1)
vel = np.stack([-r * w * np.sin(lat0 + w * t) * np.cos(lon0), -r * w * np.sin(lat0 + w * t) * np.sin(lon0), r * w * np.cos(lat0 + w * t)], -1)
epoch = dt.datetime(2021, 1, 1, 6, 57, 0)
orb = Orbit([epoch + dt.timedelta(seconds=float(x)) for x in t], pos, vel)
xx, yy = np.meshgrid(np.linspace(-119.5, -115.5, n), np.linspace(34.5, 31.5, n))
xyz = torch.from_numpy(np.stack(lla2ecef(yy, xx, np.zeros_like(yy)), -1)).cuda()
los = orb.look_vectors(xyz); torch.cuda.synchronize()
t0 = time.perf_counter()
for _ in range(5):
    los = orb.look_vectors(xyz)
torch.cuda.synchronize()
dt_ = (time.perf_counter() - t0) / 5
print(json.dumps(dict(targets=n * n, ms=dt_ * 1e3, targets_per_s=n * n / dt_, nan_share=float(torch.isnan(los).double().mean()),
                      unit_norm_err=float((torch.linalg.norm(los, dim=-1) - 1).abs().max()))))
