#!/usr/bin/env python3
"""gpurun_out/e2e (tools/e2e_round.sh) -> profiles/r03_e2e.json, r03_write_probe.txt, r03_pin_probe.txt, r03_bench_per_pixel.json"""
import json
import re
import shutil
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
src = REPO / 'gpurun_out' / 'e2e'
out = {}
for f in sorted(src.glob('tropo_*.json')):
    d = json.loads(f.read_text().strip().splitlines()[-1])
    best = min(v for k, v in d.items() if k.startswith('run') and k != 'run0_s')
    out[f.stem] = dict(what='tropo_delay(datetime, processed-cube NetCDF on disk, grid AOI, Raytracing(inc raster, heading), heights) -> NumPy delay cubes '
                            '(tools/e2e_tropo_delay.py; best of 5 warm calls, each with fresh result arrays)',
                       rays=d['rays'], ms=best * 1e3, rays_per_s=d['rays'] / best, first_call_ms=d['run0_s'] * 1e3, mean_hydro_m=d['mean_hydro'], nan=d['nan'])
zf = src / 'zenith_1000x1000x40.json'
if zf.exists():
    d = json.loads(zf.read_text().strip().splitlines()[-1])
    best = min(v for k, v in d.items() if k.startswith('run') and k != 'run0_s')
    out['zenith_1000x1000x40'] = dict(what='tropo_delay(datetime, processed-cube NetCDF on disk, 1000 x 1000 grid AOI, Zenith(), 40 heights) -> NumPy delay cubes: BASELINE configs[1] '
                                           'sizes through the host API (tools/e2e_zenith.py)', points=d['rays'], ms=best * 1e3, points_per_s=d['rays'] / best)
line = (src / 'orbit_1000x1000x8.json').read_text().strip().splitlines()[-1]
m = re.search(r'= ([\d.]+) M rays in ([\d.]+) ms', line)
out['orbit_1000x1000x8'] = dict(what='_build_cube_ray through Raytracing(<orbit file>): grid -> ECEF -> zero-Doppler look vectors -> ray batch on the device, NumPy cubes back '
                                     '(tools/e2e_orbit.py)', rays=float(m.group(1)) * 1e6, ms=float(m.group(2)), rays_per_s=float(m.group(1)) * 1e6 / (float(m.group(2)) * 1e-3))
b = json.loads([ln for ln in (src / 'bench.json').read_text().splitlines() if ln.startswith('{')][-1])
out['bench_numpy_boundary'] = b.get('end_to_end')
out['bench_device_resident'] = dict(rays_per_s=b['value'], ms_per_step=b['ms_per_step'], frac_valu=b['roofline']['frac_valu'], frac_hbm_measured=b['roofline']['frac_hbm_measured'],
                                    traffic_over_compulsory=b['roofline']['traffic_over_compulsory'], counters_source=b['roofline']['counters_source'])
pp = json.loads([ln for ln in (src / 'bench_per_pixel.json').read_text().splitlines() if ln.startswith('{')][-1])
out['bench_per_pixel_heights'] = dict(rays_per_s=pp['value'], ms_per_step=pp['ms_per_step'], march_ms=pp['roofline']['march_ms_per_step'], crossings_ms=pp['roofline']['crossings_ms_per_step'],
                                      vgpr=pp['roofline']['vgpr'], scratch_bytes=pp['roofline']['scratch_bytes'], gpu_vs_oracle_max_abs_m=pp['cpu_baseline']['gpu_vs_oracle_max_abs_m'],
                                      end_to_end=pp.get('end_to_end'))
sl = [ln for ln in (src / 'bench_slices.txt').read_text().splitlines() if ln.startswith('{')]
if sl:
    out['bench_slices'] = json.loads(sl[-1])
(REPO / 'profiles' / 'r03_e2e.json').write_text(json.dumps(out, indent=1) + '\n')
shutil.copy(src / 'write_probe.txt', REPO / 'profiles' / 'r03_write_probe.txt')
shutil.copy(src / 'pin_probe.txt', REPO / 'profiles' / 'r03_pin_probe.txt')
shutil.copy(src / 'bench_per_pixel.json', REPO / 'profiles' / 'r03_bench_per_pixel.json')
print(json.dumps(out, indent=1)[:3000])
