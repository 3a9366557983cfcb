"""GPU: the bench line itself - the fields the measurement contract asks for are there, typed, and consistent with each other (small
scene; the numbers are not asserted, their relations are)."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(*args):
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '3', '--warmup', '1', '--rows', '640', '--cols', '704', '--cpu-sample', '96'] + list(args),
                         capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = out.stdout.splitlines()          # the contract: stdout is ONE line, the JSON (library banners and logs go to stderr)
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_line_is_complete_and_self_consistent():
    d = _bench()
    for k in ('metric', 'value', 'unit', 'n_gpus', 'steps', 'warmup', 'ms_per_step', 'higher_is_better', 'scaling', 'vs_baseline', 'dtype', 'data', 'config', 'roofline',
              'cpu_baseline', 'end_to_end', 'parity'):
        assert k in d, k
    assert d['n_gpus'] == 1 and d['steps'] == 3 and d['warmup'] == 1 and d['higher_is_better'] is True and d['vs_baseline'] is None and d['dtype'] == 'f64'
    assert 'workload' in d['config'] and 'model' not in d['config'] and d['config']['rays_per_gpu'] == 640 * 704
    assert abs(d['value'] * d['ms_per_step'] * 1e-3 - 640 * 704) < 1.0                     # value = rays of one step / time of one step
    r = d['roofline']
    for k in ('bound', 'achieved', 'peak', 'unit', 'frac', 'traffic', 'frac_valu', 'frac_hbm_measured', 'traffic_over_compulsory', 'vgpr', 'lds_bytes', 'scratch_bytes',
              'source_hash', 'library_source_hash', 'march_ms_per_step', 'crossings_ms_per_step'):
        assert k in r, k
    assert r['source_hash'] == r['library_source_hash'] and len(r['source_hash']) == 16   # the binary that ran is the tree's
    assert r['vgpr'] == 128 and r['scratch_bytes'] == 0 and 10000 < r['lds_bytes'] < 65536  # from the loaded code object, not a profiler column
    assert 0 < r['march_ms_per_step'] < d['ms_per_step'] and 0 < r['crossings_ms_per_step'] < r['march_ms_per_step']
    if r['frac'] is not None:                                                               # a digest of this source tree is committed
        assert r['frac'] == r['frac_valu'] and 0.1 < r['frac'] <= 1.0 and r['counters_source'].startswith('profiles/')
    # round 4: how GOOD the kernel is beside how busy - the useful-work figure, the clock the chip really ran at, and their relations
    for k in ('useful_flops_frac', 'useful_TFLOPs', 'fp64_vector_peak_TFLOPs', 'valu_per_reference_sample', 'clock_GHz_measured', 'clock_GHz_assumed_by_peak',
              'frac_at_measured_clock', 'useful_flops_frac_at_measured_clock', 'flop_model'):
        assert k in r, k
    S, n = d['config']['samples_per_ray_S'], d['config']['rays_per_gpu']
    assert abs(r['fp64_vector_peak_TFLOPs'] - 78.6432) < 1e-3 and r['clock_GHz_assumed_by_peak'] == 2.4
    assert abs(r['useful_TFLOPs'] * 1e12 - 110.0 * S * n / (r['march_ms_per_step'] * 1e-3)) < 1e-6 * r['useful_TFLOPs'] * 1e12
    assert abs(r['useful_flops_frac'] - r['useful_TFLOPs'] / r['fp64_vector_peak_TFLOPs']) < 1e-12 and 0.0 < r['useful_flops_frac'] < 1.0
    assert 1.0 < r['clock_GHz_measured'] < 2.6                                              # the shader clock during the timed steps, from the chip's counters
    assert abs(r['useful_flops_frac_at_measured_clock'] - r['useful_flops_frac'] * 2.4 / r['clock_GHz_measured']) < 1e-12
    if r['frac'] is not None:
        assert abs(r['frac_at_measured_clock'] - r['frac'] * 2.4 / r['clock_GHz_measured']) < 1e-12 and r['frac_at_measured_clock'] <= 1.02
        assert abs(r['valu_per_reference_sample'] - r['valu_instr_per_raywave'] / S) < 1e-9
        assert r['valu_per_reference_sample'] < r['valu_per_evaluated_sample']             # shared segment ends are evaluated once
    p = d['parity']                                                                         # a full-scene record is cited only when made with THESE kernels
    assert p['record'] is None or p['source_hash'] == r['source_hash']
    e = d['end_to_end']
    assert e['bit_identical_to_device_path'] is True and e['value'] < d['value'] and e['h2d_bytes'] == 640 * 704 * 24 and e['d2h_bytes'] == 640 * 704 * 16
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] >= 1 and c['value'] > 0 and c['gpu_vs_oracle_max_abs_m'] < 1e-6 and 'sample' in c


def test_bench_per_pixel_heights_line():
    d = _bench('--per-pixel-ht', '--no-e2e')
    assert d['config']['workload'].startswith('c3b') and 'end_to_end' not in d
    assert d['roofline']['kernel'].endswith('true>') and d['roofline']['frac'] is None     # (no counter digest is kept for the secondary workload)
    assert d['cpu_baseline']['gpu_vs_oracle_max_abs_m'] < 1e-6 and d['config']['nan_fraction'] == 0.0
