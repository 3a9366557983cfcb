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
    assert 120 <= r["vgpr"] <= 128 and r['scratch_bytes'] == 0 and 10000 < r['lds_bytes'] < 65536  # from the loaded code object, not a profiler column
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
    # round 5: the instruction mix is MEASURED (SQ class counters through the same digest); three different fractions, related as they must be
    for k in ('executed_fp64_flops_frac', 'executed_fp64_TFLOPs', 'frac_class_priced', 'valu_mix', 'valu_mix_per_raywave', 'valu_mix_source'):
        assert k in r, k
    if r['frac'] is not None and r['valu_mix'] is not None:
        mix = r['valu_mix_per_raywave']
        assert abs(sum(r['valu_mix'].values()) - 1.0) < 1e-9 and r['valu_mix']['fp64'] > 0.4 and r['valu_mix']['cvt'] > 0.1
        assert abs(mix['valu'] - r['valu_instr_per_raywave']) < 0.01 * mix['valu']                 # the class passes saw the same kernel as the VALU pass
        assert abs(mix['fp64_flops_per_lane'] - (2 * mix['fp64_fma'] + mix['fp64_add'] + mix['fp64_mul'] + mix['fp64_trans'])) < 1.0
        assert abs(mix['flops_fp64_counter_per_wave'] - mix['fp64_flops_per_lane']) < 0.02 * mix['fp64_flops_per_lane']   # SQ_INSTS_VALU_FLOPS_FP64 agrees with the class counts
        assert 0.0 < r['executed_fp64_flops_frac'] <= r['frac'] + 1e-12                          # executed flops can never exceed issue busy-ness
        assert r['executed_fp64_flops_frac'] < r['useful_flops_frac']                            # ... and the kernel executes FEWER flops than the reference's algorithm counts
        assert r['frac_class_priced'] <= r['frac'] + 1e-12                                       # cheaper classes priced at their own rate
        assert abs(r['executed_fp64_TFLOPs'] * 1e12 - mix['fp64_flops_per_lane'] * n / (r['march_ms_per_step'] * 1e-3)) < 1e-6 * r['executed_fp64_TFLOPs'] * 1e12
    # round 6: SURVEY 8(d)'s models over the WHOLE step beside the busy-ness figure, which is labelled as what it is
    for k in ('frac_is', 'survey_flops_frac_step', 'survey_bytes_over_hbm_peak_step', 'survey_bytes_note', 'dependent_chain_note'):
        assert k in r, k
    assert r['frac_is'] == 'valu_issue_busy'
    assert 0 < r['survey_flops_frac_step'] <= r['useful_flops_frac']                       # step time >= march time
    assert abs(r['survey_bytes_over_hbm_peak_step'] * 8.0e12 * d['ms_per_step'] * 1e-3 - (64 * S + 64) * n) < 1e-6 * (64 * S + 64) * n
    assert len(d['config']['devices']) == 1 and d['config']['devices'][0]['rank'] == 0 and d['config']['distinct_devices'] == 1
    sec = d['secondary']                                                                    # the two gather workloads beside the headline
    rl = sec['real_levels']                                                                 # the same scene on the REAL level heights of ERA5 / HRRR
    for tag, nlev in (('era5_145', 145), ('hrrr_57', 57)):
        assert 'error' not in rl[tag], rl[tag]
        e_ = rl[tag]
        assert e_['levels'] == nlev and e_['K'] < nlev and e_['S'] >= 2 * e_['K'] and e_['evaluated_samples_per_ray'] == e_['S'] - (e_['K'] - 1)
        assert e_['rays_per_s'] > 1e8 and e_['gpu_vs_oracle_max_abs_m'] < 1e-9 and e_['nan_fraction'] == 0.0
        assert e_['march_ms_per_step'] + e_['crossings_ms_per_step'] <= e_['ms_per_step'] * 1.001
    assert rl['era5_145']['S'] > rl['hrrr_57']['S']
    for wl, unit in (('c2', 'points/s'), ('c5', 'points/s')):
        assert 'error' not in sec[wl], sec[wl]
        assert sec[wl]['unit'] == unit and sec[wl]['value'] > 1e8 and sec[wl]['roofline_bound'] == 'hbm' and sec[wl]['roofline_frac'] > 0
        assert sec[wl]['kernels_ms_per_step'] <= sec[wl]['ms_per_step'] * 1.001
    assert sec['c2']['gpu_vs_oracle_max_abs_m'] < 1e-12 and sec['c5']['gpu_vs_oracle_max_abs'] < 1e-9    # metres of delay; N-units of refractivity
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


def test_c2_line():
    """`bench.py --workload c2` = BASELINE configs[1] as a driver-runnable line: points/s with the inputs resident in HBM, an HBM roofline on
    SURVEY 8(d)'s 168 B per query, the host-buffer route beside it (bit-identical), the NumPy oracle timed on the same box."""
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--workload', 'c2', '--steps', '5', '--warmup', '2'], capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
    d = json.loads(lines[0])
    assert d['unit'] == 'points/s' and d['n_gpus'] == 1 and d['dtype'] == 'f64' and 'configs[1]' in d['config']['workload'] and d['vs_baseline'] is None
    assert d['config']['points_all_gpus'] == 1000 * 1000 and abs(d['value'] * d['ms_per_step'] * 1e-3 - 1e6) < 1.0
    r = d['roofline']
    assert r['bound'] == 'hbm' and r['unit'] == 'GB/s' and r['peak'] == 8000.0 and abs(r['frac'] - r['achieved'] / r['peak']) < 1e-12
    assert abs(r['algorithmic_bytes_per_step'] - (168.0 * 1e6 + 200.0 * d['config']['intermediate_nodes'])) < 1.0
    assert r['kernels_ms_per_step'] < d['ms_per_step'] and r['source_hash'] == r['library_source_hash']
    if r['traffic'] is not None:
        assert r['counters_source'].startswith('profiles/') and 0 < r['frac_hbm_measured'] < 1.0
    e = d['end_to_end']
    assert e['bit_identical_to_device_path'] is True and e['value'] < d['value'] and e['h2d_bytes'] == 24_000_000
    c = d['cpu_baseline']
    assert c['kind'] == 'port' and c['cores'] == 1 and c['gpu_vs_oracle_max_abs_m'] < 1e-12 and c['nan_masks_equal'] is True
    assert 1.5 < d['config']['mean_hydro_m'] < 4.0 and d['config']['nan_fraction'] == 0.0
