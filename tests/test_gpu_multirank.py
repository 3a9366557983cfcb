"""GPU: the N > 1 path of bench.py, launched exactly as the driver launches N = 1 (`python bench.py --gpus N ...`): bench.py
re-executes itself through torch.distributed.run.  On a 1-GPU box the two ranks share the device (RCCL refuses two ranks on one
GPU, so the collectives go through gloo on device-resident tensors - the same code path, rdr_ray_prepass_device ->
all_reduce(MAX) -> rdr_ray_march_device, as with nccl on an 8-GPU node)."""
import json
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent


def _bench(tmp_path, tag, *args):
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '2', '--warmup', '1', '--cpu-sample', '0', '--cols', '1200',
                          '--no-e2e', '--dump', str(tmp_path / tag)] + list(args), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = out.stdout.splitlines()          # the contract: stdout is ONE line, the JSON (library banners and logs go to stderr)
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_self_launches_two_ranks_and_matches_one_rank(tmp_path):
    """`python bench.py --gpus 2` (no torchrun): two ranks trace the two halves of a 1400x1200 scene with ONE MAX all-reduce of
    the K+4-element partition per step; the concatenated slabs equal the single-rank result of the whole scene bit for bit
    (delay.py:283 semantics across ranks), which shard-local nParts would not give (golden g5b)."""
    one = _bench(tmp_path, 'one', '--gpus', '1', '--rows', '1400')
    two = _bench(tmp_path, 'two', '--gpus', '2', '--rows', '700')
    assert one['n_gpus'] == 1 and two['n_gpus'] == 2 and two['config']['ranks'] == 2
    assert two['config']['rays_per_gpu'] == 700 * 1200 and two['value'] > 0 and two['scaling'] == 'weak'
    assert two['config']['backend'] in ('nccl', 'gloo')
    # weak scaling: the one-GPU reference is ONE slab traced alone; efficiency = t1 / tN
    assert abs(two['scaling_efficiency'] - two['one_gpu_same_scene']['ms_per_step'] / two['ms_per_step']) < 1e-12 and 'strong_scaling_efficiency' not in two
    assert len(two['config']['devices']) == 2 and two['parity_sample']['max_abs_hydro_m'] < 1e-9
    a = np.load(tmp_path / 'one.rank0.npz')
    b0, b1 = np.load(tmp_path / 'two.rank0.npz'), np.load(tmp_path / 'two.rank1.npz')
    assert np.array_equal(a['nparts'], b0['nparts']) and np.array_equal(a['nparts'], b1['nparts'])
    assert np.array_equal(a['hydro'], np.concatenate([b0['hydro'], b1['hydro']])) and np.array_equal(a['wet'], np.concatenate([b0['wet'], b1['wet']]))
    assert np.isfinite(a['hydro']).all()


def test_strong_scaling_is_the_default_for_n_gt_1(tmp_path):
    """`python bench.py --gpus 2` with no --rows = BASELINE configs[3] semantics: ONE scene split into contiguous row blocks
    (shard_rows), `value` counts the scene's rays once, `scaling` = "strong"; here on a reduced 1401 x 1200 scene (uneven split:
    701 + 700 rows) against the one-rank run of the whole scene."""
    one = _bench(tmp_path, 'one', '--gpus', '1', '--rows', '1401')
    two = _bench(tmp_path, 'two', '--gpus', '2', '--total-rows', '1401')
    assert two['scaling'] == 'strong' and two['config']['rays_per_step_all_gpus'] == 1401 * 1200 and two['config']['rays_per_gpu'] == 701 * 1200
    assert 'configs[3]' in two['config']['workload'] and two['config']['world_size_seen_by_backend'] == 2
    assert abs(two['value'] * two['ms_per_step'] * 1e-3 - 1401 * 1200) < 1.0          # value = scene rays / step time
    a = np.load(tmp_path / 'one.rank0.npz')
    b0, b1 = np.load(tmp_path / 'two.rank0.npz'), np.load(tmp_path / 'two.rank1.npz')
    assert b0['hydro'].shape == (701, 1200) and b1['hydro'].shape == (700, 1200)
    assert np.array_equal(a['nparts'], b0['nparts']) and np.array_equal(a['nparts'], b1['nparts'])
    assert np.array_equal(a['hydro'], np.concatenate([b0['hydro'], b1['hydro']])) and np.array_equal(a['wet'], np.concatenate([b0['wet'], b1['wet']]))


def test_per_pixel_heights_across_two_ranks(tmp_path):
    """A scene on a DEM sharded over two ranks: the level table starts at the lowest pixel of the WHOLE scene (one MIN all-reduce), the
    per-level maxima are reduced over the ranks as for a slice - the slabs equal the one-rank result bit for bit."""
    one = _bench(tmp_path, 'one', '--gpus', '1', '--rows', '900', '--per-pixel-ht')
    two = _bench(tmp_path, 'two', '--gpus', '2', '--total-rows', '900', '--per-pixel-ht')
    assert two['scaling'] == 'strong' and two['config']['workload'].startswith('c3b')
    a = np.load(tmp_path / 'one.rank0.npz')
    b0, b1 = np.load(tmp_path / 'two.rank0.npz'), np.load(tmp_path / 'two.rank1.npz')
    assert np.array_equal(a['nparts'], b0['nparts']) and np.array_equal(a['nparts'], b1['nparts'])
    assert np.array_equal(a['hydro'], np.concatenate([b0['hydro'], b1['hydro']])) and np.array_equal(a['wet'], np.concatenate([b0['wet'], b1['wet']]))
    assert np.isfinite(a['hydro']).all()


def test_c5_station_workload_across_two_ranks(tmp_path):
    """`bench.py --workload c5 --gpus 2` = BASELINE configs[4] launched as the driver would launch it: the two epochs go out in two
    packed broadcasts, every rank blends them and gathers its contiguous block of the station list - no data-path collective.  The
    two blocks together equal the one-rank result bit for bit; `value` counts the job's stations once (strong scaling)."""
    def run(tag, *args):
        out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--workload', 'c5', '--stations', '400001', '--steps', '2', '--warmup', '1', '--cpu-sample', '0',
                              '--dump', str(tmp_path / tag)] + list(args), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        lines = out.stdout.splitlines()
        assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
        return json.loads(lines[0])
    one = run('one', '--gpus', '1')
    two = run('two', '--gpus', '2')
    assert one['unit'] == 'points/s' and one['n_gpus'] == 1 and two['n_gpus'] == 2 and two['scaling'] == 'strong'
    assert 'configs[4]' in two['config']['workload'] and two['config']['stations_all_gpus'] == 400001 and two['config']['stations_this_rank'] == 200001
    assert two['config']['world_size_seen_by_backend'] == 2 and abs(two['value'] * two['ms_per_step'] * 1e-3 - 400001) < 1.0
    assert two['roofline']['bound'] == 'hbm' and 0 < two['roofline']['frac'] < 1 and two['roofline']['unit'] == 'GB/s'
    a = np.load(tmp_path / 'one.rank0.npz')
    b0, b1 = np.load(tmp_path / 'two.rank0.npz'), np.load(tmp_path / 'two.rank1.npz')
    assert int(b0['cnt']) == 200001 and int(b1['p0']) == 200001 and int(b1['cnt']) == 200000
    assert np.array_equal(a['wet'], np.concatenate([b0['wet'], b1['wet']])) and np.array_equal(a['hydro'], np.concatenate([b0['hydro'], b1['hydro']]))
    assert np.isfinite(a['hydro']).all() and a['hydro'].mean() > 50.0          # (refractivities, not delays: N units)


# ---- world = 8 on ONE device (round 5): every branch an 8-GPU node will take, rehearsed --------------------------------------------------
def _run(tmp_path, tag, *args, timeout=1200):
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '2', '--warmup', '1', '--cpu-sample', '0', '--no-e2e', '--dump', str(tmp_path / tag)] + list(args),
                         capture_output=True, text=True, timeout=timeout)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = out.stdout.splitlines()
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
    return json.loads(lines[0])


def _self_contained(line, world, units):
    """round 6: an N > 1 line carries its own one-GPU reference (same job, same node, same run), the identity of every rank's device and a
    parity sample - the first real 8-GPU SCALE record must be readable without comparing it with another line."""
    cfg = line['config']
    assert len(cfg['devices']) == world and [d['rank'] for d in cfg['devices']] == list(range(world))
    assert all(set(d) >= {'rank', 'hip_device', 'uuid', 'pci', 'name'} for d in cfg['devices'])
    assert 1 <= cfg['distinct_devices'] <= world
    if cfg['backend'] == 'nccl':
        assert cfg['distinct_devices'] == world                                             # RCCL: one rank per physical GPU
    one = line['one_gpu_same_scene']
    assert one['ms_per_step'] > 0 and abs(one['value'] * one['ms_per_step'] * 1e-3 - units) < 1.0 and 1 <= one['steps'] <= 3
    assert abs(line['strong_scaling_efficiency'] - one['ms_per_step'] / (world * line['ms_per_step'])) < 1e-12
    assert line['scaling_efficiency'] == line['strong_scaling_efficiency'] and 0 < line['scaling_efficiency'] < 1.5
    assert 'parity_sample' in line


def _covers(shards, total):
    """contiguous blocks, in rank order, covering [0, total) exactly once"""
    pos = 0
    for r0, cnt in shards:
        if r0 != pos or cnt < 0:
            return False
        pos += cnt
    return pos == total


def test_world8_strong_scaling_of_one_scene_matches_one_rank(tmp_path):
    """`python bench.py --gpus 8` as the driver's scaling run launches it (self-launch through torch.distributed.run, eight ranks): BASELINE
    configs[3] semantics on a reduced 2005 x 4000 scene - uneven row blocks (251 x 5 + 250 x 3), ONE packed cube broadcast, ONE MAX all-reduce
    of K+4 doubles per step, max-over-ranks timing.  On a one-GPU box the ranks share the device (`--backend auto` -> gloo with device-resident
    collective tensors: RCCL refuses duplicates); on an 8-GPU node the same command runs over RCCL.  The eight slabs equal the one-rank run of
    the whole scene bit for bit; the line reports the world the backend saw, every rank's block and every rank's own time."""
    one = _run(tmp_path, 'one', '--gpus', '1', '--rows', '2005', '--cols', '4000', '--no-secondary')
    eight = _run(tmp_path, 'eight', '--gpus', '8', '--total-rows', '2005', '--cols', '4000')
    cfg = eight['config']
    assert eight['n_gpus'] == 8 and eight['scaling'] == 'strong' and cfg['ranks'] == 8 and cfg['world_size_seen_by_backend'] == 8
    assert cfg['backend'] in ('nccl', 'gloo') and cfg['ranks_per_device'] * cfg['devices_visible'] >= 8
    assert 'configs[3]' in cfg['workload'] and cfg['rays_per_step_all_gpus'] == 2005 * 4000
    assert _covers(cfg['shards'], 2005) and [c for _, c in cfg['shards']] == [251] * 5 + [250] * 3
    assert len(cfg['rank_ms_per_step']) == 8 and all(t > 0 for t in cfg['rank_ms_per_step'])
    assert abs(max(cfg['rank_ms_per_step']) - eight['ms_per_step']) < 1e-6 * eight['ms_per_step']       # the line's time IS the slowest rank's
    assert abs(eight['value'] * eight['ms_per_step'] * 1e-3 - 2005 * 4000) < 1.0
    assert 'secondary' not in eight                                                                       # (an appendix of the one-GPU line only)
    _self_contained(eight, 8, 2005 * 4000)
    assert eight['one_gpu_same_scene']['same_bits_as_sharded_run'] is True
    ps = eight['parity_sample']
    assert ps['block'] == [250, 250] and ps['rays_compared_all_ranks'] == 8 * 250 * 250 and len(ps['per_rank_max_abs_m']) == 8
    assert ps['max_abs_wet_m'] < 1e-9 and ps['max_abs_hydro_m'] < 1e-9 and ps['nan_mask_mismatches'] == 0
    a = np.load(tmp_path / 'one.rank0.npz')
    parts = [np.load(tmp_path / f'eight.rank{r}.npz') for r in range(8)]
    assert [int(p['row0']) for p in parts] == [r0 for r0, _ in cfg['shards']]
    assert all(np.array_equal(a['nparts'], p['nparts']) for p in parts)
    assert np.array_equal(a['hydro'], np.concatenate([p['hydro'] for p in parts])) and np.array_equal(a['wet'], np.concatenate([p['wet'] for p in parts]))
    assert np.isfinite(a['hydro']).all()


def test_world8_station_and_point_workloads_match_one_rank(tmp_path):
    """`--workload c5` (configs[4]) and `--workload c2` (configs[1]) on eight ranks: the cubes go out in packed broadcasts, every rank takes its
    contiguous block of the point list, there is no data-path collective; blocks cover the list exactly and equal the one-rank result bit for bit."""
    one = _run(tmp_path, 'c5one', '--gpus', '1', '--workload', 'c5', '--stations', '400003')
    eight = _run(tmp_path, 'c5eight', '--gpus', '8', '--workload', 'c5', '--stations', '400003')
    cfg = eight['config']
    assert eight['n_gpus'] == 8 and eight['scaling'] == 'strong' and cfg['world_size_seen_by_backend'] == 8 and _covers(cfg['shards'], 400003)
    assert len(cfg['rank_ms_per_step']) == 8 and abs(eight['value'] * eight['ms_per_step'] * 1e-3 - 400003) < 1.0
    _self_contained(eight, 8, 400003)
    assert eight['one_gpu_same_scene']['same_bits_as_sharded_run'] is True and eight['parity_sample']['max_abs'] < 1e-9
    assert eight['parity_sample']['points_compared_all_ranks'] == 8 * 2048
    a = np.load(tmp_path / 'c5one.rank0.npz')
    parts = [np.load(tmp_path / f'c5eight.rank{r}.npz') for r in range(8)]
    assert [int(p['p0']) for p in parts] == [r0 for r0, _ in cfg['shards']]
    assert np.array_equal(a['wet'], np.concatenate([p['wet'] for p in parts])) and np.array_equal(a['hydro'], np.concatenate([p['hydro'] for p in parts]))
    one = _run(tmp_path, 'c2one', '--gpus', '1', '--workload', 'c2', '--points', '301')
    eight = _run(tmp_path, 'c2eight', '--gpus', '8', '--workload', 'c2', '--points', '301')
    assert one['unit'] == 'points/s' and 'configs[1]' in eight['config']['workload'] and eight['config']['world_size_seen_by_backend'] == 8
    assert eight['config']['points_all_gpus'] == 301 * 301 and abs(eight['value'] * eight['ms_per_step'] * 1e-3 - 301 * 301) < 1.0
    assert eight['roofline']['bound'] == 'hbm' and eight['roofline']['frac'] > 0
    _self_contained(eight, 8, 301 * 301)
    assert eight['parity_sample']['max_abs_m'] < 1e-12 and eight['parity_sample']['nan_masks_equal'] is True
    a = np.load(tmp_path / 'c2one.rank0.npz')
    parts = [np.load(tmp_path / f'c2eight.rank{r}.npz') for r in range(8)]
    assert sum(int(p['cnt']) for p in parts) == 301 * 301
    assert np.array_equal(a['wet'], np.concatenate([p['wet'] for p in parts])) and np.array_equal(a['hydro'], np.concatenate([p['hydro'] for p in parts]))
    assert np.isfinite(a['hydro']).all() and 1.5 < a['hydro'].mean() < 4.0                          # metres of slant delay at 39 deg incidence


def test_nccl_preflight_is_one_clear_message(tmp_path):
    """`bench.py --gpus 8 --backend nccl` on a box with fewer than eight devices: ONE sentence saying what is needed, a non-zero exit code, no
    torchrun stack - the first real 8-GPU run must not die on something a dry run could have told."""
    import torch
    if torch.cuda.device_count() >= 8:
        pytest.skip('eight devices are visible: the nccl path itself runs (test_gpu_rccl.py)')
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--gpus', '8', '--backend', 'nccl', '--steps', '1', '--warmup', '0'], capture_output=True, text=True, timeout=300)
    assert out.returncode != 0 and out.stdout.strip() == ''
    assert 'needs one GPU per rank: 8 ranks asked' in out.stderr and '--backend auto' in out.stderr
    assert 'Traceback' not in out.stderr and 'ChildFailedError' not in out.stderr
