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
