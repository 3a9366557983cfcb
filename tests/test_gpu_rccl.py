"""GPU: the RCCL (torch.distributed backend "nccl") path on the one GPU a box has - a ONE-RANK process group still goes through
ncclCommInitRank, ncclBroadcast and ncclAllReduce on device tensors, and through torch's ordering of the collective's internal
stream against the stream the library launches on (raider_amd adopts torch's current stream for device arrays).  What an 8-GPU
node adds is peers, not code: bench.py --gpus N runs exactly these calls.  (Two ranks cannot share a GPU under RCCL: the
two-rank tests use gloo on device tensors, tests/test_gpu_multirank.py.)"""
import json
import os
import subprocess
import sys
from pathlib import Path

import numpy as np
import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

SCRIPT = r'''
import os, sys, json, socket
import numpy as np, torch, torch.distributed as dist
sys.path.insert(0, %(root)r)
import raider_amd as R
from raider_amd import distributed as D
from raider_amd.synthetic import synthetic_cube, scene_grid
s = socket.socket(); s.bind(('127.0.0.1', 0)); port = s.getsockname()[1]; s.close()
os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
dev = torch.device('cuda', 0); torch.cuda.set_device(0)
dist.init_process_group('nccl', rank=0, world_size=1, device_id=dev)
assert dist.get_backend() == 'nccl' and dist.get_world_size() == 1 and D.is_distributed()
c = synthetic_cube(60, 70, 40, seed=0)
# one packed broadcast, header announced by the collective itself (receivers would not know the shape)
axes, wet, hyd = D.broadcast_cube_packed({k: c[k] for k in ('ys', 'xs', 'zs', 'wet', 'hydro')}, src=0, device=dev)
assert wet.is_cuda and wet.dtype == torch.float32 and tuple(wet.shape) == (40, 60, 70)
ax = axes.cpu().numpy()
assert np.array_equal(ax, np.concatenate([c['ys'], c['xs'], c['zs']]))
assert np.array_equal(wet.cpu().numpy(), c['wet']) and np.array_equal(hyd.cpu().numpy(), c['hydro'])
cube = R.Cube(ax[:60], ax[60:130], ax[130:], wet, hyd, order='zyx')
ref = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
zref = float(c['zs'].max() - 1)
xp, yp, inc, hd = scene_grid(300, 400)
xp = np.linspace(-120.5, -113.5, 400); yp = np.linspace(35.5, 30.5, 300)
xt, yt = torch.from_numpy(xp).to(dev), torch.from_numpy(yp).to(dev)
it = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc, (300, 400)))).to(dev)
los = R.Rays.grid(xt, yt, inc=it, hd=torch.full((300, 400), hd, dtype=torch.float64, device=dev)).look_vectors()
rays = R.Rays.grid(xt, yt, los=los)
w0, h0, nparts, flags = ref.raytrace(rays, 0.0, zref)
K = len(nparts)
part = torch.zeros(K + 4, dtype=torch.float64, device=dev)
outs = []
for rep in range(3):          # back to back: pass 1 -> ncclAllReduce(MAX) -> pass 2, nothing synchronised in between
    ow = torch.full((300, 400), -1.0, dtype=torch.float64, device=dev); oh = torch.full_like(ow, -1.0)
    D.raytrace_slab_async(cube, rays, 0.0, zref, part, out=(ow, oh))
    outs.append((ow, oh))
torch.cuda.synchronize()
ok = all(torch.equal(ow, w0) and torch.equal(oh, h0) for ow, oh in outs)
# the synchronous variant (host partition through the same backend: all_reduce on a device copy of the K+4 doubles)
w1, h1, np1 = D.raytrace_slab(cube, rays, 0.0, zref, device=dev)
ok2 = torch.equal(w1, w0) and torch.equal(h1, h0) and np.array_equal(np1, nparts)
# on a side stream: the collective must be ordered against THAT stream's kernels
st = torch.cuda.Stream()
with torch.cuda.stream(st):
    ow = torch.full((300, 400), -1.0, dtype=torch.float64, device=dev); oh = torch.full_like(ow, -1.0)
    D.raytrace_slab_async(cube, rays, 0.0, zref, part, out=(ow, oh))
st.synchronize()
ok3 = torch.equal(ow, w0) and torch.equal(oh, h0)
# config 5 on every rank: both epochs broadcast, blended on the rank's own device, the rank's block of the stations interpolated
c2 = synthetic_cube(60, 70, 40, seed=1)
ep = [{k: c[k] for k in ('ys', 'xs', 'zs', 'wet', 'hydro')}, {k: c2[k] for k in ('ys', 'xs', 'zs', 'wet', 'hydro')}]
blended = D.broadcast_and_blend(ep, (0.25, 0.75), src=0, device=dev)
want = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx').blend(0.25, R.Cube(c2['ys'], c2['xs'], c2['zs'], c2['wet'], c2['hydro'], order='zyx'), 0.75)
rng = np.random.default_rng(3)
pts = torch.from_numpy(np.stack([rng.uniform(30.2, 35.8, 5000), rng.uniform(-120.8, -113.2, 5000), rng.uniform(0, 4000, 5000)], -1)).to(dev)
p0, cnt, sw, sh = D.interp_points_sharded(blended, pts)
rw, rh = want.interp(pts)
ok4 = p0 == 0 and cnt == 5000 and torch.equal(sw, rw) and torch.equal(sh, rh) and bool(torch.isfinite(sw).all())
# the deferred error check of the asynchronous path: the partition gives the reference's nParts, or the reference's exception
ok5 = np.array_equal(D.check_partition(part), nparts)
bad = los.clone(); bad[5, 7] = float('nan')                      # one failed look vector: nParts is undefined (delay.py:283)
bw, bh = D.raytrace_slab_async(cube, R.Rays.grid(xt, yt, los=bad), 0.0, zref, part)
try:
    D.check_partition(part)
    ok5 = False
except ValueError as e:
    ok5 = ok5 and 'NaN' in str(e) and bool(torch.isnan(bw).all()) and bool(torch.isnan(bh).all())
nan_frac = float(torch.isnan(h0).double().mean())
print(json.dumps(dict(ok=bool(ok), ok2=bool(ok2), ok3=bool(ok3), ok4=bool(ok4), ok5=bool(ok5), K=K, nan_frac=nan_frac, partition_max=float(part[:K].max()))))
dist.destroy_process_group()
'''


def test_one_rank_rccl_group_broadcast_allreduce_bit_identical():
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    out = subprocess.run([sys.executable, '-c', SCRIPT % dict(root=str(ROOT))], capture_output=True, text=True, timeout=600, env=env)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['ok'], 'pass 1 -> ncclAllReduce -> pass 2 (asynchronous, device partition) differs from Cube.raytrace'
    assert res['ok2'], 'the synchronous slab path through the nccl backend differs from Cube.raytrace'
    assert res['ok3'], 'on a side stream the collective was not ordered against the ray kernels'
    assert res['ok4'], 'two epochs broadcast + blended per rank + sharded station gather differ from the single-process result'
    assert res['ok5'], 'check_partition: nParts from the device partition, or the reference\'s ValueError on a NaN look vector'
    assert res['K'] > 20 and res['partition_max'] > 1000.0 and res['nan_frac'] < 0.5


def _bench(tmp_path, tag, *args):
    out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--steps', '2', '--warmup', '1', '--cpu-sample', '0', '--no-e2e',
                          '--dump', str(tmp_path / tag)] + list(args), capture_output=True, text=True, timeout=900)
    assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
    lines = out.stdout.splitlines()          # the contract: stdout is ONE line, the JSON (library banners and logs go to stderr)
    assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
    return json.loads(lines[0])


def test_bench_n1_through_rccl_equals_plain_run(tmp_path):
    """bench.py's whole N > 1 code path (packed cube broadcast, device partition, MAX all-reduce per step, max-over-ranks timing)
    on a one-rank RCCL group gives the plain single-GPU run's delays bit for bit."""
    plain = _bench(tmp_path, 'plain', '--rows', '900', '--cols', '1100')
    rccl = _bench(tmp_path, 'rccl', '--rows', '900', '--cols', '1100', '--force-dist', '--backend', 'nccl')
    assert plain['config']['backend'] is None and rccl['config']['backend'] == 'nccl' and rccl['config']['world_size_seen_by_backend'] == 1
    a, b = np.load(tmp_path / 'plain.rank0.npz'), np.load(tmp_path / 'rccl.rank0.npz')
    assert np.array_equal(a['nparts'], b['nparts'])
    assert np.array_equal(a['hydro'], b['hydro']) and np.array_equal(a['wet'], b['wet']) and np.isfinite(a['hydro']).all()
    assert rccl['value'] > 0.5 * plain['value']        # a one-rank all-reduce between the passes must not serialise the step


def test_bench_c5_through_one_rank_rccl_group(tmp_path):
    """BASELINE configs[4] (`bench.py --workload c5`) through the nccl backend on the one GPU of this box: two packed epoch broadcasts
    over RCCL, per-rank blend, sharded station gather - the line the first 8-GPU SCALE run can use - equals the plain run bit for bit,
    and the oracle (blend_cubes + the scipy-RGI restatement) on a sample of the stations."""
    def run(tag, *args):
        out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--workload', 'c5', '--stations', '300000', '--steps', '2', '--warmup', '1',
                              '--dump', str(tmp_path / tag)] + list(args), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        lines = out.stdout.splitlines()
        assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
        return json.loads(lines[0])
    plain = run('plain', '--cpu-sample', '1')
    rccl = run('rccl', '--force-dist', '--backend', 'nccl', '--cpu-sample', '0')
    assert plain['config']['backend'] is None and rccl['config']['backend'] == 'nccl' and rccl['config']['world_size_seen_by_backend'] == 1
    assert plain['cpu_baseline']['kind'] == 'port' and plain['cpu_baseline']['gpu_vs_oracle_max_abs'] < 1e-9      # N units: 1e-9 of ~300
    assert plain['metric'].startswith('GNSS station points/sec') and plain['roofline']['source_hash'] == plain['roofline']['library_source_hash']
    a, b = np.load(tmp_path / 'plain.rank0.npz'), np.load(tmp_path / 'rccl.rank0.npz')
    assert np.array_equal(a['wet'], b['wet']) and np.array_equal(a['hydro'], b['hydro']) and np.isfinite(a['wet']).all()


def test_loaded_library_was_built_from_this_tree():
    """The binary the tests run carries the digest of the sources it was compiled from; it must be the tree's."""
    sys.path.insert(0, str(ROOT))
    import bench
    import raider_amd
    lib = raider_amd.load_library()
    assert lib.rdr_source_hash().decode() == bench.kernel_source_hash()
    cube_attrs = None
    from raider_amd.synthetic import synthetic_cube
    c = synthetic_cube(20, 20, 12)
    q = raider_amd.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    for which in (0, 1):
        cube_attrs = q.ray_kernel_attributes(which)
        assert 64 < cube_attrs['vgpr'] <= 128 and cube_attrs['scratch'] == 0 and cube_attrs['lds_dynamic'] > 1000, cube_attrs


def test_bench_c2_through_one_rank_rccl_group(tmp_path):
    """BASELINE configs[1] (`bench.py --workload c2`) through the nccl backend on the one GPU of this box: the f64 totals cube in one packed
    RCCL broadcast, the rank's point block through the intermediate delay cube - bit for bit the plain run."""
    def run(tag, *args):
        out = subprocess.run([sys.executable, str(ROOT / 'bench.py'), '--workload', 'c2', '--points', '400', '--steps', '2', '--warmup', '1', '--cpu-sample', '0', '--no-e2e',
                              '--dump', str(tmp_path / tag)] + list(args), capture_output=True, text=True, timeout=900)
        assert out.returncode == 0, (out.stdout[-2000:], out.stderr[-4000:])
        lines = out.stdout.splitlines()
        assert len(lines) == 1 and lines[0].startswith('{'), out.stdout[-2000:]
        return json.loads(lines[0])
    plain = run('plain')
    rccl = run('rccl', '--force-dist', '--backend', 'nccl')
    assert plain['config']['backend'] is None and rccl['config']['backend'] == 'nccl' and rccl['config']['world_size_seen_by_backend'] == 1
    assert 'one packed broadcast' in rccl['config']['parallelism'] and rccl['config']['shards'] == [[0, 160000]]
    a, b = np.load(tmp_path / 'plain.rank0.npz'), np.load(tmp_path / 'rccl.rank0.npz')
    assert np.array_equal(a['wet'], b['wet']) and np.array_equal(a['hydro'], b['hydro']) and np.isfinite(a['hydro']).all()
