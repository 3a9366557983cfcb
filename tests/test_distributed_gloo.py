"""CPU, world_size 2, gloo: the N>1 host path (row sharding + the MAX all-reduce that keeps nParts batch-global).
The per-shard arithmetic is played by the oracle here (no GPU in this container); the collective logic under test
is raider_amd.distributed, exactly what bench.py / multi-GPU callers use."""
import os
import socket
import sys
from pathlib import Path

import numpy as np
import pytest

REPO = Path(__file__).resolve().parent.parent
sys.path.insert(0, str(REPO))


def _free_port():
    s = socket.socket(); s.bind(('127.0.0.1', 0)); p = s.getsockname()[1]; s.close(); return p


def _worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from oracle import raider_oracle as O
        from raider_amd import distributed as D
        from raider_amd.engine import nparts_from_maxlen
        g = np.load(REPO / 'tests' / 'golden' / 'g5b_whole_vs_halves.npz')
        # cube broadcast from rank 0
        fields = None
        if rank == 0:
            c = O.synthetic_cube(50, 50, 40, seed=0)
            fields = {k: c[k] for k in ('xs', 'ys', 'zs', 'wet', 'hydro')}
        # the cube in ONE broadcast: header announced by the collective (rank 1 does not know the shape) ...
        axes, wet_t, hyd_t = D.broadcast_cube_packed(fields, src=0)
        got = dict(ys=axes[:50].numpy(), xs=axes[50:100].numpy(), zs=axes[100:].numpy(), wet=wet_t.numpy(), hydro=hyd_t.numpy())
        # ... and with the shape known to every rank (what bench.py does): the same bytes, no header round
        axes2, wet2, hyd2 = D.broadcast_cube_packed(fields, src=0, header=(50, 50, 40, 0, 40, 50, 50))
        assert bool((axes2 == axes).all()) and bool((wet2 == wet_t).all()) and bool((hyd2 == hyd_t).all())
        old = {k: v.numpy() for k, v in D.broadcast_cube_fields(fields, src=0).items()}
        assert all(np.array_equal(old[k], got[k]) for k in got) and got['wet'].dtype == np.float32 and got['wet'].shape == (40, 50, 50)
        ip = list(O.getInterpolators(got['xs'], got['ys'], got['zs'], got['wet'], got['hydro']))
        zref = float(g['zref'])
        # column shard here (the golden halves are column halves); shard_rows is exercised on the column count
        c0, nc = D.shard_rows(64, world, rank)
        xp, yp, inc = g['xpts'][c0:c0 + nc], g['ypts'], g['inc'][:, c0:c0 + nc]
        xx, yy = np.meshgrid(xp, yp)
        xyz = np.stack(O.lla2ecef(yy, xx, np.zeros_like(yy)), -1)
        los = O.look_vectors_from_inc_hd(inc, np.full(yy.shape, -167.9), yy, xx, 0.0)
        L, lo, hi = O.build_ray(got['zs'], 0.0, xyz, los, zref)
        local_max = L.reshape(L.shape[0], -1).max(1)
        maxlen, flags = D.reduce_partition(local_max, 2 | 4 | 8)
        D.check_partition_flags(flags)
        nparts = nparts_from_maxlen(maxlen)
        local_nparts = nparts_from_maxlen(local_max)
        look = lambda ht, llh, xyz_, yy_: los
        wet, hyd = O.build_cube_ray(xp, yp, np.array([0.0]), look, ip, MAX_TROPO_HEIGHT=zref, nParts_override=[nparts])
        err = max(np.abs(hyd[0] - g['hydro'][0][:, c0:c0 + nc]).max(), np.abs(wet[0] - g['wet'][0][:, c0:c0 + nc]).max())
        q.put((rank, bool(np.array_equal(nparts, g['nparts'])), bool(np.array_equal(local_nparts, g['nparts'])), float(err), int(flags)))
    finally:
        dist.destroy_process_group()


def test_two_rank_partition_allreduce():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    for rank, global_ok, local_same, err, flags in res:
        assert global_ok, 'all-reduced nParts must equal the whole-slice nParts of the reference'
        assert err < 1e-9, f'rank {rank}: shard driven by the global partition differs from the whole-slice reference by {err}'
        assert flags == (2 | 4 | 8)
    assert not all(r[2] for r in res), 'at least one shard-local partition must differ (otherwise the test is vacuous)'


def test_shard_rows_cover_exactly():
    from raider_amd.distributed import shard_rows
    for ny in (1, 7, 64, 4000, 10000):
        for world in (1, 2, 3, 8):
            blocks = [shard_rows(ny, world, r) for r in range(world)]
            assert blocks[0][0] == 0 and sum(b[1] for b in blocks) == ny
            for (a0, an), (b0, bn) in zip(blocks, blocks[1:]):
                assert a0 + an == b0
            assert max(b[1] for b in blocks) - min(b[1] for b in blocks) <= 1


def test_reduce_partition_single_process_passthrough():
    from raider_amd.distributed import check_partition_flags, reduce_partition
    m, f = reduce_partition(np.array([1.0, 2.0]), 6)
    assert np.array_equal(m, [1.0, 2.0]) and f == 6
    with pytest.raises(ValueError, match='geo2rdr did not converge'):
        check_partition_flags(1)
    with pytest.raises(ValueError):
        check_partition_flags(3)
    # the deferred check of the asynchronous path: K maxima + 4 flag elements (host copy here)
    from raider_amd.distributed import check_partition
    assert np.array_equal(check_partition(np.array([999.0, 1000.0, 1000.5, 0.0, 1.0, 1.0, 1.0])), [2, 2, 3])      # delay.py:283
    with pytest.raises(ValueError, match='NaN'):
        check_partition(np.array([999.0, 1.0, 1.0, 0.0, 0.0]))
    with pytest.raises(ValueError, match='geo2rdr'):
        check_partition(np.array([0.0, 0.0, 0.0, 0.0, 0.0]))


def test_pack_unpack_cube_roundtrip_f32_f64_any_layout():
    from raider_amd.distributed import pack_cube, unpack_cube
    rng = np.random.default_rng(0)
    for dt in (np.float32, np.float64):
        for shape in ((5, 6, 7), (7, 5, 6)):
            ys, xs, zs = rng.random(5), rng.random(6), rng.random(7)
            w, h = rng.random(shape).astype(dt), rng.random(shape).astype(dt)
            buf, hdr = pack_cube(ys, xs, zs, w, h)
            assert buf.numel() == 18 * 8 + 2 * w.nbytes and hdr == (5, 6, 7, 0 if dt == np.float32 else 1) + shape
            ax, w2, h2 = unpack_cube(buf, hdr)
            assert np.array_equal(ax.numpy(), np.concatenate([ys, xs, zs])) and np.array_equal(w2.numpy(), w) and np.array_equal(h2.numpy(), h)
    with pytest.raises(ValueError):
        unpack_cube(buf[:-8], hdr)
    with pytest.raises(ValueError):
        pack_cube(ys, xs, zs, w.astype(np.int32), h)


def _points_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from raider_amd import distributed as D

        class FakeCube:                     # (no GPU here: the per-rank arithmetic is a stand-in, the sharding logic is what is tested)
            def interp(self, p):
                return p[:, 0] * 2.0, p[:, 1] - 1.0
        pts = np.arange(3 * 1001, dtype=np.float64).reshape(1001, 3)
        p0, cnt, w, h = D.interp_points_sharded(FakeCube(), pts)
        # the two-epoch blend as part of the query (round 4): the rank's block picks blend-at-the-corners or blended-cube-then-gather by the
        # byte model (Cube.interp_blend(via_cube=...)); the stand-in returns the same numbers either way, which route was taken is recorded
        took = []

        class FakeEpoch(FakeCube):
            dtype = np.float32
            def __init__(self, shape): self.shape = shape
            def interp_blend(self, w1, other, w2, p, via_cube=False): took.append('cube' if via_cube else 'corners'); return self.interp(p)
        for shape in ((1000, 1000, 50), (10, 10, 5)):         # a big cube (the block is small against it), a tiny one
            b0, bc, bw, bh = D.interp_points_sharded(FakeEpoch(shape), pts, blend=(0.25, FakeEpoch(shape), 0.75))
            assert (b0, bc) == (p0, cnt) and np.array_equal(bw, w) and np.array_equal(bh, h)
        assert took == ['corners', 'cube'], took
        q.put((rank, p0, cnt, float(w.sum()), float(h.sum())))
    finally:
        dist.destroy_process_group()


def test_station_points_shard_without_a_collective():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_points_worker, args=(r, 3, port, q)) for r in range(3)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    pts = np.arange(3 * 1001, dtype=np.float64).reshape(1001, 3)
    assert [r[1] for r in res] == [0, 334, 668] and [r[2] for r in res] == [334, 334, 333]
    assert abs(sum(r[3] for r in res) - (pts[:, 0] * 2).sum()) < 1e-6 and abs(sum(r[4] for r in res) - (pts[:, 1] - 1).sum()) < 1e-6


def _height_worker(rank, world, port, q):
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port), RANK=str(rank), WORLD_SIZE=str(world))
    import torch.distributed as dist
    dist.init_process_group('gloo', rank=rank, world_size=world)
    try:
        from raider_amd import distributed as D
        from raider_amd.engine import Rays
        hts = np.array([[100.0 + 50.0 * rank, 900.0], [400.0, 2500.0 - 100.0 * rank]])
        rays = Rays.grid(np.array([-118.0, -117.0]), np.array([34.0, 33.0]), inc=35.0, hd=-167.9, hts=hts)
        lo = D.global_table_height(rays)                                   # MIN over the ranks' own minima
        lo2 = D.global_table_height(rays, -20.0)                           # an explicit, lower table height is kept
        one = D.global_table_height(Rays.grid(np.array([-118.0]), np.array([34.0]), inc=35.0, hd=-167.9), 250.0)
        err = None
        try:
            D.raytrace_slab_async(None, rays, None, 1000.0, None)
        except ValueError as e:
            err = str(e)
        q.put((rank, rays.ht_min, lo, lo2, one, err))
    finally:
        dist.destroy_process_group()


def test_per_pixel_heights_share_one_table_height_across_ranks():
    import torch.multiprocessing as mp
    ctx = mp.get_context('spawn')
    q = ctx.Queue()
    port = _free_port()
    procs = [ctx.Process(target=_height_worker, args=(r, 2, port, q)) for r in range(2)]
    for p in procs:
        p.start()
    res = sorted(q.get(timeout=240) for _ in procs)
    for p in procs:
        p.join(60)
        assert p.exitcode == 0
    assert [r[1] for r in res] == [100.0, 150.0]                           # each rank's own lowest pixel
    assert all(r[2] == 100.0 and r[3] == -20.0 and r[4] == 250.0 for r in res)
    assert all(r[5] and 'global_table_height' in r[5] for r in res)
