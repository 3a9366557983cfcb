"""Multi-GPU sharding of the ray-traced path: one process per GPU, torch.distributed (RCCL on MI355X, gloo on CPU).

The path shards embarrassingly over rays EXCEPT for two batch-level quantities the reference computes over
the whole (ny,nx) slice:
  * nParts[k] = ceil(max_over_slice(ray_length_k)/MAX_SEGMENT_LENGTH)+1      (delay.py:283)
  * the all-pixels z-clamp predicates                                         (delay.py:306-311)
so between pass 1 and pass 2 every rank joins ONE tiny all-reduce (MAX over K doubles + 4 flag words).
Skipping it changes hydro delays by up to 3e-5 m (tests/golden/g5b).  The weather cube is broadcast once
from rank 0; there is no output collective (each rank keeps / writes its own slab).
"""
import numpy as np

from ._lib import FLAG_ANY_FINITE, FLAG_ANY_NAN
from .engine import nparts_from_maxlen


def shard_rows(ny, world, rank):
    """Contiguous row block [row0, row0+nrows) of rank `rank` (row blocks keep lateral locality)."""
    base, rem = divmod(int(ny), int(world))
    nrows = base + (1 if rank < rem else 0)
    row0 = rank * base + min(rank, rem)
    return row0, nrows


def _dist():
    import torch.distributed as dist
    return dist


def is_distributed():
    """A process group exists (of ANY size: a one-rank group still issues its collectives, which is how the RCCL path is
    exercised on a single-GPU box)."""
    try:
        dist = _dist()
        return dist.is_available() and dist.is_initialized()
    except ImportError:
        return False


def reduce_partition(maxlen, flags, group=None, device=None):
    """All-reduce the pass-1 results: element-wise MAX of the per-level maxima and OR of the flag bits
    (encoded as 4 extra doubles so a single MAX all-reduce does both).  Returns (maxlen, flags) global."""
    maxlen = np.asarray(maxlen, dtype=np.float64)
    if not is_distributed():
        return maxlen, int(flags)
    import torch
    dist = _dist()
    buf = np.concatenate([maxlen, [float(bool(flags & 1)), float(bool(flags & 2)), float(bool(flags & 4)), float(bool(flags & 8))]])
    t = torch.from_numpy(buf)
    if device is not None:
        t = t.to(device)
    dist.all_reduce(t, op=dist.ReduceOp.MAX, group=group)
    out = t.cpu().numpy()
    f = sum(bit for bit, v in zip((1, 2, 4, 8), out[-4:]) if v > 0)
    return out[:-4].copy(), int(f)


def check_partition_flags(flags):
    """The reference's failure modes on the pass-1 result (delay.py:279-283)."""
    if not (flags & FLAG_ANY_FINITE):
        raise ValueError('geo2rdr did not converge. Check orbit coverage')
    if flags & FLAG_ANY_NAN:
        raise ValueError('some ray lengths are NaN: the number of integration parts (delay.py:283) is undefined')


def check_partition(partition, max_seg=1000.0):
    """The deferred half of raytrace_slab_async: read the K+4-element device partition back (this synchronises), raise what the
    reference raises on it (delay.py:279-283 - the asynchronous path itself wrote NaN in that case) and return the slice's nParts.
    Call it whenever the host may wait: after the scene, or once per batch of scenes."""
    p = partition.detach().cpu().numpy() if hasattr(partition, 'detach') else np.asarray(partition, dtype=np.float64)
    K = p.size - 4
    if K < 0:
        raise ValueError('a partition holds K per-level maxima and 4 flag elements')
    check_partition_flags(sum(bit for bit, v in zip((1, 2, 4, 8), p[K:]) if v > 0))
    return nparts_from_maxlen(p[:K], max_seg)


def broadcast_cube_fields(fields, src=0, device=None, group=None):
    """Broadcast {xs, ys, zs, wet, hydro} (NumPy on `src`, None elsewhere) to every rank, field by field (kept for callers
    with other field sets; the cube itself goes out in one piece through broadcast_cube_packed)."""
    import torch
    dist = _dist()
    meta = [None]
    if dist.get_rank() == src:
        meta[0] = {k: (v.shape, str(v.dtype)) for k, v in fields.items()}
    dist.broadcast_object_list(meta, src=src, group=group)
    out = {}
    for k, (shape, dtype) in meta[0].items():
        if dist.get_rank() == src:
            t = torch.from_numpy(np.ascontiguousarray(fields[k]))
        else:
            t = torch.empty(shape, dtype=getattr(torch, dtype))
        if device is not None:
            t = t.to(device)
        dist.broadcast(t, src=src, group=group)
        out[k] = t
    return out


_DTYPES = {0: 'float32', 1: 'float64'}


def pack_cube(ys, xs, zs, wet, hydro, device=None):
    """ONE contiguous byte buffer holding a cube: ys | xs | zs (float64) | wet | hydro (the cube's dtype, any layout - the layout
    travels as the shape).  Returns (uint8 tensor, header) with header = (ny, nx, nz, dtype code, d0, d1, d2) describing it."""
    import torch
    wet = np.ascontiguousarray(wet); hydro = np.ascontiguousarray(hydro, dtype=wet.dtype)
    if wet.dtype not in (np.float32, np.float64) or wet.shape != hydro.shape or wet.ndim != 3:
        raise ValueError('pack_cube: wet / hydro must be float32 or float64 arrays of one 3-D shape')
    ax = np.concatenate([np.asarray(a, dtype=np.float64).ravel() for a in (ys, xs, zs)])
    buf = np.concatenate([ax.view(np.uint8), wet.reshape(-1).view(np.uint8), hydro.reshape(-1).view(np.uint8)])
    header = (np.size(ys), np.size(xs), np.size(zs), 0 if wet.dtype == np.float32 else 1) + tuple(wet.shape)
    t = torch.from_numpy(buf)
    return (t.to(device) if device is not None else t), header


def unpack_cube(buf, header):
    """Views into a packed cube buffer: (axes float64 [ny+nx+nz], wet, hydro) - no copy."""
    import torch
    ny, nx, nz, code, d0, d1, d2 = (int(v) for v in header)
    dt = getattr(torch, _DTYPES[code])
    na = (ny + nx + nz) * 8
    nf = d0 * d1 * d2 * (4 if code == 0 else 8)
    if buf.numel() != na + 2 * nf:
        raise ValueError(f'packed cube buffer holds {buf.numel()} bytes, header describes {na + 2 * nf}')
    axes = buf[:na].view(torch.float64)
    wet = buf[na:na + nf].view(dt).view(d0, d1, d2)
    hyd = buf[na + nf:].view(dt).view(d0, d1, d2)
    return axes, wet, hyd


def broadcast_cube_packed(cube_fields, src=0, device=None, group=None, header=None):
    """The weather cube to every rank in ONE broadcast (once per job, 57.6 MB for an ERA5-sized f32 cube, GPU -> GPU
    over xGMI with the nccl backend): `cube_fields` = dict(ys, xs, zs, wet, hydro) of NumPy arrays on `src`, ignored elsewhere.
    `header` = (ny, nx, nz, dtype code, d0, d1, d2) when every rank already knows the cube's shape (bench.py does); otherwise a
    7-integer header goes round first.  Returns (axes, wet, hydro) tensors on `device` (views of the one received buffer)."""
    import torch
    dist = _dist()
    rank = dist.get_rank()                  # (`src` is a global rank)
    announce = header is None              # (the same on every rank: it is an argument of the collective call)
    buf = None
    if rank == src:
        buf, hdr = pack_cube(cube_fields['ys'], cube_fields['xs'], cube_fields['zs'], cube_fields['wet'], cube_fields['hydro'], device=device)
        if header is not None and tuple(int(v) for v in header) != tuple(int(v) for v in hdr):
            raise ValueError(f'broadcast_cube_packed: header {tuple(header)} does not describe the cube {hdr}')
        header = hdr
    if announce:
        h = torch.tensor([int(v) for v in header] if rank == src else [0] * 7, dtype=torch.int64, device=device)
        dist.broadcast(h, src=src, group=group)
        header = tuple(int(v) for v in h.cpu())
    if buf is None:
        ny, nx, nz, code, d0, d1, d2 = (int(v) for v in header)
        buf = torch.empty((ny + nx + nz) * 8 + 2 * d0 * d1 * d2 * (4 if code == 0 else 8), dtype=torch.uint8, device=device)
    dist.broadcast(buf, src=src, group=group)
    return unpack_cube(buf, header)


def global_table_height(rays, ht=None, group=None, device=None):
    """The height every rank builds its level table for.  One slice: `ht` itself.  A batch with per-pixel origin heights (Rays(hts=...)):
    the lowest height of the WHOLE scene - one MIN all-reduce of a scalar over the ranks' own minima (a rank-local minimum would give the
    ranks tables of different lengths and partitions that cannot be reduced against each other).  An explicit `ht` at or below it is kept."""
    if rays.ht_min is None:
        return rays.table_height(ht)
    lo = rays.table_height(ht)
    if not is_distributed():
        return lo
    import torch
    dist = _dist()
    t = torch.tensor([lo], dtype=torch.float64, device=device if device is not None else 'cpu')
    dist.all_reduce(t, op=dist.ReduceOp.MIN, group=group)
    return float(t.item())


def raytrace_slab_async(cube, rays, ht, zref, partition, max_seg=1000.0, out=None, group=None):
    """The same slice without any host round trip (device-resident rays, outputs and partition; RCCL all-reduce on the
    K+4-element device tensor `partition`): pass 1 -> MAX all-reduce -> pass 2, all enqueued asynchronously.  The
    reference's error conditions (all-NaN slice etc.) are not raised here - the outputs are NaN instead; check_partition(partition)
    raises them afterwards, at a point where the host may wait.
    Per-pixel heights: pass the scene-wide table height (global_table_height, once per scene - it synchronises) as `ht`."""
    if ht is None and rays.ht_min is not None and is_distributed():
        raise ValueError('per-pixel heights on several ranks: pass ht = global_table_height(rays) (the scene-wide lowest height)')
    cube.ray_prepass_device(rays, ht, zref, partition)
    if is_distributed():
        dist = _dist()
        dist.all_reduce(partition, op=dist.ReduceOp.MAX, group=group)
    return cube.ray_march_device(rays, ht, zref, partition, max_seg=max_seg, out=out)


def raytrace_slab(cube, rays, ht, zref, max_seg=1000.0, out=None, group=None, device=None):
    """One slice of _build_cube_ray for THIS rank's slab of the scene, with the batch-global partition.
    Returns (wet, hydro, nparts).  Per-pixel heights (Rays(hts=...), ht=None): the level table of the whole scene starts at the lowest
    height of ALL ranks (global_table_height)."""
    if rays.ht_min is not None:
        ht = global_table_height(rays, ht, group=group, device=device)
    maxlen, flags = cube.ray_prepass(rays, ht, zref)
    maxlen, flags = reduce_partition(maxlen, flags, group=group, device=device)
    check_partition_flags(flags)
    nparts = nparts_from_maxlen(maxlen, max_seg)
    wet, hyd = cube.ray_march(rays, ht, zref, nparts, flags, out=out)
    return wet, hyd, nparts


def raytrace_heights_sharded(cube, rays_for, hts, zref, max_seg=1000.0, world=None, rank=None):
    """The OTHER way a ray-traced cube shards: by HEIGHT.  Every output height of _build_cube_ray is its own slice with its own
    level table, slice maxima and nParts (delay.py:256-323), so ranks that take disjoint blocks of heights need no collective at
    all - unlike row blocks of one slice, which share the per-level maxima (raytrace_slab*).  Rank r integrates heights
    [h0, h0 + nh) of `hts` in one batched launch pair (Cube.raytrace_slices) and keeps / writes that slab of the (nz, ny, nx) cube.
    `rays_for(heights)` returns the Rays batch for those heights (e.g. `los.ray_batch_slices(xpts, ypts, heights)`).
    Returns (h0, nh, wet, hydro, K, nparts, flags); nh == 0 when there are more ranks than heights."""
    if world is None or rank is None:
        if is_distributed():
            dist = _dist()
            world, rank = dist.get_world_size(), dist.get_rank()
        else:
            world, rank = 1, 0
    hts = np.atleast_1d(np.asarray(hts, dtype=np.float64))
    h0, nh = shard_rows(hts.size, world, rank)
    if nh == 0:
        return h0, 0, None, None, None, None, None
    mine = np.ascontiguousarray(hts[h0:h0 + nh])
    wet, hyd, K, nparts, flags = cube.raytrace_slices(rays_for(mine), mine, zref, max_seg)
    return h0, nh, wet, hyd, K, nparts, flags


def blend_on_the_fly_pays(cube, npoints):
    """The two ways to interpolate the two-epoch blend at `npoints` points of one rank: apply it at the corners (Cube.interp_blend: 8 cache
    lines = 1144 B per random point, no cube), or make the blended cube for the gather - 26 B per cell measured, the SAME on every rank,
    written pair-interleaved into scratch - and gather from it (3 lines = 463 B per point measured; profiles/r06_secondary.json).  On the
    fly pays when its extra 681 B per point stay below the blend's bytes: below ~1.9 M points on a 50 M-cell f32 cube."""
    ny, nx, nz = cube.shape
    return 681.0 * npoints < 26.0 * ny * nx * nz * (1.0 if cube.dtype == np.float32 else 2.0)


def interp_points_sharded(cube, pts, world=None, rank=None, blend=None):
    """Station / zenith queries (BASELINE configs[1], configs[4]: 5 M GNSS stations on 8 GPUs) shard with NO collective at all: every
    rank holds the (broadcast, possibly blended) cube and interpolates its contiguous block of the point list
    (delay.py:116-121 applied to rows [p0, p0 + np) of `pts[n, 3]` = (y, x, z)).  Returns (p0, np, wet, hydro); np == 0 when there are
    more ranks than points.
    blend=(w1, other, w2): the temporal interpolation w1 * cube + w2 * other (cli/raider.py:817-819) is part of the query - made as a
    cube first or applied at the corners of this rank's points, whichever moves fewer bytes (blend_on_the_fly_pays); same bits."""
    if world is None or rank is None:
        if is_distributed():
            dist = _dist()
            world, rank = dist.get_world_size(), dist.get_rank()
        else:
            world, rank = 1, 0
    n = pts.shape[0]
    p0, cnt = shard_rows(n, world, rank)
    if cnt == 0:
        return p0, 0, None, None
    mine = pts[p0:p0 + cnt]
    if blend is None:
        wet, hyd = cube.interp(mine)
    else:
        w1, other, w2 = blend
        # (the blended cube has no other reader here: it is made in the gather's own layout, in scratch - Cube.interp_blend(via_cube=True))
        wet, hyd = cube.interp_blend(w1, other, w2, mine, via_cube=not blend_on_the_fly_pays(cube, cnt))
    return p0, cnt, wet, hyd


def broadcast_and_blend(epochs, weights, src=0, device=None, group=None, header=None, ctx=None):
    """The two-epoch temporal interpolation of cli/raider.py:817-819 on every rank of a multi-GPU job: each epoch's cube goes out in one
    packed broadcast (broadcast_cube_packed), the blend w1 * a + w2 * b runs on each rank's own device (blend_kernel) - cheaper than
    blending on one rank and broadcasting, since the epochs are what the ranks need anyway when several times are interpolated.
    `epochs`: on `src` a list of dict(ys, xs, zs, wet, hydro) (file order z, y, x), elsewhere ignored (None); `weights`: one float per
    epoch, known to every rank.  Returns the blended device Cube."""
    from .engine import Cube
    cubes = []
    for k in range(len(weights)):
        f = epochs[k] if (epochs is not None and _dist().get_rank() == src) else None
        axes, wet, hyd = broadcast_cube_packed(f, src=src, device=device, group=group, header=header)
        nz, ny, nx = (int(v) for v in wet.shape)
        ax = axes.cpu().numpy()
        if device is None:
            wet, hyd = wet.numpy(), hyd.numpy()
        cubes.append(Cube(ax[:ny], ax[ny:ny + nx], ax[ny + nx:], wet, hyd, order='zyx', ctx=ctx))
    out = cubes[0]
    if len(cubes) == 1:
        return out
    if len(cubes) != 2:
        raise ValueError('broadcast_and_blend: one or two epochs (more: s1_azimuth_timing.combine_cubes)')
    return cubes[0].blend(float(weights[0]), cubes[1], float(weights[1]))
