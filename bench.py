#!/usr/bin/env python3
"""bench.py - LOS rays/sec (wet+hydro slant delay) through an ERA5-sized cube on MI355X.

Workload (BASELINE.json configs[2], the configuration the metric is quoted on): ray-traced slant delay
of a 4000x4000 SAR scene (16 M rays) as ONE slice at ht = 0 through the synthetic 300x300x80 cube of
SURVEY.md §8(d), per-pixel ECEF look vectors (an input array resident in HBM), zref = max(zs)-1,
MAX_SEGMENT_LENGTH = 1000 m.  One "step" = one pass of the hot path over that batch:
   pass 1 (build_ray fused: per-level batch max of ray length)  ->  [N>1: all-reduce MAX over ranks]
   pass 2 (trapezoid integration of wet+hydro along every ray).
N GPUs (`--scaling`): default for N > 1 is STRONG scaling on BASELINE.json configs[3] - ONE 10000x10000 scene (100 M rays)
split into contiguous row blocks (raider_amd.distributed.shard_rows), `value` = 100 M rays x steps / time; `--rows R` (or
`--scaling weak`) gives weak scaling instead: every rank traces its own R x cols slab of an (N*R) x cols scene.  Either way
the cube goes out ONCE in one packed RCCL broadcast and the only data-path collective is the MAX all-reduce of K+4 doubles
between the two passes that keeps nParts batch-global; there is no output collective.

Prints ONE JSON line on rank 0 (contract in the task statement) with `roofline` and `cpu_baseline`.
"""
import argparse
import json
import os
import sys
import time
from pathlib import Path

import numpy as np

REPO = Path(__file__).resolve().parent
sys.path.insert(0, str(REPO))

HBM_PEAK_GBS = 8000.0          # MI355X_MICROARCH.md: 8.0 TB/s HBM3E spec
# vector-ALU issue peak: 256 CUs x 4 SIMDs x 2.4 GHz / 4 cycles per wave64 instruction (fp64 FMA is full rate on CDNA4)
VALU_ISSUE_PEAK = 256 * 4 * 2.4e9 / 4
FP64_VECTOR_PEAK = 256 * 4 * 16 * 2 * 2.4e9      # flop/s: 78.6 TFLOP/s fp64 vector (MI355X_MICROARCH.md), an FMA = 2 flop per lane and 4-cycle pass
FLOP_PER_REFERENCE_SAMPLE = 110.0               # ~1.1e2 fp64 operations per (ray, sample) of the reference's algorithm


def kernel_source_hash():
    """sha256 over the HIP sources of libraider_hip.so: ties the counter digests under profiles/ to the code they measured."""
    from raider_amd import _lib
    return _lib.source_hash()


def load_counters(cube, rows, cols):
    """The newest profiles/r*_counters.json (written by tools/profile_digest.py from rocprofv3 --pmc passes) whose source hash
    equals the current kernels' and whose workload matches; (dict, path) or (None, reason).  Nothing measured is a literal
    in this file: a kernel change without a fresh profile yields nulls in the bench line."""
    cands = sorted((REPO / 'profiles').glob('r*_counters.json'), reverse=True)
    want = kernel_source_hash()
    why = 'no profiles/r*_counters.json'
    for f in cands:
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if d.get('source_hash') != want:
            why = f'{f.name}: source hash {d.get("source_hash")} != current {want} (stale profile)'
            continue
        if d.get('cube') != cube:
            why = f'{f.name}: profiled on cube {d.get("cube")}, this run uses {cube}'
            continue
        return d, str(f.relative_to(REPO))
    return None, why


def load_parity(tag):
    """The newest profiles/r*_full_scene_parity_<tag>.json (tools/full_scene_parity.py: EVERY ray of the scene against the C oracle)
    made with the CURRENT kernels; (dict, path) or (None, reason).  A record of an earlier source version is not cited."""
    cands = sorted((REPO / 'profiles').glob(f'r*_full_scene_parity_{tag}.json'), reverse=True)
    want = kernel_source_hash()
    why = f'no profiles/r*_full_scene_parity_{tag}.json'
    for f in cands:
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if d.get('source_hash') != want:
            why = f'{f.name}: source hash {d.get("source_hash")} != current {want} (stale parity record: not cited)'
            continue
        return d, str(f.relative_to(REPO))
    return None, why


def nccl_needs_devices(world, ndev):
    return (f'bench.py: --backend nccl (RCCL) needs one GPU per rank: {world} ranks asked, {ndev} device(s) visible (RCCL refuses two ranks on one device). '
            f'Run on a node with {world} GPUs, or use --backend auto: ranks that share a device then go through gloo with device-resident collective tensors '
            f'(the same pass 1 -> all-reduce -> pass 2 code path; a dry run of the plumbing, not a scaling measurement).')


def gather_rank_times(dist, dt, coll_dev, world):
    """Every rank's own time of the timed region (s), on every rank: one all-gather of a double (outside the timed region; host tensors
    unless the backend only takes device ones)."""
    import torch
    t = torch.tensor([dt], dtype=torch.float64, device=coll_dev if (coll_dev is not None and dist.get_backend() == 'nccl') else 'cpu')
    out = [torch.zeros_like(t) for _ in range(world)]
    dist.all_gather(out, t)
    return [float(o.item()) for o in out]


def device_identities(torch, dist, dist_on, backend, local, rank, world):
    """[{rank, hip_device, uuid, pci, name}] of every rank (all-gathered), so the line itself says which physical GPUs ran: under nccl
    (= RCCL, one rank per GPU) the identities must be `world` distinct devices."""
    p = torch.cuda.get_device_properties(local)
    uuid = getattr(p, 'uuid', None)
    pci = None
    if hasattr(p, 'pci_bus_id'):
        pci = f'{int(getattr(p, "pci_domain_id", 0)):04x}:{int(p.pci_bus_id):02x}:{int(p.pci_device_id):02x}'
    me = {'rank': rank, 'hip_device': local, 'uuid': str(uuid) if uuid is not None else None, 'pci': pci, 'name': p.name}
    if not dist_on:
        return [me]
    allv = [None] * world
    dist.all_gather_object(allv, me)
    ids = {(d['uuid'], d['pci']) for d in allv}
    if backend == 'nccl' and world > 1 and len(ids) != world:
        raise SystemExit(f'bench.py: {world} ranks under nccl but only {len(ids)} distinct devices: {allv}')
    return allv


def scaling_reference(kind, t1_ms, units, n, tn_ms, steps, what):
    """The one-GPU reference of an N > 1 line, measured on rank 0 of the SAME node in the SAME run while the other ranks wait at a barrier:
    strong scaling - one GPU does the whole job (efficiency t1 / (N tN)); weak - one GPU does one rank's share alone (t1 / tN)."""
    eff = (t1_ms / (n * tn_ms)) if kind == 'strong' else (t1_ms / tn_ms)
    return {'ms_per_step': t1_ms, 'value': units / (t1_ms * 1e-3), 'steps': steps, 'what': what}, eff


def oracle_block(c, xpts, ypts, inc_cols, hd, zref, nparts, r0, c0, n, hts=None, full_nparts=None):
    """One n x n block (rows r0.., columns c0..) of the scene on the C oracle (oracle/oracle_c.c) with the WHOLE slice's partition:
    (wet, hydro).  hts / full_nparts: the per-pixel-height workload (heights of the block, nParts indexed by model interval)."""
    from oracle import raider_oracle as O
    from oracle import oracle_c as OC
    xp = xpts[c0:c0 + n]; yp = ypts[r0:r0 + n]
    xx, yy = np.meshgrid(xp, yp)
    los = O.look_vectors_from_inc_hd(np.broadcast_to(inc_cols[c0:c0 + n], yy.shape), np.full(yy.shape, hd), yy, xx, 0.0)
    if hts is None:
        w, h, _ = OC.build_cube_ray_slice(c, xp, yp, 0.0, los, zref, nparts=nparts)
    else:
        w, h, _ = OC.build_cube_ray_per_pixel(c, yy, xx, hts, los, zref, nparts=full_nparts)
    return w, h


def self_launch(args):
    """`python bench.py --gpus N` invoked plainly: re-exec through torch.distributed.run with N ranks on this node."""
    import socket
    import subprocess
    if args.backend == 'nccl':          # preflight: ONE clear message here instead of N torchrun tracebacks
        import torch
        ndev = torch.cuda.device_count() if torch.cuda.is_available() else 0
        if ndev < args.gpus:
            raise SystemExit(nccl_needs_devices(args.gpus, ndev))
    with socket.socket() as s_:
        s_.bind(('127.0.0.1', 0)); port = s_.getsockname()[1]
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', f'--nproc-per-node={args.gpus}', '--master-addr', '127.0.0.1',
           '--master-port', str(port), str(Path(__file__).resolve())] + sys.argv[1:]
    env = dict(os.environ, HSA_ENABLE_IPC_MODE_LEGACY=os.environ.get('HSA_ENABLE_IPC_MODE_LEGACY', '0'))
    raise SystemExit(subprocess.call(cmd, env=env))


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=5)
    ap.add_argument('--warmup', type=int, default=2)
    ap.add_argument('--rows', type=int, default=None, help='N = 1: rows of the scene (default 4000); N > 1: rows PER RANK (implies --scaling weak)')
    ap.add_argument('--cols', type=int, default=None, help='columns of the scene (default 4000; 10000 for the strong-scaling scene)')
    ap.add_argument('--scaling', choices=('auto', 'strong', 'weak'), default='auto',
                    help='N > 1: strong = one --total-rows x cols scene (default 10000x10000 = configs[3]) sharded by rows; weak = --rows per rank. '
                         'auto = strong unless --rows is given')
    ap.add_argument('--total-rows', type=int, default=None, help='rows of the whole scene in strong scaling (default 10000)')
    ap.add_argument('--per-pixel-ht', action='store_true', help='the secondary workload "c3b" of SURVEY 8(d): every pixel starts at its own height, '
                                                                'rng(2).uniform(0, 3000) m (a scene on a DEM) instead of one slice at ht = 0; no reference '
                                                                'semantics - the rule is in include/raider_hip.h (rdr_rays.hts) and DESIGN.md 5c')
    ap.add_argument('--force-dist', action='store_true', help='N = 1: still create the process group and issue every collective of the N > 1 path '
                                                              '(one-rank RCCL group: how the nccl path is exercised on a single-GPU box)')
    ap.add_argument('--no-e2e', action='store_true', help='skip the end-to-end (H2D + kernels + D2H through the NumPy boundary) figure')
    ap.add_argument('--cube', type=str, default='300x300x80')
    ap.add_argument('--cube-f64', action='store_true', help='experiment: upload the f32 refractivities as f64 (no cvt in the gather)')
    ap.add_argument('--axes-f32', action='store_true', help='experiment: lat / lon axes rounded to float32 (not exactly uniform any more: the LDS-table cell search)')
    ap.add_argument('--coll-device', action='store_true', help='keep collective tensors on the GPU even with --backend gloo (dry run of the async path)')
    ap.add_argument('--backend', type=str, default='auto', help='torch.distributed backend for N>1: nccl (= RCCL), gloo, or auto = nccl when every rank has '
                                                                'its own GPU, else gloo with device-resident collective tensors (ranks sharing a GPU: RCCL refuses duplicates)')
    ap.add_argument('--dump', type=str, default='', help='write this rank\'s slab of the hydrostatic / wet delays to <dump>.rank<r>.npz (tests)')
    ap.add_argument('--cpu-sample', type=int, default=640, help='edge of the square ray block timed on the CPU oracle (0 = skip)')
    ap.add_argument('--workload', choices=('rays', 'c2', 'c5'), default='rays',
                    help='rays (default): the ray-traced scene the metric is quoted on (configs[2] / configs[3]); its line also carries `secondary`: the kernel-level '
                         'figures + parity errors of c2 and c5 (--no-secondary skips them).  c2: BASELINE configs[1] - Conventional slant (inc 39 deg), --points x --points '
                         'query points with their own heights through the point branch (intermediate delay cube on the AOI grid + one gather + / cos(inc)) on the '
                         'ERA5-sized f64 totals cube; value = points/s.  c5: BASELINE configs[4] - two HRRR-sized '
                         '1000x1000x50 f32 epochs on the 3-km LCC grid, blended (0.25, 0.75), --stations GNSS station points gathered from the blend; '
                         'N > 1: the epochs go out in two packed broadcasts, every rank blends and gathers its block of the stations (no data-path collective); '
                         'value = stations/s')
    ap.add_argument('--stations', type=int, default=5_000_000, help='--workload c5: station points of the whole job')
    ap.add_argument('--points', type=int, default=1000, help='--workload c2: edge of the square point set (default 1000 x 1000)')
    ap.add_argument('--no-secondary', action='store_true', help='default workload: skip the `secondary` c2 / c5 / real-levels figures')
    ap.add_argument('--parity-block', type=int, default=256, help='N > 1: edge of the square block of EVERY rank\'s slab compared with the C oracle after the timed '
                                                                   'region (`parity_sample`, max over ranks; c2 / c5: that many x 64 points of the rank\'s block); 0 = skip')
    ap.add_argument('--no-one-gpu-ref', action='store_true', help='N > 1: skip the one-GPU reference run of the same job on rank 0 (`one_gpu_same_scene`, `scaling_efficiency`)')
    args = ap.parse_args()
    if args.scaling == 'auto':
        args.scaling = 'strong' if (args.gpus > 1 and args.rows is None) else 'weak'
    if args.scaling == 'strong' and args.rows is not None:
        raise SystemExit('bench.py: --rows is the per-rank slab of weak scaling; the strong-scaling scene is --total-rows x --cols')
    if args.gpus > 1 and 'WORLD_SIZE' not in os.environ:
        self_launch(args)

    os.environ.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')       # (before the HSA runtime starts: this pool's driver only has dmabuf IPC, RCCL needs it)
    # stdout carries ONE line, the result: whatever libraries print there (RCCL's version banner goes to the C stdout and comes out
    # when the process exits, i.e. AFTER a Python print) is sent to stderr; the JSON line is written to the saved descriptor
    sys.stdout.flush()
    result_fd = os.dup(1)
    os.dup2(2, 1)
    import torch
    import torch.distributed as dist
    import raider_amd as R
    from raider_amd.synthetic import synthetic_cube, scene_grid
    from raider_amd import distributed as D

    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        raise SystemExit(f'bench.py: --gpus {args.gpus} but WORLD_SIZE={world}')
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X (no CPU fallback)')
    ndev = torch.cuda.device_count()
    local = local % ndev
    torch.cuda.set_device(local)
    dev = torch.device('cuda', local)
    if args.backend == 'auto':
        args.backend = 'nccl' if world <= ndev else 'gloo'
        if args.backend == 'gloo':
            args.coll_device = True
    elif args.backend == 'nccl' and world > ndev:      # (ranks started by a launcher other than self_launch: the same message, from rank 0 only)
        if rank == 0:
            sys.stderr.write(nccl_needs_devices(world, ndev) + '\n')
        raise SystemExit(2)
    dist_on = world > 1 or args.force_dist
    if dist_on:
        os.environ.setdefault('MASTER_ADDR', '127.0.0.1')
        if 'MASTER_PORT' not in os.environ:
            import socket
            with socket.socket() as s_:
                s_.bind(('127.0.0.1', 0)); os.environ['MASTER_PORT'] = str(s_.getsockname()[1])
        if args.backend == 'nccl':
            dist.init_process_group('nccl', rank=rank, world_size=world, device_id=dev)
        else:
            dist.init_process_group(args.backend, rank=rank, world_size=world)
    coll_dev = dev if (args.backend == 'nccl' or args.coll_device) else None      # where the collectives' tensors live

    ctx = R.Context(local)
    # (raider_amd launches on torch's current stream whenever it is handed device tensors)
    if args.workload == 'c5':
        return run_c5(args, ctx, dev, coll_dev, dist_on, rank, world, ndev, result_fd)
    if args.workload == 'c2':
        return run_c2(args, ctx, dev, coll_dev, dist_on, rank, world, ndev, result_fd)

    # ---- weather cube: generated on rank 0, sent to every rank in ONE packed broadcast (RCCL over xGMI), packed (y,x,z) on device
    ny, nx, nz = (int(v) for v in args.cube.split('x'))
    fields = None
    if rank == 0:
        c = synthetic_cube(ny, nx, nz, seed=0)
        fields = {k: c[k] for k in ('ys', 'xs', 'zs', 'wet', 'hydro')}
    t_bcast = 0.0
    if dist_on:
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        # every rank knows the cube's shape from --cube: no header round, one collective
        axes, wet, hyd = D.broadcast_cube_packed(fields, src=0, device=coll_dev, header=(ny, nx, nz, 0, nz, ny, nx))
        if coll_dev is None:      # gloo dry run on host tensors
            axes, wet, hyd = axes.to(dev), wet.to(dev), hyd.to(dev)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        ax = axes.cpu().numpy()
    else:
        ax = np.concatenate([c['ys'], c['xs'], c['zs']])
        wet = torch.from_numpy(c['wet']).to(dev); hyd = torch.from_numpy(c['hydro']).to(dev)
    ys, xs, zs = ax[:ny], ax[ny:ny + nx], ax[ny + nx:]
    if args.cube_f64:
        wet, hyd = wet.double(), hyd.double()
    if args.axes_f32:
        ys, xs = ys.astype(np.float32).astype(np.float64), xs.astype(np.float32).astype(np.float64)
    cube = R.Cube(ys, xs, zs, wet, hyd, order='zyx', ctx=ctx)
    zref = float(zs.max() - 1.0)                                 # delay.py:78,86-87
    ht = 0.0

    # ---- this rank's slab of the scene; look vectors generated on device, then used as an INPUT array ---
    if args.scaling == 'strong':
        total_rows = args.total_rows or 10000
        cols = args.cols or 10000
        row0, rows = D.shard_rows(total_rows, world, rank)
    else:
        rows = args.rows or 4000
        cols = args.cols or 4000
        total_rows, row0 = rows * world, rank * rows
    xpts, ypts, inc_cols, hd = scene_grid(rows, cols, row0=row0, nrows=rows, total_rows=total_rows)
    xpts_t = torch.from_numpy(xpts).to(dev); ypts_t = torch.from_numpy(ypts).to(dev)
    inc_t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
    hd_t = torch.full((rows, cols), hd, dtype=torch.float64, device=dev)
    los_t = R.Rays.grid(xpts_t, ypts_t, inc=inc_t, hd=hd_t).look_vectors(ctx)       # (rows, cols, 3) f64 in HBM
    del inc_t, hd_t
    hts_t = None
    if args.per_pixel_ht:      # drawn for the WHOLE scene, this rank keeps its rows
        hts_np = np.random.default_rng(2).uniform(0.0, 3000.0, (total_rows, cols))[row0:row0 + rows]
        hts_t = torch.from_numpy(np.ascontiguousarray(hts_np)).to(dev)
        ht = None
    rays = R.Rays.grid(xpts_t, ypts_t, los=los_t, hts=hts_t)
    if hts_t is not None and dist_on:       # the level table of the whole scene starts at the lowest pixel of ALL ranks
        ht = D.global_table_height(rays, None, device=coll_dev)
    out_w = torch.empty((rows, cols), dtype=torch.float64, device=dev)
    out_h = torch.empty_like(out_w)
    n_rays = rows * cols

    partition = None      # device-resident pass-1 result (K+4 doubles) the RCCL all-reduce works on

    def step():
        if not dist_on:
            cube.raytrace(rays, ht, zref, out=(out_w, out_h), want_nparts=False)       # fully asynchronous
        elif partition is not None:                                                    # pass 1 -> RCCL MAX all-reduce (K+4 doubles, on the device) -> pass 2
            D.raytrace_slab_async(cube, rays, ht, zref, partition, out=(out_w, out_h))
        else:                                                                          # gloo dry run: the partition travels through the host
            D.raytrace_slab(cube, rays, ht, zref, out=(out_w, out_h), device=coll_dev)

    # nParts / S for the roofline formula (one synchronous untimed call, which also raises the reference's error conditions)
    if not dist_on:
        _, _, nparts, flags = cube.raytrace(rays, ht, zref, out=(out_w, out_h), want_nparts=True)
    else:
        _, _, nparts = D.raytrace_slab(cube, rays, ht, zref, out=(out_w, out_h), device=coll_dev)
    S = int(np.sum(nparts)); K = int(len(nparts))
    if dist_on and coll_dev is not None:
        partition = torch.zeros(K + 4, dtype=torch.float64, device=dev)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    # the shader clock DURING the timed steps (one sleeping wave on another stream reads the cycle and the wall counter, rdr_clock_sample_*):
    # sized from one untimed step so that it ends inside the timed region
    torch.cuda.synchronize()
    tw = time.perf_counter(); step(); torch.cuda.synchronize(); tw = time.perf_counter() - tw
    if dist_on:
        dist.barrier()
    ctx.set_profiling(True)                                      # HIP event pairs around every kernel launch
    torch.cuda.synchronize()
    clock_end = None
    if rank == 0:
        try:
            clock_end = ctx.clock_sample(max(0.2, 0.8 * tw * args.steps * 1e3))
        except (RuntimeError, ValueError):       # (a diagnostic: the line is still valid without it - clock_GHz_measured: null)
            clock_end = None
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    try:
        clock_ghz = clock_end() if clock_end is not None else None
    except (RuntimeError, ValueError):
        clock_ghz = None
    n_pre, ms_pre = ctx.profile_get(0)
    n_march, ms_march = ctx.profile_get(1)
    ctx.set_profiling(False)
    rank_s = [dt]
    if dist_on:
        rank_s = gather_rank_times(dist, dt, coll_dev, world)
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev is not None else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())

    if args.dump:
        np.savez(f'{args.dump}.rank{rank}.npz', wet=out_w.cpu().numpy(), hydro=out_h.cpu().numpy(), nparts=np.asarray(nparts), row0=row0, rows=rows)
    nan_frac = float(torch.isnan(out_h).double().mean().item())
    mean_h = float(torch.nanmean(out_h).item()); mean_w = float(torch.nanmean(out_w).item())

    # ---- N > 1: what makes the line readable on its own (VERDICT r5 task 2) ---------------------------------------------------------------
    devices = device_identities(torch, dist, dist_on, args.backend, local, rank, world)
    parity_sample = None
    if world > 1 and args.parity_block > 0:
        # every rank: one block at the centre of ITS slab against the C oracle driven with the all-reduced (whole-scene) partition
        min_rows = min(D.shard_rows(total_rows, world, r_)[1] for r_ in range(world)) if args.scaling == 'strong' else rows
        nb = int(min(args.parity_block, min_rows, cols))                             # the same block size on every rank
        r0b, c0b = (rows - nb) // 2, (cols - nb) // 2
        cc = synthetic_cube(ny, nx, nz, seed=0)
        if args.per_pixel_ht:
            fullnp = np.zeros(nz - 1, dtype=np.int32); fullnp[cube.ray_levels(ht, zref)[2]] = nparts
            ow_, oh_ = oracle_block(cc, xpts, ypts, inc_cols, hd, zref, nparts, r0b, c0b, nb, hts=hts_np[r0b:r0b + nb, c0b:c0b + nb], full_nparts=fullnp)
        else:
            ow_, oh_ = oracle_block(cc, xpts, ypts, inc_cols, hd, zref, nparts, r0b, c0b, nb)
        gw_ = out_w[r0b:r0b + nb, c0b:c0b + nb].cpu().numpy(); gh_ = out_h[r0b:r0b + nb, c0b:c0b + nb].cpu().numpy()
        mism = int((np.isnan(gw_) != np.isnan(ow_)).sum() + (np.isnan(gh_) != np.isnan(oh_)).sum())
        mine_ = [float(np.nanmax(np.abs(gw_ - ow_))), float(np.nanmax(np.abs(gh_ - oh_))), float(mism)]
        allp = [None] * world
        dist.all_gather_object(allp, mine_)
        parity_sample = {'block': [nb, nb], 'where': 'centre of every rank\'s slab', 'rays_compared_all_ranks': nb * nb * world,
                         'max_abs_wet_m': max(a_[0] for a_ in allp), 'max_abs_hydro_m': max(a_[1] for a_ in allp), 'nan_mask_mismatches': int(sum(a_[2] for a_ in allp)),
                         'per_rank_max_abs_m': [max(a_[0], a_[1]) for a_ in allp], 'tolerance_m': 1e-6,
                         'what': 'GPU slab vs the C oracle (oracle/oracle_c.c) driven with the all-reduced whole-scene partition; max over ranks'}
        del cc
    one_gpu = None; eff = None
    if world > 1 and not args.no_one_gpu_ref:
        # rank 0 ALONE (the others wait at the barrier) traces the job one GPU would have: the whole scene (strong) or one slab (weak)
        torch.cuda.synchronize(); dist.barrier()
        if rank == 0:
            if args.scaling == 'strong':
                xa, ya, inca, _ = scene_grid(total_rows, cols)
                xa_t = torch.from_numpy(xa).to(dev); ya_t = torch.from_numpy(ya).to(dev)
                inc_a = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inca, (total_rows, cols)))).to(dev)
                hd_a = torch.full((total_rows, cols), hd, dtype=torch.float64, device=dev)
                los_a = R.Rays.grid(xa_t, ya_t, inc=inc_a, hd=hd_a).look_vectors(ctx)
                del inc_a, hd_a
                hts_a = None
                if args.per_pixel_ht:
                    hts_a = torch.from_numpy(np.ascontiguousarray(np.random.default_rng(2).uniform(0.0, 3000.0, (total_rows, cols)))).to(dev)
                rays_a = R.Rays.grid(xa_t, ya_t, los=los_a, hts=hts_a)
                ref_rows = total_rows
            else:
                rays_a, ref_rows = rays, rows
            ow1 = torch.empty((ref_rows, cols), dtype=torch.float64, device=dev); oh1 = torch.empty_like(ow1)
            k1 = max(1, min(args.steps, 3))
            cube.raytrace(rays_a, ht, zref, out=(ow1, oh1), want_nparts=False)            # warm-up (workspace of the larger batch)
            torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k1):
                cube.raytrace(rays_a, ht, zref, out=(ow1, oh1), want_nparts=False)
            torch.cuda.synchronize()
            t1 = (time.perf_counter() - t1) / k1 * 1e3
            which = 'the WHOLE' if args.scaling == 'strong' else 'one slab of the'
            one_gpu, eff = scaling_reference(args.scaling, t1, ref_rows * cols, world, dt / args.steps * 1e3, k1,
                                             f'rank 0 alone, other ranks idle at a barrier: {which} scene = {ref_rows}x{cols} rays, cube.raytrace (no collective), '
                                             f'after the timed region of the same run')
            if args.scaling == 'strong':
                one_gpu['same_bits_as_sharded_run'] = bool(torch.equal(oh1[row0:row0 + rows], out_h) and torch.equal(ow1[row0:row0 + rows], out_w))
            del ow1, oh1, rays_a
        dist.barrier()

    # ---- end to end through the NumPy boundary (reported separately, never `value`): the same scene handed over as
    # HOST arrays - look vectors up (24 B/ray), both passes, both delays down (16 B/ray); the library pipelines the transfers
    e2e = None
    if world == 1 and not args.no_e2e:
        los_np = los_t.cpu().numpy()
        hts_h = hts_np if args.per_pixel_ht else None
        eo = (np.empty((rows, cols)), np.empty((rows, cols)))
        cube.raytrace(R.Rays.grid(xpts, ypts, los=los_np, hts=hts_h), ht, zref, out=eo, want_nparts=False)      # warm-up (stages, page faults)
        reps = 3
        t0 = time.perf_counter()
        for _ in range(reps):
            cube.raytrace(R.Rays.grid(xpts, ypts, los=los_np, hts=hts_h), ht, zref, out=eo, want_nparts=False)
        dte = (time.perf_counter() - t0) / reps
        same = bool(np.array_equal(eo[1], out_h.cpu().numpy(), equal_nan=True))
        e2e = {'value': n_rays / dte, 'unit': 'rays/s', 'ms': dte * 1e3, 'h2d_bytes': n_rays * (32 if args.per_pixel_ht else 24), 'd2h_bytes': n_rays * 16,
               'what': 'same scene through the NumPy (host-buffer) boundary: look-vector upload + pass 1 + pass 2 + download, pageable host '
                       'memory, transfers overlapped with the kernels in 8 row chunks', 'bit_identical_to_device_path': same}
        del los_np, eo

    if rank == 0:
        scene_rays = total_rows * cols                            # all ranks together
        total_rays = scene_rays * args.steps
        value = total_rays / dt
        bytes_per_ray = 64 * S + 64                               # SURVEY.md §8(d) gather model
        # per-STEP kernel time (a step may launch a kernel several times when the batch is marched in chunks)
        march_ms = ms_march / args.steps
        pre_ms = ms_pre / args.steps
        gather_GBps = bytes_per_ray * n_rays / (march_ms * 1e-3) / 1e9
        # SQ / HBM counters of THIS source version on THIS workload, from the tracked digest tools/profile_digest.py wrote
        prof, prof_src = load_counters(args.cube, rows, cols)
        wl_ok = prof is not None and prof.get('cube') == args.cube and not args.per_pixel_ht
        km = (prof or {}).get('kernels', {}).get('march_kernel', {}) if wl_ok else {}
        kc = (prof or {}).get('kernels', {}).get('crossings_kernel', {}) if wl_ok else {}
        valu_rw = km.get('valu_per_raywave')                      # per 64-ray wave: independent of the number of rays
        scene_ok = wl_ok and prof.get('hbm_scene') == [rows, cols] and world == 1
        traffic = (km.get('hbm_read_bytes', 0) + km.get('hbm_write_bytes', 0)) if (scene_ok and 'hbm_read_bytes' in km) else None
        step_traffic = (traffic + kc.get('hbm_read_bytes', 0) + kc.get('hbm_write_bytes', 0)) if (traffic is not None and 'hbm_read_bytes' in kc) else None
        valu_rate = (valu_rw * (n_rays / 64.0) / (march_ms * 1e-3)) if valu_rw else None
        compulsory = 64 + (ny * nx * nz * 8) / n_rays             # B/ray: look vector in, two delays out, the cube once
        pr_ = 2 if args.per_pixel_ht else 0
        ka_m, ka_c = cube.ray_kernel_attributes(1 + pr_), cube.ray_kernel_attributes(0 + pr_)      # from the loaded code object
        frac_valu = valu_rate / VALU_ISSUE_PEAK if valu_rate else None
        # MEASURED instruction mix of the march kernel (rocprofv3 SQ class counters, profiles/r*_instr_mix.json through the same digest): what
        # the issued instructions were, the fp64 flops they executed, and the issue cycles they cost at the per-class rates
        mix = km.get('instr_mix')
        waves = n_rays / 64.0
        executed_fp64 = (mix['fp64_flops_per_lane'] * 64.0 * waves / (march_ms * 1e-3)) if mix else None
        class_cycles = (mix['issue_cycles_model'] * waves / (march_ms * 1e-3)) if mix else None            # SIMD cycles per second the mix needs
        simd_cycles = 256 * 4 * 2.4e9
        # how GOOD the kernel is, beside how busy: the reference's arithmetic (SURVEY 8d flop model: 110 fp64 flop per ray and
        # REFERENCE sample, S of them per ray) per second of march time against the fp64 vector peak - a kernel that issues more
        # instructions for the same rays scores lower here and the same in `frac`
        useful_flops = FLOP_PER_REFERENCE_SAMPLE * S * n_rays / (march_ms * 1e-3)
        parity, parity_src = load_parity('c3b' if args.per_pixel_ht else ('c3' if (rows, cols) == (4000, 4000) else 'c4' if (total_rows, cols) == (10000, 10000) else 'none'))
        frac_hbm = (traffic / (march_ms * 1e-3) / 1e9 / HBM_PEAK_GBS) if traffic is not None else None
        if args.per_pixel_ht:
            wl = (f'c3b (SURVEY 8d secondary workload, no reference semantics): Raytracing LOS, {total_rows}x{cols} scene on a DEM - per-pixel origin heights '
                  f'rng(2).uniform(0, 3000) m - per-pixel ECEF look vectors, synthetic ERA5-sized {args.cube} f32 cube, zref=max(z)-1, MAX_SEGMENT_LENGTH=1000; '
                  f'level table from the lowest pixel, nParts from the per-level maximum over the rays that reach the level')
        elif world == 1:
            wl = (f'configs[2]: Raytracing LOS, {rows}x{cols} scene ({n_rays/1e6:.1f}M rays), one slice at ht=0, per-pixel ECEF look vectors, '
                  f'synthetic ERA5-sized {args.cube} f32 cube, zref=max(z)-1, MAX_SEGMENT_LENGTH=1000')
        elif args.scaling == 'strong':
            wl = (f'configs[3]: Raytracing LOS, ONE {total_rows}x{cols} scene ({scene_rays/1e6:.0f}M rays) sharded into {world} contiguous row blocks, '
                  f'one slice at ht=0, per-pixel ECEF look vectors, synthetic ERA5-sized {args.cube} f32 cube broadcast over RCCL, zref=max(z)-1, '
                  f'MAX_SEGMENT_LENGTH=1000')
        else:
            wl = (f'configs[2] per GPU (weak scaling): Raytracing LOS, {rows}x{cols} slab per rank of a {total_rows}x{cols} scene, one slice at ht=0, '
                  f'per-pixel ECEF look vectors, synthetic ERA5-sized {args.cube} f32 cube broadcast over RCCL, zref=max(z)-1, MAX_SEGMENT_LENGTH=1000')
        res = {
            'metric': 'LOS rays/sec (wet+hydro slant delay) through ERA5 cube; achieved HBM GB/s',
            'value': value, 'unit': 'rays/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup,
            'ms_per_step': dt / args.steps * 1e3, 'higher_is_better': True, 'scaling': args.scaling if world > 1 else 'weak', 'vs_baseline': None,
            'dtype': 'f64', 'data': 'synthetic',
            'config': {'workload': wl,
                       'rays_per_gpu': n_rays, 'rays_per_step_all_gpus': scene_rays, 'cube': args.cube, 'levels_K': K, 'samples_per_ray_S': S,
                       'parallelism': (f'rows sharded x{world} ({args.backend}, {ndev} device(s) visible), cube: one packed broadcast ({t_bcast*1e3:.1f} ms), '
                                       f'MAX all-reduce of {K}+4 doubles per step, no output collective') if dist_on else 'single GPU',
                       'ranks': world, 'backend': (dist.get_backend() if dist_on else None),
                       'world_size_seen_by_backend': (dist.get_world_size() if dist_on else 1),
                       # every rank's row block [row0, rows] of the scene and its own time per step (the line's ms_per_step is their maximum)
                       'shards': ([list(D.shard_rows(total_rows, world, r_)) for r_ in range(world)] if args.scaling == 'strong' else
                                  [[r_ * rows, rows] for r_ in range(world)]),
                       'rank_ms_per_step': [t_ / args.steps * 1e3 for t_ in rank_s],
                       'devices_visible': ndev, 'ranks_per_device': -(-world // ndev),
                       'mean_hydro_m': mean_h, 'mean_wet_m': mean_w, 'nan_fraction': nan_frac},
            # The limiter the SQ counters show is fp64 vector-ALU issue, so THAT is the roofline (frac <= 1 by construction:
            # instructions actually issued / issue slots of the chip).  north_star's ">= 60 % of HBM" is not meetable at S = 178:
            # the gathers never reach HBM.  Both fractions are at the top level of this object.
            'roofline': {'bound': 'valu_fp64_issue', 'kernel': 'march_kernel<float2,false,1,true>' if args.per_pixel_ht else 'march_kernel<float2,false,1>',
                         'achieved': valu_rate / 1e9 if valu_rate else None, 'peak': VALU_ISSUE_PEAK / 1e9, 'unit': 'G wave64-instr/s',
                         'frac': frac_valu, 'frac_valu': frac_valu, 'frac_hbm_measured': frac_hbm,
                         # frac prices every VALU instruction at the fp64 rate and the 2.4 GHz data-sheet clock: it says how BUSY the issue
                         # ports are.  useful_flops_frac (the headline for kernel quality) says how much of the chip's fp64
                         # vector peak goes into the reference's own arithmetic; the clock the chip really ran at is beside it.
                         # three different things: `frac` = how BUSY the issue ports are (every instruction priced at 4 cycles);
                         # `useful_flops_frac` = the REFERENCE's arithmetic per second against the fp64 peak (a throughput-equivalent: flops the kernel
                         # avoids still count); `executed_fp64_flops_frac` = the fp64 flops the kernel actually executed (counted by the SQ) against that
                         # peak = its EFFICIENCY as an fp64 machine; `frac_class_priced` = issue cycles of the measured mix at the per-class rates
                         # (fp64 + conversions 4 cycles, f32 / integer 2) over the cycles the chip had at 2.4 GHz
                         'executed_fp64_flops_frac': (executed_fp64 / FP64_VECTOR_PEAK) if executed_fp64 else None,
                         'executed_fp64_TFLOPs': (executed_fp64 / 1e12) if executed_fp64 else None,
                         'frac_class_priced': (class_cycles / simd_cycles) if class_cycles else None,
                         'valu_mix': ({k: mix[k] / mix['valu'] for k in ('fp64', 'cvt', 'int32', 'int64', 'f32', 'other')} if mix else None),
                         'valu_mix_per_raywave': mix,
                         'valu_mix_source': 'rocprofv3 --pmc SQ_INSTS_VALU_{FMA,ADD,MUL,TRANS}_F64 / _CVT / _INT32 / _INT64 / _{FMA,ADD,MUL,TRANS}_F32 (tools/profile_round.sh sq3-sq6)' if mix else None,
                         'useful_flops_frac': useful_flops / FP64_VECTOR_PEAK, 'useful_TFLOPs': useful_flops / 1e12, 'fp64_vector_peak_TFLOPs': FP64_VECTOR_PEAK / 1e12,
                         'flop_model': f'{FLOP_PER_REFERENCE_SAMPLE:.0f} fp64 flop x S = {S} reference samples per ray (SURVEY 8d)',
                         'valu_per_reference_sample': (valu_rw / S) if valu_rw else None,
                         'clock_GHz_measured': clock_ghz, 'clock_GHz_assumed_by_peak': 2.4,
                         'frac_at_measured_clock': (valu_rate / (256 * 4 * clock_ghz * 1e9 / 4)) if (valu_rate and clock_ghz) else None,
                         'useful_flops_frac_at_measured_clock': (useful_flops / (FP64_VECTOR_PEAK * clock_ghz / 2.4)) if clock_ghz else None,
                         'traffic': traffic, 'traffic_unit': 'HBM bytes per march_kernel launch (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes)',
                         'traffic_over_compulsory': (step_traffic / (compulsory * n_rays)) if step_traffic is not None else None,
                         'valu_instr_per_raywave': valu_rw, 'valu_busy_frac': km.get('valu_busy_frac'),
                         'valu_per_evaluated_sample': (valu_rw / (S - (K - 1))) if valu_rw else None,
                         # SURVEY 8(d)'s models over the WHOLE driver-timed step (both kernels are needed to produce a ray), beside the busy-ness figure:
                         # `frac` says how busy the issue ports are while march_kernel runs, NOT how close the step is to a hard ceiling
                         'frac_is': 'valu_issue_busy',
                         'survey_flops_frac_step': FLOP_PER_REFERENCE_SAMPLE * S * n_rays / (dt / args.steps) / FP64_VECTOR_PEAK,
                         'survey_bytes_over_hbm_peak_step': bytes_per_ray * n_rays / (dt / args.steps) / (HBM_PEAK_GBS * 1e9),
                         'survey_bytes_note': 'SURVEY 8(d) gather bytes (64 S + 64 per ray) per second of step time over the HBM peak: above 1 because the gathers are '
                                              'cache-served (L1 / L2 / Infinity Cache) - a model rate, not a utilisation; the measured HBM fraction is hbm.step_measured_frac',
                         'dependent_chain_note': 'figures of merit are VALU instructions and dependent-chain length PER EVALUATED SAMPLE (one sample in flight per lane, 4 waves per '
                                                 'SIMD): round-5/6 A/Bs - 4.2 % fewer VALU bought 1.3 % (pk-diff), 6.3 % fewer (32-bit bookkeeping) bought 1-4 % box to box (LDS level records), 8.6 % fewer through LDS tiles LOST 7 % - the kernel '
                                                 'is co-limited by the one-sample chain, so only fewer fp64-rate operations per sample or a shorter chain move it',
                         'vgpr': ka_m['vgpr'], 'lds_bytes': ka_m['lds_static'] + ka_m['lds_dynamic'], 'scratch_bytes': ka_m['scratch'],
                         'resources_source': 'hipFuncGetAttributes on the loaded code object + the launch\'s dynamic LDS size',
                         'crossings': {'valu_instr_per_raywave': kc.get('valu_per_raywave'), 'valu_busy_frac': kc.get('valu_busy_frac'),
                                       'vgpr': ka_c['vgpr'], 'lds_bytes': ka_c['lds_static'] + ka_c['lds_dynamic'], 'scratch_bytes': ka_c['scratch'],
                                       'frac': (kc['valu_per_raywave'] * (n_rays / 64.0) / (pre_ms * 1e-3) / VALU_ISSUE_PEAK) if kc.get('valu_per_raywave') and pre_ms > 0 else None},
                         'march_ms_per_step': march_ms, 'crossings_ms_per_step': pre_ms, 'march_launches_timed': n_march,
                         'counters_source': prof_src if prof is not None else None, 'counters_note': None if prof is not None else prof_src,
                         'source_hash': kernel_source_hash(), 'library_source_hash': R.load_library().rdr_source_hash().decode(),
                         'hbm': {'peak_GBps': HBM_PEAK_GBS,
                                 'algorithmic_bytes_per_ray': bytes_per_ray, 'compulsory_bytes_per_ray': compulsory,
                                 'measured_GBps': (traffic / (march_ms * 1e-3) / 1e9) if traffic is not None else None,
                                 'hbm_measured_frac': frac_hbm,
                                 'step_traffic_bytes': step_traffic,
                                 'step_measured_frac': (step_traffic / ((march_ms + pre_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS) if step_traffic is not None else None,
                                 'gather_model_GBps': gather_GBps,
                                 'gather_model_note': '(64*S+64) B/ray of SURVEY 8d x rays / march time: an ALGORITHMIC rate above the HBM peak because the '
                                                      'gathers are L2/MALL hits - not a utilisation'}},
        }
        res['parity'] = ({'record': parity_src, 'rays_compared': parity.get('rays'), 'max_abs_wet_m': parity.get('max_abs_wet_m'),
                          'max_abs_hydro_m': parity.get('max_abs_hydro_m'), 'nan_mask_mismatches': parity.get('nan_mask_mismatches'),
                          'nparts_equal': parity.get('nparts_equal'), 'tolerance_m': parity.get('tolerance_m'), 'source_hash': parity.get('source_hash'),
                          'what': 'every ray of this scene against the C oracle (tools/full_scene_parity.py), made with the kernels of this source hash'}
                         if parity is not None else {'record': None, 'note': parity_src})
        res['config']['devices'] = devices
        res['config']['distinct_devices'] = len({(d_['uuid'], d_['pci']) for d_ in devices})
        if parity_sample is not None:
            res['parity_sample'] = parity_sample
        if one_gpu is not None:
            res['one_gpu_same_scene'] = one_gpu
            res['scaling_efficiency'] = eff
            res['scaling_efficiency_is'] = ('strong: t(1 GPU, whole scene) / (N x t(N GPUs))' if args.scaling == 'strong' else 'weak: t(1 GPU, one slab alone) / t(N GPUs, one slab each)')
            if args.scaling == 'strong':
                res['strong_scaling_efficiency'] = eff
        if e2e is not None:
            res['end_to_end'] = e2e
        if world == 1 and args.cpu_sample > 0:
            pp = (hts_np, cube.ray_levels(rays.ht_min, zref)[2]) if args.per_pixel_ht else None
            res['cpu_baseline'] = cpu_baseline(args, rows, cols, xpts, ypts, inc_cols, hd, nparts, zref, out_w, out_h, pp)
        if world == 1 and not args.no_secondary and not args.per_pixel_ht:
            # the two gather workloads of BASELINE.json beside the headline (a few ms of GPU time each; their own lines: --workload c2 / c5)
            del los_t, out_w, out_h, rays, cube, wet, hyd
            torch.cuda.empty_cache()
            sec = {}
            for name, fn in (('c2', lambda: c2_measure(ctx, dev, n_side=1000, steps=5, warmup=2, oracle_sample=20000)),
                             ('c5', lambda: c5_measure(ctx, dev, stations=5_000_000, steps=5, warmup=2, oracle_sample=20000))):
                try:
                    sec[name] = compact_secondary(fn())
                except Exception as exc:          # (a diagnostic appendix: the headline line stays valid without it)
                    sec[name] = {'error': f'{type(exc).__name__}: {exc}'}
            # the same scene on the REAL level heights a RAiDER user runs (ERA5: 145 levels to 80.3 km; HRRR: 57 to 26.2 km), synthetic fields
            rl = {}
            for tag, model in (('era5_145', 'era5'), ('hrrr_57', 'hrrr')):
                try:
                    rl[tag] = real_levels_measure(ctx, dev, model, rows, cols, steps=3, block=(256 if args.cpu_sample > 0 else 0))
                except Exception as exc:
                    rl[tag] = {'error': f'{type(exc).__name__}: {exc}'}
                torch.cuda.empty_cache()
            sec['real_levels'] = rl
            res['secondary'] = sec
        os.write(result_fd, (json.dumps(res) + '\n').encode())
    if dist_on:
        dist.destroy_process_group()


def real_levels_measure(ctx, dev, model, rows, cols, steps=3, block=256):
    """The headline scene (rows x cols rays, per-pixel ECEF look vectors, one slice at ht = 0) through a cube on a model's REAL level heights
    (raider_amd.synthetic.real_level_heights: ERA5's 145 levels to 80.3 km, HRRR's 57 to 26.2 km; models/model_levels.py) with the
    synthetic fields of SURVEY 8(d): the quadratic 80-level axis of the headline has 74 % level tops among its evaluated samples, real axes
    have other mixes.  Returns rays/s, S, K, the kernels' HIP-event times, time per evaluated sample and a block against the C oracle."""
    import torch
    import raider_amd as R
    from raider_amd.synthetic import synthetic_cube, scene_grid, real_level_heights
    zs = real_level_heights(model)
    c = synthetic_cube(300, 300, zs.size, seed=0, zs=zs)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    cube = R.Cube(c['ys'], c['xs'], c['zs'], torch.from_numpy(c['wet']).to(dev), torch.from_numpy(c['hydro']).to(dev), order='zyx', ctx=ctx)
    zref = float(zs.max() - 1.0)
    xpts, ypts, inc_cols, hd = scene_grid(rows, cols)
    xt, yt = torch.from_numpy(xpts).to(dev), torch.from_numpy(ypts).to(dev)
    inc_t = torch.from_numpy(np.ascontiguousarray(np.broadcast_to(inc_cols, (rows, cols)))).to(dev)
    hd_t = torch.full((rows, cols), hd, dtype=torch.float64, device=dev)
    los_t = R.Rays.grid(xt, yt, inc=inc_t, hd=hd_t).look_vectors(ctx)
    del inc_t, hd_t
    rays = R.Rays.grid(xt, yt, los=los_t)
    ow = torch.empty((rows, cols), dtype=torch.float64, device=dev); oh = torch.empty_like(ow)
    _, _, nparts, _ = cube.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=True)
    S = int(np.sum(nparts)); K = int(len(nparts))
    cube.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=False)
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        cube.raytrace(rays, 0.0, zref, out=(ow, oh), want_nparts=False)
    torch.cuda.synchronize()
    dt = (time.perf_counter() - t0) / steps
    _, ms_pre = ctx.profile_get(0)
    _, ms_march = ctx.profile_get(1)
    ctx.set_profiling(False)
    n = rows * cols
    ev = S - (K - 1)
    r = {'levels': int(zs.size), 'z_top_m': float(zs.max()), 'S': S, 'K': K, 'evaluated_samples_per_ray': ev, 'level_top_share_of_evaluated': K / ev,
         'rays_per_s': n / dt, 'ms_per_step': dt * 1e3, 'march_ms_per_step': ms_march / steps, 'crossings_ms_per_step': ms_pre / steps,
         'march_ps_per_ray_and_evaluated_sample': ms_march / steps * 1e9 / (n * ev), 'mean_hydro_m': float(torch.nanmean(oh).item()),
         'nan_fraction': float(torch.isnan(oh).double().mean().item())}
    prof, _ = load_kernel_counters('real_levels')
    km = ((prof or {}).get(model) or {})
    r['valu_per_raywave'] = km.get('valu_per_raywave')
    r['valu_per_evaluated_sample'] = (km['valu_per_raywave'] / ev) if km.get('valu_per_raywave') else None
    if block:
        nb = int(min(block, rows, cols)); r0, c0 = (rows - nb) // 2, (cols - nb) // 2
        w_, h_ = oracle_block(c, xpts, ypts, inc_cols, hd, zref, nparts, r0, c0, nb)
        gw, gh = ow[r0:r0 + nb, c0:c0 + nb].cpu().numpy(), oh[r0:r0 + nb, c0:c0 + nb].cpu().numpy()
        r['gpu_vs_oracle_max_abs_m'] = float(max(np.nanmax(np.abs(gw - w_)), np.nanmax(np.abs(gh - h_))))
        r['oracle_block'] = [nb, nb]
    return r


def load_kernel_counters(tag):
    """profiles/r*_<tag>_counters.json (tools/gather_digest.py: rocprofv3 --kernel-trace + FETCH_SIZE / WRITE_SIZE passes over
    `bench.py --workload <tag>`) made with the CURRENT kernels; (dict, path) or (None, reason)."""
    want = kernel_source_hash()
    why = f'no profiles/r*_{tag}_counters.json'
    for f in sorted((REPO / 'profiles').glob(f'r*_{tag}_counters.json'), reverse=True):
        try:
            d = json.loads(f.read_text())
        except (OSError, ValueError):
            continue
        if d.get('source_hash') != want:
            why = f'{f.name}: source hash {d.get("source_hash")} != current {want} (stale profile)'
            continue
        return d, str(f.relative_to(REPO))
    return None, why


def c2_inputs(n_side):
    """BASELINE configs[1]: the ERA5-sized 300x300x80 processed model (f64 totals), n_side x n_side query points on the
    scene of configs[2] (lon -119.5..-115.5, lat 34.5..31.5) with their own heights rng(1).uniform(0, 3000) m, Conventional line of
    sight with a fixed incidence of 39 deg (heading -167.9 deg: it does not enter delay / cos(inc), losreader.py:130-133).  The
    intermediate grid is the one tropo_delay lays out for a station AOI: the points' bounding box at the model's own spacing
    (delay.py:142-151, llreader.py:173-191), every model level."""
    from raider_amd.synthetic import synthetic_cube, scene_grid
    from raider_amd.delay import PointsAOI
    c = synthetic_cube(300, 300, 80, seed=0)
    xp, yp, _, _ = scene_grid(n_side, n_side)
    xx, yy = np.meshgrid(xp, yp)
    lats, lons = np.ascontiguousarray(yy.ravel()), np.ascontiguousarray(xx.ravel())
    hgts = np.random.default_rng(1).uniform(0.0, 3000.0, lats.size)
    aoi = PointsAOI(lats, lons, hgts)
    aoi.set_output_spacing(ll_res=min(np.diff(c['xs']).mean(), np.diff(c['ys']).mean()))
    aoi.set_output_xygrid(4326)
    return c, lats, lons, hgts, np.asarray(aoi.xpts), np.asarray(aoi.ypts), np.asarray(c['zs'], dtype=np.float64)


def c2_measure(ctx, dev, n_side=1000, steps=20, warmup=3, oracle_sample=200000, block=None, cube_tensors=None, e2e=False, sync=None):
    """One step of configs[1] = the point branch of tropo_delay (delay.py:96-128) for this rank's block of the points, inputs resident
    in HBM: _build_cube of the intermediate delay cube on the AOI grid (rdr_build_cube_to_cube: setup + gather + packing, the cube stays
    on the device) and ONE gather of both fields at the points with delay / cos(inc) in the same launch (rdr_interp3_project).
    Returns the raw figures; `block` = (p0, cnt) of the flattened point list (default: all), `cube_tensors` = (wet_total, hydro_total)
    device tensors when the cube came through a broadcast."""
    import torch
    import raider_amd as R
    c, lats, lons, hgts, xg, yg, zl = c2_inputs(n_side)
    n_all = lats.size
    p0, cnt = block if block is not None else (0, n_all)
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    wt, ht = cube_tensors if cube_tensors is not None else (torch.from_numpy(c['wet_total']).to(dev), torch.from_numpy(c['hydro_total']).to(dev))
    tot = R.Cube(c['ys'], c['xs'], c['zs'], wt, ht, order='zyx', ctx=ctx)
    yt, xt, zt = (torch.from_numpy(np.ascontiguousarray(a[p0:p0 + cnt])).to(dev) for a in (lats, lons, hgts))
    inc = 39.0
    out = [None, None]
    keep = [None]

    def build():
        keep[0] = tot.build_delay_cube(xg, yg, zl)

    def gather():
        out[0], out[1] = keep[0].interp_project(yt, xt, zt, inc=inc)

    def step():
        build(); gather()
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    if sync is not None:
        sync()                                   # N > 1: every rank enters the timed region together (barrier)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    if sync is not None:
        sync()
    dt = time.perf_counter() - t0
    # the two halves by themselves, HIP events on the stream the kernels run on (the library's own pairs around every launch)
    ctx.set_profiling(True)
    for _ in range(steps):
        build()
    torch.cuda.synchronize()
    n_b, ms_b = ctx.profile_get(2)
    ctx.set_profiling(True)
    for _ in range(steps):
        gather()
    torch.cuda.synchronize()
    n_g, ms_g = ctx.profile_get(2)
    ctx.set_profiling(False)
    nodes = int(xg.size * yg.size * zl.size)
    r = dict(points_all=n_all, points_this_rank=cnt, p0=p0, nodes=nodes, grid=[int(zl.size), int(yg.size), int(xg.size)], step_s=dt / steps,
             build_ms=ms_b / steps, build_launches=n_b, gather_ms=ms_g / steps, gather_launches=n_g, inc=inc,
             wet=out[0], hydro=out[1], has_nan=bool(keep[0].has_nan()),
             mean_hydro=float(torch.nanmean(out[1]).item()), mean_wet=float(torch.nanmean(out[0]).item()), nan_fraction=float(torch.isnan(out[1]).double().mean().item()))
    if e2e:       # the same job through the host-buffer entry (rdr_point_delays: points up, delays down, everything between on the device)
        la, lo, hg = (np.ascontiguousarray(a[p0:p0 + cnt]) for a in (lats, lons, hgts))
        for _ in range(3):                            # warm-up: scratch, page-locked result blocks in the recycling pool
            ew = eh = None
            ew, eh, _ = tot.point_delays(xg, yg, zl, la, lo, hg, inc=inc)
        reps = 5
        t0 = time.perf_counter()
        for _ in range(reps):
            ew = eh = None                            # (the caller is done with the previous results: their blocks are recycled)
            ew, eh, _ = tot.point_delays(xg, yg, zl, la, lo, hg, inc=inc)
        dte = (time.perf_counter() - t0) / reps
        r['e2e'] = {'value': cnt / dte, 'unit': 'points/s', 'ms': dte * 1e3, 'h2d_bytes': cnt * 24, 'd2h_bytes': cnt * 16,
                    'what': 'the same job through rdr_point_delays (NumPy arrays in, NumPy arrays out): points up, intermediate cube built while they travel, one gather, '
                            'delays down', 'bit_identical_to_device_path': bool(np.array_equal(eh, out[1].cpu().numpy(), equal_nan=True) and np.array_equal(ew, out[0].cpu().numpy(), equal_nan=True))}
    if oracle_sample:
        from oracle import raider_oracle as O
        ns = int(min(cnt, oracle_sample))
        t0 = time.perf_counter()
        ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet_total'], c['hydro_total']))
        cw, ch = O.build_cube(xg, yg, zl, ip)
        t_build = time.perf_counter() - t0
        sel = np.linspace(0, cnt - 1, ns).astype(np.int64) + p0
        t0 = time.perf_counter()
        ow, oh = O.points_from_cube(lats[sel], lons[sel], hgts[sel], xg, yg, zl, cw, ch)
        ow, oh = ow / O.cosd(inc), oh / O.cosd(inc)
        t_pts = time.perf_counter() - t0
        gw, gh = out[0].cpu().numpy()[sel - p0], out[1].cpu().numpy()[sel - p0]
        r['oracle'] = {'sample_points': ns, 'build_s': t_build, 'points_s': t_pts,
                       'points_per_s': ns / (t_build * ns / n_all + t_pts),
                       'max_abs': float(max(np.nanmax(np.abs(ow - gw)), np.nanmax(np.abs(oh - gh)))),
                       'nan_masks_equal': bool(np.array_equal(np.isnan(oh), np.isnan(gh)))}
    return r


def c2_roofline(m):
    """168 B per trilinear query on the f64 totals (8 corners x 16 B + 24 B of coordinates + 16 B of results) - the 10^6 points
    AND the nodes of the intermediate cube are such queries; the packing of the planar results into the (y,x,z) cube moves 32 B per node."""
    alg_g = 168.0 * m['points_this_rank']
    alg_b = (168.0 + 32.0) * m['nodes']
    kern_ms = m['build_ms'] + m['gather_ms']
    prof, src = load_kernel_counters('c2')
    k = (prof or {}).get('kernels', {})
    traffic = sum(v.get('hbm_read_bytes', 0) + v.get('hbm_write_bytes', 0) for v in k.values() if v.get('per_step')) if k else None
    return {'bound': 'hbm', 'kernel': 'build_cube_setup_kernel + build_cube_kernel<double2> + pack_cube_xfast_kernel<double> + interp_points_kernel<double2> (one step of one rank)',
            'achieved': (alg_g + alg_b) / (kern_ms * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s', 'frac': (alg_g + alg_b) / (kern_ms * 1e-3) / 1e9 / HBM_PEAK_GBS,
            'traffic': traffic, 'traffic_unit': 'HBM bytes per step, all kernels of the step (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE, separate passes)',
            'frac_hbm_measured': (traffic / (m['step_s']) / 1e9 / HBM_PEAK_GBS) if traffic else None,
            'algorithmic_bytes_per_step': alg_g + alg_b, 'gather_ms_per_step': m['gather_ms'], 'build_ms_per_step': m['build_ms'],
            'gather_algorithmic_GBps': alg_g / (m['gather_ms'] * 1e-3) / 1e9, 'build_algorithmic_GBps': alg_b / (m['build_ms'] * 1e-3) / 1e9,
            'kernels_ms_per_step': kern_ms, 'launch_and_sync_ms_per_step': m['step_s'] * 1e3 - kern_ms,
            'note': 'algorithmic bytes over the HIP-event time of the step\'s four kernels (setup + gather of the intermediate cube, its packing, the point gather); the 39 MB '
                    'intermediate cube and the source cube\'s AOI slab are L2 / Infinity-Cache resident, so the algorithmic rate may exceed the HBM peak - `traffic` is what HBM saw. '
                    'Nothing in the step waits on the host (the intermediate cube is made asynchronously, its NaN verdict read after the loop): launch_and_sync_ms_per_step is launch gaps',
            'counters_source': src if prof is not None else None, 'counters_note': None if prof is not None else src,
            'source_hash': kernel_source_hash()}


def compact_secondary(m):
    """What the default line carries of a gather workload: kernel-level rate, step rate, roofline fraction, parity error."""
    if 'stations_this_rank' in m:
        return {'workload': 'configs[4]: 5 M stations, two-epoch blend of 1000x1000x50 f32 cubes', 'value': m['stations_this_rank'] / m['step_s'], 'unit': 'points/s',
                'ms_per_step': m['step_s'] * 1e3, 'kernels_ms_per_step': m['blend_ms'] + m['interp_ms'], 'roofline_frac': m['alg_bytes'] / ((m['blend_ms'] + m['interp_ms']) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                'roofline_bound': 'hbm', 'gpu_vs_oracle_max_abs': m.get('oracle', {}).get('max_abs'), 'oracle_sample_points': m.get('oracle', {}).get('sample_points'),
                'route': m['route']}
    rf = c2_roofline(m)
    return {'workload': 'configs[1]: Conventional (inc 39 deg), 1000x1000 points, 300x300x80 f64 totals cube', 'value': m['points_this_rank'] / m['step_s'], 'unit': 'points/s',
            'ms_per_step': m['step_s'] * 1e3, 'kernels_ms_per_step': rf['kernels_ms_per_step'], 'gather_points_per_s_kernel': m['points_this_rank'] / (m['gather_ms'] * 1e-3),
            'roofline_frac': rf['frac'], 'roofline_bound': 'hbm', 'hbm_traffic_bytes_per_step': rf['traffic'],
            'gpu_vs_oracle_max_abs_m': m.get('oracle', {}).get('max_abs'), 'oracle_sample_points': m.get('oracle', {}).get('sample_points')}


def run_c2(args, ctx, dev, coll_dev, dist_on, rank, world, ndev, result_fd):
    """BASELINE configs[1]: Conventional slant delay at 1000 x 1000 query points through the ERA5-sized cube (c2_inputs / c2_measure).
    N > 1: the f64 totals go out in ONE packed broadcast, every rank builds the (small) intermediate cube and gathers its contiguous
    block of the point list - no data-path collective (the intermediate grid is the job's, not the block's: same bits as one rank)."""
    import torch
    import torch.distributed as dist
    from raider_amd import distributed as D
    import raider_amd as R
    n_side = int(args.points)
    cube_tensors = None
    t_bcast = 0.0
    if dist_on:
        from raider_amd.synthetic import synthetic_cube
        fields = None
        if rank == 0:
            c = synthetic_cube(300, 300, 80, seed=0)
            fields = dict(ys=c['ys'], xs=c['xs'], zs=c['zs'], wet=c['wet_total'], hydro=c['hydro_total'])
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        axes, wt, ht = D.broadcast_cube_packed(fields, src=0, device=coll_dev, header=(300, 300, 80, 1, 80, 300, 300))
        if coll_dev is None:
            wt, ht = wt.to(dev), ht.to(dev)
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
        cube_tensors = (wt, ht)
    n_all = n_side * n_side
    p0, cnt = D.shard_rows(n_all, world, rank)
    if dist_on:
        dist.barrier()
    osamp = (200000 if args.cpu_sample > 0 else 0) if world == 1 else 64 * max(args.parity_block, 0)
    m = c2_measure(ctx, dev, n_side=n_side, steps=args.steps, warmup=args.warmup, oracle_sample=osamp,
                   block=(p0, cnt), cube_tensors=cube_tensors, e2e=(world == 1 and not args.no_e2e), sync=(dist.barrier if dist_on else None))
    dt = m['step_s'] * args.steps
    rank_s = [dt]
    if dist_on:
        rank_s = gather_rank_times(dist, dt, coll_dev, world)
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev is not None else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    devices = device_identities(torch, dist, dist_on, args.backend, dev.index, rank, world)
    parity_sample = None
    if world > 1 and 'oracle' in m:
        allp = [None] * world
        dist.all_gather_object(allp, [m['oracle']['max_abs'], m['oracle']['sample_points'], bool(m['oracle']['nan_masks_equal'])])
        parity_sample = {'points_compared_all_ranks': int(sum(a_[1] for a_ in allp)), 'max_abs_m': max(a_[0] for a_ in allp), 'per_rank_max_abs_m': [a_[0] for a_ in allp],
                         'nan_masks_equal': all(a_[2] for a_ in allp), 'tolerance_m': 1e-6,
                         'what': 'every rank: evenly spaced points of ITS block vs the NumPy oracle (getInterpolators + build_cube + points_from_cube + / cosd(inc)); max over ranks'}
    one_gpu = None; eff = None
    if world > 1 and not args.no_one_gpu_ref:
        torch.cuda.synchronize(); dist.barrier()
        if rank == 0:                        # rank 0 alone, the whole point list (the other ranks wait at the barrier)
            k1 = max(1, min(args.steps, 3))
            m1 = c2_measure(ctx, dev, n_side=n_side, steps=k1, warmup=1, oracle_sample=0, block=None, cube_tensors=cube_tensors, e2e=False, sync=None)
            one_gpu, eff = scaling_reference('strong', m1['step_s'] * 1e3, n_all, world, dt / args.steps * 1e3, k1,
                                             f'rank 0 alone, other ranks idle at a barrier: all {n_all} points, same step (intermediate cube + gather), after the timed region of the same run')
            del m1
        dist.barrier()
    if args.dump:
        np.savez(f'{args.dump}.rank{rank}.npz', wet=m['wet'].cpu().numpy(), hydro=m['hydro'].cpu().numpy(), p0=p0, cnt=cnt)
    if rank == 0:
        res = {'metric': 'query points/sec (Conventional slant delay: intermediate delay cube + wet/hydro gather + 1/cos(inc)) through ERA5 cube; achieved HBM GB/s',
               'value': n_all * args.steps / dt, 'unit': 'points/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': dt / args.steps * 1e3,
               'higher_is_better': True, 'scaling': 'strong' if world > 1 else 'weak', 'vs_baseline': None, 'dtype': 'f64', 'data': 'synthetic',
               'config': {'workload': f'configs[1]: Conventional slant (fixed incidence {m["inc"]} deg, heading -167.9 deg), {n_side}x{n_side} query points with heights '
                                      f'rng(1).uniform(0, 3000) m, synthetic ERA5-sized 300x300x80 f64 totals cube; one step = the point branch of tropo_delay with inputs '
                                      f'resident in HBM: _build_cube of the {m["grid"][0]}x{m["grid"][1]}x{m["grid"][2]} intermediate cube (bounding box of the points at the '
                                      f'model spacing, every model level) + one gather of both fields at the points + delay / cos(inc)',
                          'points_all_gpus': n_all, 'points_this_rank': cnt, 'intermediate_nodes': m['nodes'], 'cube': '300x300x80 f64 totals',
                          'parallelism': (f'points sharded x{world} ({args.backend}, {ndev} device(s) visible), cube: one packed broadcast ({t_bcast*1e3:.1f} ms), intermediate cube '
                                          f'replicated per rank, no data-path collective') if dist_on else 'single GPU',
                          'ranks': world, 'backend': (dist.get_backend() if dist_on else None), 'world_size_seen_by_backend': (dist.get_world_size() if dist_on else 1),
                          'shards': [list(D.shard_rows(n_all, world, r_)) for r_ in range(world)], 'rank_ms_per_step': [t_ / args.steps * 1e3 for t_ in rank_s],
                          'devices_visible': ndev, 'ranks_per_device': -(-world // ndev),
                          'mean_hydro_m': m['mean_hydro'], 'mean_wet_m': m['mean_wet'], 'nan_fraction': m['nan_fraction'], 'intermediate_cube_has_nan': m['has_nan']},
               'roofline': dict(c2_roofline(m), library_source_hash=R.load_library().rdr_source_hash().decode())}
        res['config']['devices'] = devices
        res['config']['distinct_devices'] = len({(d_['uuid'], d_['pci']) for d_ in devices})
        if parity_sample is not None:
            res['parity_sample'] = parity_sample
        if one_gpu is not None:
            res['one_gpu_same_scene'] = one_gpu; res['scaling_efficiency'] = eff; res['strong_scaling_efficiency'] = eff
            res['scaling_efficiency_is'] = 'strong: t(1 GPU, all points) / (N x t(N GPUs))'
        if 'e2e' in m:
            res['end_to_end'] = m['e2e']
        if 'oracle' in m and world == 1:
            o = m['oracle']
            res['cpu_baseline'] = {'value': o['points_per_s'], 'unit': 'points/s', 'cores': 1, 'kind': 'port',
                                   'sample': f'NumPy oracle (oracle/raider_oracle.py: getInterpolators + build_cube + points_from_cube + / cosd(inc)), one thread: the intermediate '
                                             f'cube of the whole job ({o["build_s"]:.2f} s, charged pro rata) + {o["sample_points"]} of the points ({o["points_s"]:.2f} s)',
                                   'gpu_vs_oracle_max_abs_m': o['max_abs'], 'nan_masks_equal': o['nan_masks_equal']}
        os.write(result_fd, (json.dumps(res) + '\n').encode())
    if dist_on:
        dist.destroy_process_group()


def c5_measure(ctx, dev, stations=5_000_000, steps=5, warmup=2, oracle_sample=20000):
    """One rank's step of configs[4] (run_c5 is the full line with the N > 1 path): blend of the two resident HRRR-sized epochs + gather of
    the stations, kernel times from HIP events; returns the raw figures for the `secondary` entry of the default line."""
    import torch
    import raider_amd as R
    from raider_amd import distributed as D
    ny = nx = 1000; nz = 50
    xs = -1.5e6 + 3000.0 * np.arange(nx); ys = -1.5e6 + 3000.0 * np.arange(ny)
    zs = np.round(-100.0 + 26100.0 * np.linspace(0, 1, nz) ** 2, 3)
    ep = [hrrr_epoch(s_, ys, xs, zs) for s_ in (0, 1)]
    ctx.set_stream(torch.cuda.current_stream().cuda_stream)
    a, b = (R.Cube(ys, xs, zs, torch.from_numpy(e['wet']).to(dev), torch.from_numpy(e['hydro']).to(dev), order='zyx', ctx=ctx) for e in ep)
    w1, w2 = 0.25, 0.75
    rng = np.random.default_rng(3)
    pts_all = np.stack([rng.uniform(-1.4e6, 1.4e6, stations), rng.uniform(-1.4e6, 1.4e6, stations), rng.uniform(0.0, 4000.0, stations)], -1)
    pts = torch.from_numpy(pts_all).to(dev)
    fly = D.blend_on_the_fly_pays(a, stations)
    out = [None, None]

    def step():
        if fly:
            out[0], out[1] = a.interp_blend(w1, b, w2, pts)
        else:
            out[0], out[1] = a.interp_blend(w1, b, w2, pts, via_cube=True)
    for _ in range(warmup):
        step()
    torch.cuda.synchronize()
    ctx.set_profiling(True)
    t0 = time.perf_counter()
    for _ in range(steps):
        step()
    torch.cuda.synchronize()
    dt = time.perf_counter() - t0
    _, ms_int = ctx.profile_get(2)
    _, ms_bl = ctx.profile_get(3)
    ctx.set_profiling(False)
    cells = ny * nx * nz
    r = dict(stations_this_rank=stations, step_s=dt / steps, interp_ms=ms_int / steps, blend_ms=ms_bl / steps, route='blend at the corners' if fly else 'blended cube (paired x columns, scratch) + gather',
             alg_bytes=(168.0 * stations) if fly else (24.0 * cells + 104.0 * stations))
    if oracle_sample:
        from oracle import raider_oracle as O
        ns = int(min(stations, oracle_sample))
        bw = O.blend_cubes(w1, ep[0]['wet'], w2, ep[1]['wet']); bh = O.blend_cubes(w1, ep[0]['hydro'], w2, ep[1]['hydro'])
        ip = list(O.getInterpolators(xs, ys, zs, bw, bh))
        ow, oh = ip[0](pts_all[:ns]), ip[1](pts_all[:ns])
        r['oracle'] = {'sample_points': ns, 'max_abs': float(max(np.nanmax(np.abs(ow - out[0][:ns].cpu().numpy())), np.nanmax(np.abs(oh - out[1][:ns].cpu().numpy()))))}
    return r


def c5_traffic(fly):
    """Measured HBM bytes of one c5 step (blend + gather, or the blend-at-the-corners gather) from the digest at the current source hash; None without one."""
    prof, _ = load_kernel_counters('c5')
    if prof is None:
        return None
    k = prof.get('kernels', {})
    names = ('interp_points_blend_kernel',) if fly else ('blend_pair_kernel', 'interp_points_pair_kernel')
    if not all(n in k and 'hbm_read_bytes' in k[n] for n in names):
        return None
    return sum(k[n]['hbm_read_bytes'] + k[n]['hbm_write_bytes'] for n in names)


def hrrr_epoch(seed, ys, xs, zs):
    """Synthetic HRRR-sized epoch of SURVEY 8(d) (configs[4]): (z,y,x) f32 wet / hydro on the 3-km LCC lattice."""
    rng = np.random.default_rng(seed)
    ny, nx = ys.size, xs.size
    gh = rng.standard_normal((ny, nx)).astype(np.float32); gw = rng.standard_normal((ny, nx)).astype(np.float32)
    z3 = zs[:, None, None]
    hyd = (np.float32(270.0) * np.exp(-z3 / 8000.0).astype(np.float32) * (1 + np.float32(0.01) * gh[None])).astype(np.float32)
    wet = (np.float32(60.0) * np.exp(-z3 / 2000.0).astype(np.float32) * (1 + np.float32(0.1) * gw[None])).astype(np.float32)
    return dict(ys=ys, xs=xs, zs=zs, wet=wet, hydro=hyd)


def run_c5(args, ctx, dev, coll_dev, dist_on, rank, world, ndev, result_fd):
    """BASELINE configs[4]: HRRR 3-km cube (1000x1000x50, f32), two-epoch temporal interpolation (cli/raider.py:817-819), 5 M GNSS
    station-height points.  Synthetic epochs per SURVEY 8(d) (seeds 0 and 1, weights 0.25 / 0.75) on the LCC lattice; stations
    rng(3) uniform in the cube interior, h ~ U(0, 4000).  One step = blend the two resident epochs on this rank + gather this rank's
    contiguous block of the station list (distributed.interp_points_sharded): station work shards with NO data-path collective,
    the blend is replicated (every rank needs the whole blended cube)."""
    import torch
    import torch.distributed as dist
    import raider_amd as R
    from raider_amd import distributed as D
    ny = nx = 1000; nz = 50
    header = (ny, nx, nz, 0, nz, ny, nx)
    xs = -1.5e6 + 3000.0 * np.arange(nx); ys = -1.5e6 + 3000.0 * np.arange(ny)
    zs = np.round(-100.0 + 26100.0 * np.linspace(0, 1, nz) ** 2, 3)

    epochs = [hrrr_epoch(0, ys, xs, zs), hrrr_epoch(1, ys, xs, zs)] if rank == 0 else None
    w1, w2 = 0.25, 0.75
    t_bcast = 0.0
    if dist_on:
        torch.cuda.synchronize(); dist.barrier()
        t0 = time.perf_counter()
        cubes = []
        for k in range(2):                      # one packed broadcast per epoch (RCCL over xGMI), packed (y,x,z) on every rank's device
            axes, wet, hyd = D.broadcast_cube_packed(epochs[k] if rank == 0 else None, src=0, device=coll_dev, header=header)
            if coll_dev is None:
                wet, hyd = wet.to(dev), hyd.to(dev)
            cubes.append(R.Cube(ys, xs, zs, wet, hyd, order='zyx', ctx=ctx))
        torch.cuda.synchronize()
        t_bcast = time.perf_counter() - t0
    else:
        cubes = [R.Cube(ys, xs, zs, torch.from_numpy(e['wet']).to(dev), torch.from_numpy(e['hydro']).to(dev), order='zyx', ctx=ctx) for e in epochs]
    a, b = cubes
    n_all = int(args.stations)
    rng = np.random.default_rng(3)
    pts_all = np.stack([rng.uniform(-1.4e6, 1.4e6, n_all), rng.uniform(-1.4e6, 1.4e6, n_all), rng.uniform(0.0, 4000.0, n_all)], -1)   # (y, x, z), every rank the same list
    p0, cnt = D.shard_rows(n_all, world, rank)
    pts = torch.from_numpy(np.ascontiguousarray(pts_all[p0:p0 + cnt])).to(dev)
    out = [None, None]
    # the blend as a cube (24 B per cell, replicated on every rank) + a 4-line gather, or applied at the corners of this rank's points
    # (8 lines per point, no cube): whichever moves fewer bytes for THIS rank's block - the same bits either way
    fly = D.blend_on_the_fly_pays(a, cnt)

    def step():
        if fly:
            out[0], out[1] = a.interp_blend(w1, b, w2, pts)
        else:
            out[0], out[1] = a.interp_blend(w1, b, w2, pts, via_cube=True)

    for _ in range(args.warmup):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    ctx.set_profiling(True)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    for _ in range(args.steps):
        step()
    torch.cuda.synchronize()
    if dist_on:
        dist.barrier()
    dt = time.perf_counter() - t0
    n_int, ms_int = ctx.profile_get(2)
    _, ms_bl = ctx.profile_get(3)                 # the blend's own launches (HIP event pairs on the stream the kernels run on); 0 on the corner route
    ctx.set_profiling(False)
    blend_ms = ms_bl / args.steps
    rank_s = [dt]
    if dist_on:
        rank_s = gather_rank_times(dist, dt, coll_dev, world)
        tmax = torch.tensor([dt], dtype=torch.float64, device=coll_dev if coll_dev is not None else 'cpu')
        dist.all_reduce(tmax, op=dist.ReduceOp.MAX)
        dt = float(tmax.item())
    wet_t, hyd_t = out
    if args.dump:
        np.savez(f'{args.dump}.rank{rank}.npz', wet=wet_t.cpu().numpy(), hydro=hyd_t.cpu().numpy(), p0=p0, cnt=cnt)
    devices = device_identities(torch, dist, dist_on, args.backend, dev.index, rank, world)
    parity_sample = None
    if world > 1 and args.parity_block > 0:
        # the first points of EVERY rank's block travel to rank 0 (a few KB), which holds the epochs and runs the NumPy oracle once for all of them
        ns_r = int(min(cnt, 8 * args.parity_block))
        allp = [None] * world
        dist.all_gather_object(allp, (p0, wet_t[:ns_r].cpu().numpy(), hyd_t[:ns_r].cpu().numpy()))
        if rank == 0:
            from oracle import raider_oracle as O
            bw = O.blend_cubes(w1, epochs[0]['wet'], w2, epochs[1]['wet']); bh = O.blend_cubes(w1, epochs[0]['hydro'], w2, epochs[1]['hydro'])
            ip = list(O.getInterpolators(xs, ys, zs, bw, bh))
            errs = []
            for q0, gw_, gh_ in allp:
                ow_, oh_ = ip[0](pts_all[q0:q0 + gw_.size]), ip[1](pts_all[q0:q0 + gw_.size])
                errs.append(float(max(np.nanmax(np.abs(ow_ - gw_)), np.nanmax(np.abs(oh_ - gh_)))) if gw_.size else 0.0)
            parity_sample = {'points_compared_all_ranks': int(sum(a_[1].size for a_ in allp)), 'max_abs': max(errs), 'per_rank_max_abs': errs, 'unit': 'N units of refractivity',
                             'tolerance': 1e-9, 'what': 'the first points of every rank\'s block vs the NumPy oracle (blend_cubes + scipy-RGI restatement) on rank 0; max over ranks'}
            del bw, bh, ip
    one_gpu = None; eff = None
    if world > 1 and not args.no_one_gpu_ref:
        torch.cuda.synchronize(); dist.barrier()
        if rank == 0:                        # rank 0 alone, the whole station list (the other ranks wait at the barrier)
            pts1 = torch.from_numpy(pts_all).to(dev)
            fly1 = D.blend_on_the_fly_pays(a, n_all)
            k1 = max(1, min(args.steps, 3))

            def step1():
                return a.interp_blend(w1, b, w2, pts1, via_cube=not fly1)
            step1(); torch.cuda.synchronize()
            t1 = time.perf_counter()
            for _ in range(k1):
                o1 = step1()
            torch.cuda.synchronize()
            t1 = (time.perf_counter() - t1) / k1 * 1e3
            one_gpu, eff = scaling_reference('strong', t1, n_all, world, dt / args.steps * 1e3, k1,
                                             f'rank 0 alone, other ranks idle at a barrier: all {n_all} stations ({"blend at the corners" if fly1 else "blended cube + gather"}), '
                                             f'after the timed region of the same run')
            one_gpu['same_bits_as_sharded_run'] = bool(torch.equal(o1[1][p0:p0 + cnt], hyd_t) and torch.equal(o1[0][p0:p0 + cnt], wet_t))
            del pts1, o1
        dist.barrier()
    if rank != 0:
        if dist_on:
            dist.destroy_process_group()
        return
    cells = ny * nx * nz
    interp_ms = ms_int / args.steps
    step_ms = dt / args.steps * 1e3
    # algorithmic bytes of one rank's step: blend 24 B per f32 cell (two reads, one write, both fields = 2 x 12), gather 104 B per point
    # (on the fly: 8 corners x 2 epochs x 8 B + 24 B of point + 16 B of delays = 168 B per station, SURVEY 8d's two-epoch form for an f32 cube)
    alg_bytes = (168.0 * cnt) if fly else (24.0 * cells + 104.0 * cnt)
    res = {
        'metric': 'GNSS station points/sec (two-epoch blend + wet/hydro gather) through HRRR cube; achieved HBM GB/s',
        'value': n_all * args.steps / dt, 'unit': 'points/s', 'n_gpus': world, 'steps': args.steps, 'warmup': args.warmup, 'ms_per_step': step_ms,
        'higher_is_better': True, 'scaling': 'strong' if world > 1 else 'weak', 'vs_baseline': None, 'dtype': 'f32 blend / f64 interpolation', 'data': 'synthetic',
        'config': {'workload': f'configs[4]: HRRR-sized 1000x1000x50 f32 cube on the 3-km LCC grid, two epochs blended ({w1}, {w2}), {n_all} station points '
                               f'(rng(3), h ~ U(0, 4000) m) sharded into {world} contiguous blocks, wet + hydro at every point',
                   'stations_all_gpus': n_all, 'stations_this_rank': cnt, 'cube': '1000x1000x50 x 2 epochs',
                   'blend': 'at the corners of the rank\'s points (rdr_interp3_blend: no blended cube)' if fly else 'blended cube per step in the context\'s scratch, x columns paired, then the gather (rdr_interp3_blend_cube)',
                   'parallelism': (f'stations sharded x{world} ({args.backend}, {ndev} device(s) visible), epochs: two packed broadcasts ({t_bcast*1e3:.1f} ms), blend replicated '
                                   f'per rank, no data-path collective') if dist_on else 'single GPU',
                   'ranks': world, 'backend': (dist.get_backend() if dist_on else None), 'world_size_seen_by_backend': (dist.get_world_size() if dist_on else 1),
                   'shards': [list(D.shard_rows(n_all, world, r_)) for r_ in range(world)], 'rank_ms_per_step': [t_ / args.steps * 1e3 for t_ in rank_s],
                   'devices_visible': ndev, 'ranks_per_device': -(-world // ndev),
                   'mean_hydro': float(torch.nanmean(hyd_t).item()), 'mean_wet': float(torch.nanmean(wet_t).item()), 'nan_fraction': float(torch.isnan(hyd_t).double().mean().item())},
        'roofline': {'bound': 'hbm', 'kernel': 'interp_points_blend_kernel<float2> (one step of one rank)' if fly else 'blend_pair_kernel<float2,2> + interp_points_pair_kernel<float2> (one step of one rank)',
                     'achieved': alg_bytes / ((blend_ms + interp_ms) * 1e-3) / 1e9, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': alg_bytes / ((blend_ms + interp_ms) * 1e-3) / 1e9 / HBM_PEAK_GBS,
                     'traffic': c5_traffic(fly), 'traffic_unit': 'HBM bytes per step of one rank holding ALL stations (rocprofv3 --pmc FETCH_SIZE x2 / WRITE_SIZE; profiles/r*_c5_counters.json)',
                     'algorithmic_bytes_per_step': alg_bytes, 'blend_ms': blend_ms, 'interp_ms_per_step': interp_ms, 'interp_launches_timed': n_int,
                     'note': ('algorithmic bytes (168 B per station: 8 corners x 2 epochs x 8 B + the point + the two delays, SURVEY 8d) over the kernel\'s HIP-event time; a random '
                              'point touches 8 x 128 B lines for them (4 per epoch)') if fly else
                             ('algorithmic bytes (24 B per cell of the blend + 104 B per station, SURVEY 8d) over the two kernels\' HIP-event time; the blend writes x columns '
                              'PAIRED (round 6), so a random point reads 2 lines when its cell starts on an even column and 4 otherwise - 3 on average against the 4 of the '
                              '(y,x,z) layout (572 B per point measured there, profiles/r05_secondary.json)'),
                     'source_hash': kernel_source_hash(), 'library_source_hash': R.load_library().rdr_source_hash().decode()},
    }
    res['config']['devices'] = devices
    res['config']['distinct_devices'] = len({(d_['uuid'], d_['pci']) for d_ in devices})
    if parity_sample is not None:
        res['parity_sample'] = parity_sample
    if one_gpu is not None:
        res['one_gpu_same_scene'] = one_gpu; res['scaling_efficiency'] = eff; res['strong_scaling_efficiency'] = eff
        res['scaling_efficiency_is'] = 'strong: t(1 GPU, all stations) / (N x t(N GPUs))'
    if world == 1 and args.cpu_sample > 0:
        from oracle import raider_oracle as O
        t0 = time.perf_counter()
        bw = O.blend_cubes(w1, epochs[0]['wet'], w2, epochs[1]['wet']); bh = O.blend_cubes(w1, epochs[0]['hydro'], w2, epochs[1]['hydro'])
        t_blend = time.perf_counter() - t0
        ns = min(cnt, 400_000)
        ip = list(O.getInterpolators(xs, ys, zs, bw, bh))
        t0 = time.perf_counter()
        ow, oh = ip[0](pts_all[:ns]), ip[1](pts_all[:ns])
        t_int = time.perf_counter() - t0
        err = float(max(np.nanmax(np.abs(ow - wet_t[:ns].cpu().numpy())), np.nanmax(np.abs(oh - hyd_t[:ns].cpu().numpy()))))
        res['cpu_baseline'] = {'value': ns / (t_blend * ns / n_all + t_int), 'unit': 'points/s', 'cores': 1, 'kind': 'port',
                               'sample': f'NumPy oracle (oracle/raider_oracle.py: blend_cubes + scipy-RGI restatement), one thread: the blend of the two epochs ({t_blend:.2f} s, '
                                         f'charged pro rata) + {ns} of the stations ({t_int:.2f} s)', 'gpu_vs_oracle_max_abs': err}
    os.write(result_fd, (json.dumps(res) + '\n').encode())
    if dist_on:
        dist.destroy_process_group()


def cpu_baseline(args, rows, cols, xpts, ypts, inc_cols, hd, nparts, zref, out_w, out_h, per_pixel=None):
    """CPU baseline on the GPU box's host, same scene, whole-slice nParts:
      * value: the C/OpenMP restatement (oracle/oracle_c.c) on ALL host cores, on a centre block sized for ~10-20 s;
      * numpy_1thread: the NumPy oracle (the reference's own formulation), one thread, on a 320x320 block.
    Both legs also check the GPU result on their block (max |GPU - oracle|).
    per_pixel = (heights of the scene, model interval of every level-table entry): the per-pixel-height workload (C leg only)."""
    from oracle import raider_oracle as O
    from oracle import oracle_c as OC
    from raider_amd.synthetic import synthetic_cube
    ny, nx, nz = (int(v) for v in args.cube.split('x'))
    c = synthetic_cube(ny, nx, nz, seed=0)
    full = None
    if per_pixel is not None:
        full = np.zeros(nz - 1, dtype=np.int32); full[per_pixel[1]] = nparts

    def block(n):
        n = min(n, rows, cols)
        r0 = (rows - n) // 2; c0 = (cols - n) // 2
        xp = xpts[c0:c0 + n]; yp = ypts[r0:r0 + n]
        inc = np.broadcast_to(inc_cols[c0:c0 + n], (n, n))
        xx, yy = np.meshgrid(xp, yp)
        los = O.look_vectors_from_inc_hd(inc, np.full(yy.shape, hd), yy, xx, 0.0)
        gw = out_w[r0:r0 + n, c0:c0 + n].cpu().numpy(); gh = out_h[r0:r0 + n, c0:c0 + n].cpu().numpy()
        hb = per_pixel[0][r0:r0 + n, c0:c0 + n] if per_pixel is not None else None
        return n, xp, yp, inc, los, gw, gh, xx, yy, hb

    def run_c(xp, yp, los, xx, yy, hb):
        if per_pixel is None:
            return OC.build_cube_ray_slice(c, xp, yp, 0.0, los, zref, nparts=nparts)
        return OC.build_cube_ray_per_pixel(c, yy, xx, hb, los, zref, nparts=full)

    # --- C / OpenMP, all cores: calibrate on 256x256, then ~15 s worth of rays
    n, xp, yp, inc, los, gw, gh, xx, yy, hb = block(256)
    t0 = time.perf_counter(); run_c(xp, yp, los, xx, yy, hb); t_cal = time.perf_counter() - t0
    rate = n * n / t_cal
    n_big = int(min(max(256, np.sqrt(rate * 15.0)), args.cpu_sample * 4, rows, cols))
    n, xp, yp, inc, los, gw, gh, xx, yy, hb = block(n_big)
    t0 = time.perf_counter(); cw, ch, _ = run_c(xp, yp, los, xx, yy, hb); dt_c = time.perf_counter() - t0
    err_c = float(max(np.nanmax(np.abs(gw - cw)), np.nanmax(np.abs(gh - ch))))
    res = {'value': n * n / dt_c, 'unit': 'rays/s', 'cores': OC.num_threads(), 'kind': 'port',
           'sample': f'{n}x{n} centre block of the same scene ({n*n} rays, {dt_c:.1f} s), C/OpenMP oracle (oracle/oracle_c.c, '
                     f'both passes, whole-{"batch" if per_pixel is not None else "slice"} nParts) on {OC.num_threads()} threads; host has {os.cpu_count()} logical cores',
           'gpu_vs_oracle_max_abs_m': err_c}
    if per_pixel is not None:
        return res
    # --- NumPy, one thread (the reference's own array formulation)
    n, xp, yp, inc, los, gw, gh, xx, yy, hb = block(min(args.cpu_sample, 320))
    look = lambda ht, llh, xyz, yy: los
    ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro']))
    t0 = time.perf_counter()
    w, h = O.build_cube_ray(xp, yp, np.array([0.0]), look, ip, MAX_TROPO_HEIGHT=zref, nParts_override=[nparts])
    dt_n = time.perf_counter() - t0
    res['numpy_1thread'] = {'value': n * n / dt_n, 'unit': 'rays/s', 'cores': 1,
                            'sample': f'{n}x{n} centre block ({n*n} rays, {dt_n:.1f} s), NumPy oracle (oracle/raider_oracle.py)',
                            'gpu_vs_oracle_max_abs_m': float(max(np.nanmax(np.abs(gw - w[0])), np.nanmax(np.abs(gh - h[0]))))}
    return res


if __name__ == '__main__':
    main()
