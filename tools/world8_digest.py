#!/usr/bin/env python3
"""gpurun_out/world8 (tools/world8_dryrun.sh) -> profiles/<round>_world8_dryrun.json: the eight-rank runs of the three workloads on ONE GPU
beside their one-rank runs - what the backend saw, every rank's block and own time per step, the broadcast time - next to DESIGN section 5's
projection for a real 8-GPU node.  A rehearsal of the plumbing (the ranks time-share one device), NOT a scaling measurement."""
import json
import sys
from pathlib import Path

REPO = Path(__file__).resolve().parent.parent
RND = sys.argv[1] if len(sys.argv) > 1 else 'r06'
src = REPO / 'gpurun_out' / 'world8'


def line(tag):
    f = src / f'{tag}.json'
    ls = [l for l in f.read_text().splitlines() if l.startswith('{')] if f.exists() else []
    return json.loads(ls[-1]) if ls else None


def covers(shards, total):
    pos = 0
    for r0, cnt in shards:
        if r0 != pos:
            return False
        pos += cnt
    return pos == total


out = {'what': 'bench.py launched as the driver launches a scaling run (python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 ... bench.py --gpus 8) on a '
               'ONE-GPU box: `--backend auto` picks gloo with device-resident collective tensors because RCCL refuses two ranks on one device; same code path otherwise (packed cube '
               'broadcast, pass 1 -> MAX all-reduce of K+4 doubles -> pass 2, barrier + max-over-ranks timing, one JSON line from rank 0).  The ranks TIME-SHARE the device: the 8-rank '
               'times below are not a scaling measurement.', 'runs': {}}
for wl, total_key in (('rays', 'rays_per_step_all_gpus'), ('c5', 'stations_all_gpus'), ('c2', 'points_all_gpus')):
    eight, one = line(f'{wl}8' if wl == 'rays' else f'{wl}_8'), line(f'{wl}1' if wl == 'rays' else f'{wl}_1')
    if not eight or not one:
        continue
    c = eight['config']
    total = {'rays': 10000, 'c5': c.get('stations_all_gpus'), 'c2': c.get('points_all_gpus')}[wl]
    out['runs'][wl] = {
        'workload': c['workload'][:160], 'backend': c['backend'], 'world_size_seen_by_backend': c['world_size_seen_by_backend'], 'devices_visible': c.get('devices_visible'),
        'shards': c.get('shards'), 'shards_cover_the_job_exactly': covers(c.get('shards') or [], total),
        'rank_ms_per_step': c.get('rank_ms_per_step'), 'ms_per_step_line': eight['ms_per_step'], 'value': eight['value'], 'unit': eight['unit'], 'scaling': eight['scaling'],
        'parallelism': c['parallelism'], 'source_hash': eight['roofline'].get('source_hash'),
        'one_rank': {'ms_per_step': one['ms_per_step'], 'value': one['value']},
        'time_shared_slowdown_vs_one_rank': eight['ms_per_step'] / one['ms_per_step']}
pf = src / 'nccl_preflight.err'
out['nccl_preflight'] = {'command': 'python bench.py --gpus 8 --backend nccl', 'exit_code': (src / 'nccl_preflight.rc').read_text().strip() if (src / 'nccl_preflight.rc').exists() else None,
                         'stdout_empty': (src / 'nccl_preflight.out').read_text().strip() == '' if (src / 'nccl_preflight.out').exists() else None,
                         'message': [l for l in pf.read_text().splitlines() if l.startswith('bench.py:')][:1] if pf.exists() else None}
out['projection_for_a_real_node'] = {
    'source': 'DESIGN.md section 5 (measured piece by piece on one GPU through a one-rank RCCL group, bench.py --force-dist --backend nccl --rows R --cols 10000)',
    'slab_ms': {'1250 rows (N=8)': 4.73, '2500 rows (N=4)': 9.29, '5000 rows (N=2)': 18.48, '10000 rows (N=1)': 36.98},
    'projected_8_gpu_ms_per_step': [4.8, 4.9], 'projected_8_gpu_rays_per_s': [20.5e9, 20.9e9], 'projected_efficiency': [0.95, 0.97],
    'note': 'PROJECTION: slab time is linear in the rows, the only serial part is the MAX all-reduce of K+4 doubles (40 us through a one-rank RCCL group; 30-100 us assumed over 8 ranks)'}
(REPO / 'profiles' / f'{RND}_world8_dryrun.json').write_text(json.dumps(out, indent=1) + '\n')
for k, v in out['runs'].items():
    print(k, v['backend'], v['world_size_seen_by_backend'], 'cover', v['shards_cover_the_job_exactly'], 'rank ms', [round(t, 2) for t in (v['rank_ms_per_step'] or [])], 'one-rank', round(v['one_rank']['ms_per_step'], 3))
print(out['nccl_preflight'])
