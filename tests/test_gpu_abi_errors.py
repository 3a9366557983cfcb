"""GPU: the C ABI's argument checking, called the way a foreign binding would get it wrong.  Every `int`-returning entry of
include/raider_hip.h is called (in a child process, so that a crash is a test failure and not the end of the suite) with
  1. NULL / zero for every argument,
  2. a live context, NULL / zero for the rest,
  3. a live context and a live cube, NULL / zero for the rest,
  4. a live context, cube AND an all-zero rdr_rays (n = 0 GRID rays without axes), NULL / zero for the rest,
  5. live objects, -1 for every integer and floating-point argument (negative sizes, counts, modes, dtypes, strides), a zero-filled
     64 KB buffer behind every plain pointer, a rdr_rays with negative counts and modes,
and must come back with a status code - negative with a message in rdr_last_error, or RDR_OK where an empty batch is legal -
never a crash, a hang or a HIP error left behind: after the sweep the same context still traces rays correctly."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

pytestmark = pytest.mark.gpu
ROOT = Path(__file__).resolve().parent.parent

SCRIPT = r'''
import ctypes as C, json, re, sys
sys.path.insert(0, %(root)r)
import numpy as np
from raider_amd import _lib as L
from raider_amd.synthetic import synthetic_cube
lib = L.load()
hdr = open(%(root)r + '/include/raider_hip.h').read()
hdr = re.sub(r'/\*.*?\*/', ' ', hdr, flags=re.S); hdr = re.sub(r'//[^\n]*', ' ', hdr)
protos = {m.group(1): m.group(2) for m in re.finditer(r'\bint\s+(rdr_\w+)\s*\(([^;]*?)\)\s*;', hdr, flags=re.S)}
ctx = C.c_void_p(); assert lib.rdr_create(0, C.byref(ctx)) == 0
c = synthetic_cube(20, 22, 16, seed=0)
ys, xs, zs = (np.ascontiguousarray(c[k], dtype=np.float64) for k in ('ys', 'xs', 'zs'))
wet = np.ascontiguousarray(c['wet'].transpose(1, 2, 0)); hyd = np.ascontiguousarray(c['hydro'].transpose(1, 2, 0))
cube = C.c_void_p()
p = lambda a: a.ctypes.data_as(C.c_void_p)
assert lib.rdr_cube_create(ctx, p(ys), 20, p(xs), 22, p(zs), 16, p(wet), p(hyd), L.RDR_F32, 22 * 16, 16, 1, L.RDR_HOST, C.byref(cube)) == 0, lib.rdr_last_error(ctx)
rays0 = L.RdrRays()
rays_neg = L.RdrRays(); rays_neg.n = -1; rays_neg.nx = -1; rays_neg.ny = -1; rays_neg.origin_mode = -1; rays_neg.los_mode = -1
scratch = C.create_string_buffer(1 << 16)          # valid, zero-filled memory behind every plain pointer of level 5
table = {name: (res, args) for name, res, args in L.SYMBOLS}
calls = 0; bad = []; ok_zero = []
for name, proto in sorted(protos.items()):
    if name in ('rdr_create', 'rdr_version'):
        continue
    res, argtypes = table[name]
    params = [a.strip() for a in proto.split(',')]
    assert len(params) == len(argtypes), (name, params, argtypes)
    for level in range(5):
        vals = []
        for prm, at in zip(params, argtypes):
            if level == 4 and name in ('rdr_host_free', 'rdr_set_stream'):       # (a pointer that is not theirs IS undefined behaviour)
                vals = None; break
            if at in (C.c_int, C.c_int32, C.c_int64): vals.append(at(0 if (level < 4 or re.search(r'\bloc$', prm)) else -1))      # (loc stays HOST: the buffers are)
            elif at in (C.c_double, C.c_float): vals.append(at(0.0 if level < 4 else -1.0))
            elif level >= 1 and re.match(r'rdr_ctx\s*\*\s*\w+$', prm): vals.append(ctx)
            elif level >= 2 and re.match(r'(const\s+)?rdr_cube\s*\*\s*\w+$', prm): vals.append(cube)
            elif level == 4 and re.match(r'const\s+rdr_rays\s*\*\s*\w+$', prm): vals.append(C.pointer(rays_neg) if at is not C.c_void_p else C.cast(C.pointer(rays_neg), C.c_void_p))
            elif level == 4 and not re.search(r'\*\s*\*|\*\s*const\s*\*', prm): vals.append(C.cast(scratch, at) if at is not C.c_void_p else C.cast(scratch, C.c_void_p))
            elif level >= 3 and re.match(r'const\s+rdr_rays\s*\*\s*\w+$', prm): vals.append(C.pointer(rays0) if at is not C.c_void_p else C.cast(C.pointer(rays0), C.c_void_p))
            else: vals.append(None)
        if vals is None:
            continue
        print(name, level, flush=True)                       # (the parent reports the last line on a crash)
        rc = getattr(lib, name)(*vals)
        calls += 1
        if rc > 0 and name not in ('rdr_cube_has_nan', 'rdr_last_nan_output'):
            bad.append((name, level, rc, 'positive status'))
        elif rc < 0:
            msg = lib.rdr_last_error(ctx if level >= 1 else None)
            if not msg:
                bad.append((name, level, rc, 'no message'))
            elif rc not in (L.RDR_ERR_INVALID, L.RDR_ERR_NO_LEVELS):
                bad.append((name, level, rc, msg.decode()))          # nonsense must be refused by name, before the device sees it
        elif rc == 0:
            ok_zero.append((name, level))
# nothing is left behind: no sticky HIP error, the context and the cube still work
assert lib.rdr_synchronize(ctx) == 0, lib.rdr_last_error(ctx)
import raider_amd as R
cb = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
w, h, nparts, _ = cb.raytrace(R.Rays.grid(np.linspace(-119.0, -116.0, 9), np.linspace(34.0, 32.0, 7), inc=35.0, hd=-167.9), 0.0, float(c['zs'].max() - 1))
lib.rdr_cube_destroy(cube); lib.rdr_destroy(ctx)
print(json.dumps(dict(calls=calls, bad=bad, ok_zero=ok_zero, entries=len(protos), finite=bool(np.isfinite(w).all() and np.isfinite(h).all()))))
'''


def test_every_entry_point_survives_null_and_zero_arguments():
    out = subprocess.run([sys.executable, '-c', SCRIPT % dict(root=str(ROOT))], capture_output=True, text=True, timeout=600)
    tail = out.stdout.strip().splitlines()[-3:]
    assert out.returncode == 0, ('crashed at / after:', tail, out.stderr[-3000:])
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['entries'] >= 45 and res['calls'] >= 5 * (res['entries'] - 2) - 2
    assert res['bad'] == []
    assert res['finite']
    # RDR_OK on all-zero arguments is legal only where "nothing to do" is a meaning: setters, queries, empty batches
    legal_ok = {'rdr_host_free', 'rdr_set_stream', 'rdr_forget_stream', 'rdr_synchronize', 'rdr_set_profiling', 'rdr_set_side_capacity', 'rdr_cube_has_nan',
                'rdr_last_nan_output', 'rdr_cube_point_index', 'rdr_trim',      # (rdr_trim(ctx, 0, NULL): free everything, report nothing)
                # every output pointer of these queries is optional; kind 0 CLEARS a projection; n = 0 points is an empty batch
                'rdr_cube_axes', 'rdr_cube_shape', 'rdr_device_info', 'rdr_ray_kernel_attributes', 'rdr_cube_set_projection', 'rdr_interp3', 'rdr_interp3_project', 'rdr_interp3_blend', 'rdr_interp3_blend_cube',
                'rdr_cube_read'}           # (level 5: the 28 KB fields of the test cube fit the scratch buffer - a valid call)
    surprising = sorted({n for n, _ in res['ok_zero']} - legal_ok)
    assert surprising == [], surprising
