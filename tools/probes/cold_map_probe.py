"""Why is Cube(<np.memmap of a NetCDF-3 variable>) 2.4 ms slower than Cube(<resident copy>) when a raw hipMemcpy from a fresh mapping is not?"""
import json, mmap, os, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import raider_amd as R
from raider_amd.synthetic import synthetic_cube
c = synthetic_cube(300, 300, 80, seed=0)
shape = c['wet_total'].shape; n = c['wet_total'].size
d = Path(tempfile.mkdtemp())
def mk(w, h): return R.Cube(c['ys'], c['xs'], c['zs'], w, h, order='zyx')
def best(fn, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(min(ts[1:]), 3)
res = {}
for off in (0, 4096, 1340):                       # header size of the fixture file: page-aligned or not
    for be in (False, True):
        f = d / f'f_{off}_{int(be)}.bin'
        dt_ = np.dtype('>f8' if be else '<f8')
        with open(f, 'wb') as fh:
            fh.write(b'\0' * off); fh.write(c['wet_total'].astype(dt_).tobytes()); fh.write(c['hydro_total'].astype(dt_).tobytes())
        def maps():
            return (np.memmap(f, dtype=dt_, mode='r', offset=off, shape=shape), np.memmap(f, dtype=dt_, mode='r', offset=off + n * 8, shape=shape))
        res[f'memmap_off{off}_{"be" if be else "le"}_ms'] = best(lambda: mk(*maps()))
        def one_map():
            m = np.memmap(f, dtype=np.uint8, mode='r')
            w = m[off:off + n * 8].view(dt_).reshape(shape); h = m[off + n * 8:off + 2 * n * 8].view(dt_).reshape(shape)
            return w, h
        res[f'one_mapping_off{off}_{"be" if be else "le"}_ms'] = best(lambda: mk(*one_map()))
w, h = c['wet_total'], c['hydro_total']
res['resident_le_ms'] = best(lambda: mk(w, h))
wb, hb = w.astype('>f8'), h.astype('>f8')
res['resident_be_ms'] = best(lambda: mk(wb, hb))
def touch():
    a, b = np.memmap(d / 'f_1340_1.bin', dtype=np.uint8, mode='r', offset=0), None
    return int(a[::4096].sum())
res['touch_every_page_of_a_fresh_mapping_ms'] = best(touch)
print(json.dumps(res))
# ---- remedies for the fault-bound upload ------------------------------------------------------------------------------------
import ctypes, threading
libc = ctypes.CDLL('libc.so.6', use_errno=True)
f = d / 'f_1340_1.bin'; off = 1340; dt_ = np.dtype('>f8')
def maps():
    return (np.memmap(f, dtype=dt_, mode='r', offset=off, shape=shape), np.memmap(f, dtype=dt_, mode='r', offset=off + n * 8, shape=shape))
def advise(a, advice):
    addr = a.ctypes.data & ~4095; ln = a.nbytes + (a.ctypes.data - addr)
    return libc.madvise(ctypes.c_void_p(addr), ctypes.c_size_t(ln), advice)
res2 = {}
for name, adv in (('MADV_WILLNEED', 3), ('MADV_POPULATE_READ', 22), ('MADV_HUGEPAGE', 14)):
    rc = []
    def run():
        w, h = maps(); rc.append((advise(w, adv), advise(h, adv))); mk(w, h)
    res2[f'{name}_then_cube_ms'] = best(run); res2[f'{name}_rc'] = rc[-1]
def threaded_touch(T):
    def run():
        w, h = maps()
        parts = [v.reshape(-1).view(np.uint8)[k::T * 4096] for v in (w, h) for k in range(0)]  # (placeholder)
        segs = []
        for v in (w, h):
            b = v.reshape(-1).view(np.uint8); step = (b.size // T + 4095) // 4096 * 4096
            segs += [b[i:i + step] for i in range(0, b.size, step)]
        th = [threading.Thread(target=lambda s_=s_: int(s_[::4096].sum())) for s_ in segs]
        [t.start() for t in th]; [t.join() for t in th]
        mk(w, h)
    return run
for T in (2, 4, 8):
    res2[f'touch_{T}x2_threads_then_cube_ms'] = best(threaded_touch(T))
print(json.dumps(res2))
