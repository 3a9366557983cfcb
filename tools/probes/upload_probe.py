"""How fast does a 58 MB cube file reach the device?  Cube() from: a resident NumPy array, np.memmap (fresh mapping each time), mmap with
MAP_POPULATE, np.fromfile (read into fresh pages), read into a reused buffer."""
import mmap, os, sys, tempfile, time
from pathlib import Path
import numpy as np
sys.path.insert(0, str(Path(__file__).resolve().parent.parent.parent))
import raider_amd as R
from raider_amd.synthetic import synthetic_cube
c = synthetic_cube(300, 300, 80, seed=0)
tmp = Path(tempfile.mkdtemp()) / 'f.bin'
with open(tmp, 'wb') as f:
    f.write(c['wet'].tobytes()); f.write(c['hydro'].tobytes())
n = c['wet'].size; shape = c['wet'].shape
def mk(w, h): return R.Cube(c['ys'], c['xs'], c['zs'], w, h, order='zyx')
def t(fn, reps=6):
    fn(); best = 1e9
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); best = min(best, time.perf_counter() - t0)
    return best * 1e3
print('resident arrays      %.2f ms' % t(lambda: mk(c['wet'], c['hydro'])))
def memmap():
    w = np.memmap(tmp, dtype=np.float32, mode='r', offset=0, shape=shape); h = np.memmap(tmp, dtype=np.float32, mode='r', offset=n * 4, shape=shape)
    mk(w, h)
print('np.memmap            %.2f ms' % t(memmap))
def populate():
    with open(tmp, 'rb') as fh:
        m = mmap.mmap(fh.fileno(), 0, flags=mmap.MAP_PRIVATE | mmap.MAP_POPULATE, prot=mmap.PROT_READ)
    w = np.frombuffer(m, dtype=np.float32, count=n).reshape(shape); h = np.frombuffer(m, dtype=np.float32, count=n, offset=n * 4).reshape(shape)
    mk(w, h)
print('mmap MAP_POPULATE    %.2f ms' % t(populate))
def willneed():
    with open(tmp, 'rb') as fh:
        m = mmap.mmap(fh.fileno(), 0, access=mmap.ACCESS_READ)
    m.madvise(mmap.MADV_WILLNEED); 
    w = np.frombuffer(m, dtype=np.float32, count=n).reshape(shape); h = np.frombuffer(m, dtype=np.float32, count=n, offset=n * 4).reshape(shape)
    mk(w, h)
print('mmap MADV_WILLNEED   %.2f ms' % t(willneed))
def fromfile():
    a = np.fromfile(tmp, dtype=np.float32); mk(a[:n].reshape(shape), a[n:].reshape(shape))
print('np.fromfile          %.2f ms' % t(fromfile))
buf = np.empty(2 * n, dtype=np.float32)
def readinto():
    with open(tmp, 'rb', buffering=0) as fh:
        fh.readinto(memoryview(buf).cast('B'))
    mk(buf[:n].reshape(shape), buf[n:].reshape(shape))
print('readinto reused buf  %.2f ms' % t(readinto))
from raider_amd import _pinned
pb = _pinned.empty((2 * n,), dtype=np.float32)
def readpinned():
    with open(tmp, 'rb', buffering=0) as fh:
        fh.readinto(memoryview(pb).cast('B'))
    mk(pb[:n].reshape(shape), pb[n:].reshape(shape))
print('readinto pinned buf  %.2f ms (pinned: %s)' % (t(readpinned), _pinned.is_pinned(pb)))
