"""Raw hipMemcpy (ctypes, no raider_amd) from np.memmap / mmap / resident arrays in a Python process: is the mapping slow here too?"""
import ctypes as C, json, mmap, os, sys, tempfile, time
from pathlib import Path
import numpy as np
hip = C.CDLL(os.environ.get('HIPLIB', '/opt/rocm/lib/libamdhip64.so'))
n = 115_200_000
d = Path(tempfile.mkdtemp()); f = d / 'f.bin'
a = np.random.default_rng(0).random(n // 8)
with open(f, 'wb') as fh:
    fh.write(b'\0' * 1340); fh.write(a.tobytes())
dev = C.c_void_p(); assert hip.hipMalloc(C.byref(dev), C.c_size_t(n)) == 0
st = C.c_void_p(); assert hip.hipStreamCreateWithFlags(C.byref(st), 1) == 0
def best(fn, reps=7):
    ts = []
    for _ in range(reps):
        t0 = time.perf_counter(); fn(); ts.append((time.perf_counter() - t0) * 1e3)
    return round(min(ts[1:]), 3)
def sync_copy(ptr, nb): assert hip.hipMemcpy(dev, C.c_void_p(ptr), C.c_size_t(nb), 1) == 0
def async_copy(ptr, nb):
    assert hip.hipMemcpyAsync(dev, C.c_void_p(ptr), C.c_size_t(nb), 1, st) == 0; assert hip.hipStreamSynchronize(st) == 0
res = {}
for name, cp in (('sync', sync_copy), ('async', async_copy)):
    res[f'resident_{name}_ms'] = best(lambda: cp(a.ctypes.data, n))
    def mm():
        m = np.memmap(f, dtype=np.float64, mode='r', offset=1340, shape=(n // 8,)); cp(m.ctypes.data, n)
    res[f'np_memmap_{name}_ms'] = best(mm)
    def mm2():
        m = np.memmap(f, dtype=np.float64, mode='r', offset=1340, shape=(n // 8,)); cp(m.ctypes.data, n // 2); cp(m.ctypes.data + n // 2, n // 2)
    res[f'np_memmap_two_halves_{name}_ms'] = best(mm2)
    def raw():
        with open(f, 'rb') as fh:
            m = mmap.mmap(fh.fileno(), 0, flags=mmap.MAP_PRIVATE, prot=mmap.PROT_READ)
        b = np.frombuffer(m, dtype=np.uint8); cp(b.ctypes.data + 1340, n)
    res[f'mmap_private_{name}_ms'] = best(raw)
print(json.dumps(res))
