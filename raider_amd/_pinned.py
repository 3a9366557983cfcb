"""Recycled page-locked result buffers (include/raider_hip.h: rdr_host_alloc).

The delay cubes `tropo_delay` hands back are hundreds of MB; written into freshly allocated pageable memory every byte costs a
first-touch page fault on top of the PCIe transfer (measured: 512 MB in 31 ms instead of 9.5 ms), and the download cannot overlap
the kernels.  `empty()` returns a NumPy array backed by a page-locked block; when the last view of it is garbage-collected the
block goes back to a free list and the next call of the same size takes it - no page faults, no mmap / munmap churn, downloads at
the link rate.  The pool keeps at most RAIDER_HIP_PINNED_POOL_BYTES of FREE blocks (default 4 GiB divided by the number of ranks on
this host, LOCAL_WORLD_SIZE: eight ranks of one node together page-lock what one process would); RAIDER_HIP_PINNED_POOL_BYTES=0
switches pinned results off (plain np.empty)."""
import collections
import ctypes as C
import os
import sys
import threading
import time

import numpy as np

from . import _lib as L

_GRAN = 2 << 20                      # blocks are multiples of 2 MiB: a slightly different cube shape still finds a block
_MIN_BYTES = 1 << 20                 # smaller results are not worth a page-locked block (10^6 station delays are 8 MB: they are)
_lock = threading.Lock()
_free = {}                           # rounded size -> [pointers]
_free_bytes = 0
_pending = collections.deque()       # (ptr, cap) of blocks whose last view died: filed under the lock by whoever holds it next


def _limit():
    v = os.environ.get('RAIDER_HIP_PINNED_POOL_BYTES')
    if v is not None:
        return int(v)
    try:
        ranks = max(1, int(os.environ.get('LOCAL_WORLD_SIZE', '1')))
    except ValueError:
        ranks = 1
    return (4 << 30) // ranks


class _Block:
    """Owner of one page-locked block; NumPy arrays made from it keep it alive through their `.base` chain."""

    def __init__(self, ptr, cap, nbytes, shape, dtype):
        self.ptr, self.cap = ptr, cap
        self.__array_interface__ = {'shape': tuple(shape), 'typestr': np.dtype(dtype).str, 'data': (ptr, False), 'version': 3}
        self.nbytes = nbytes

    def __del__(self):
        # A finaliser may run INSIDE another thread-of-control's critical section (the cyclic collector fires on any allocation, also
        # one made while this thread holds _lock): it only queues the block - deque.append is atomic and takes no lock - and files it
        # itself when the lock happens to be free.  empty() / trim() / free_bytes() drain the queue too.
        _pending.append((self.ptr, self.cap))
        try:
            if _lock.acquire(blocking=False):
                try:
                    over = _drain_locked()
                finally:
                    _lock.release()
                _give_back(over)
        except Exception:            # interpreter shutdown
            pass


_DEBUG = bool(os.environ.get('RAIDER_HIP_PINNED_DEBUG'))


def _log(*a):
    if _DEBUG and sys is not None:
        print(f'[pinned {time.perf_counter():.4f}]', *a, file=sys.stderr, flush=True)


def _drain_locked():
    """File the queued blocks into the free list (caller holds _lock); returns the pointers beyond the pool limit, to be handed back
    to the driver OUTSIDE the lock."""
    global _free_bytes
    over = []
    lim = _limit()
    while True:
        try:
            ptr, cap = _pending.popleft()
        except IndexError:
            return over
        _log('release', hex(ptr), cap)
        if _free_bytes + cap <= lim:
            lst = _free.get(cap)
            if lst is None:
                lst = _free[cap] = []
            lst.append(ptr)
            _free_bytes += cap
        else:
            over.append(ptr)


def _give_back(ptrs):
    for p in ptrs:
        try:
            L.load().rdr_host_free(C.c_void_p(p))
        except Exception:            # interpreter shutdown
            pass


def empty(shape, dtype=np.float64):
    """np.empty(shape, dtype) in recycled page-locked memory (plain np.empty for small arrays, when the pool is switched off, or
    when page-locked memory cannot be had)."""
    global _free_bytes
    shape = tuple(int(v) for v in np.atleast_1d(shape)) if not isinstance(shape, tuple) else tuple(int(v) for v in shape)
    nbytes = int(np.prod(shape, dtype=np.int64)) * np.dtype(dtype).itemsize
    if nbytes < _MIN_BYTES or _limit() <= 0:
        return np.empty(shape, dtype=dtype)
    cap = (nbytes + _GRAN - 1) // _GRAN * _GRAN
    ptr = None
    with _lock:
        over = _drain_locked()
        lst = _free.get(cap)
        if lst:
            ptr = lst.pop()
            _free_bytes -= cap
    _give_back(over)
    _log('reuse' if ptr is not None else 'alloc', cap)
    if ptr is None:
        p = C.c_void_p()
        try:
            rc = L.load().rdr_host_alloc(cap, C.byref(p))
        except Exception:
            rc = -1
        if rc != L.RDR_OK or not p.value:
            return np.empty(shape, dtype=dtype)
        ptr = p.value
    return np.asarray(_Block(ptr, cap, nbytes, shape, dtype))


def is_pinned(a):
    b = a
    while isinstance(b, np.ndarray) and b.base is not None:
        b = b.base
    return isinstance(b, _Block)


def trim():
    """Give every free block back to the driver."""
    global _free_bytes
    with _lock:
        blocks = _drain_locked()
        blocks += [p for lst in _free.values() for p in lst]
        _free.clear(); _free_bytes = 0
    _give_back(blocks)


def free_bytes():
    with _lock:
        over = _drain_locked()
    _give_back(over)
    return _free_bytes
