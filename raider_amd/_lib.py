"""ctypes binding of libraider_hip.so (C ABI: include/raider_hip.h).

There is NO CPU fallback anywhere in this package: if the shared library is missing, or no MI355X
is visible, every compute entry point raises RuntimeError.
"""
import ctypes as C
import os
import threading
from pathlib import Path

import numpy as np

_HERE = Path(__file__).resolve().parent
LIB_PATH = Path(os.environ.get('RAIDER_HIP_LIB', _HERE / 'libraider_hip.so'))

RDR_OK = 0
RDR_ERR_INVALID, RDR_ERR_HIP, RDR_ERR_NODEVICE, RDR_ERR_ALL_NAN, RDR_ERR_NO_LEVELS, RDR_ERR_NAN_LENGTH, RDR_ERR_OOM = -1, -2, -3, -4, -5, -6, -7
RDR_F32, RDR_F64 = 0, 1
RDR_BYTESWAPPED = 0x100
RDR_HOST, RDR_DEVICE = 0, 1
ORIGIN_GRID, ORIGIN_LLH, ORIGIN_XYZ = 0, 1, 2
LOS_VEC, LOS_INC_HD, LOS_INC_HD_SCALAR, LOS_ZENITH = 0, 1, 2, 3
FLAG_ANY_NAN, FLAG_ANY_FINITE, FLAG_FIRST_NOT_BELOW, FLAG_LAST_NOT_ABOVE, FLAG_DIVERGED, FLAG_BAD_HEIGHT, FLAG_NAN_OUTPUT = 1, 2, 4, 8, 16, 32, 64

c_dp = C.POINTER(C.c_double)
c_ip = C.POINTER(C.c_int32)
c_lp = C.POINTER(C.c_int64)


class RdrRays(C.Structure):
    _fields_ = [
        ('n', C.c_int64), ('origin_mode', C.c_int32), ('los_mode', C.c_int32),
        ('nx', C.c_int64), ('ny', C.c_int64),
        ('xpts', C.c_void_p), ('ypts', C.c_void_p), ('lat', C.c_void_p), ('lon', C.c_void_p),
        ('xyz', C.c_void_p), ('los', C.c_void_p), ('inc', C.c_void_p), ('hd', C.c_void_p),
        ('inc0', C.c_double), ('hd0', C.c_double), ('loc', C.c_int32), ('_pad', C.c_int32),
        ('hts', C.c_void_p),
    ]


class NoLevels(Exception):
    """build_ray would return (None, None, None) (losreader.py:832-833)."""


class DeviceOutOfMemory(MemoryError):
    """RDR_ERR_OOM: the GPU could not hold an allocation of the call (the caller may retry with a smaller batch)."""


_lib = None
_lock = threading.Lock()

# every symbol include/raider_hip.h declares: (name, restype, argtypes)
_VP = C.c_void_p
SYMBOLS = [
    ('rdr_version', C.c_int, []),
    ('rdr_source_hash', C.c_char_p, []),
    ('rdr_create', C.c_int, [C.c_int, C.POINTER(_VP)]),
    ('rdr_destroy', None, [_VP]),
    ('rdr_last_error', C.c_char_p, [_VP]),
    ('rdr_set_stream', C.c_int, [_VP, _VP]),
    ('rdr_forget_stream', C.c_int, [_VP, _VP]),
    ('rdr_synchronize', C.c_int, [_VP]),
    ('rdr_device_info', C.c_int, [_VP, C.c_char_p, C.c_int, C.POINTER(C.c_int), c_lp]),
    ('rdr_set_profiling', C.c_int, [_VP, C.c_int]),
    ('rdr_host_alloc', C.c_int, [C.c_int64, C.POINTER(_VP)]),
    ('rdr_host_free', C.c_int, [_VP]),
    ('rdr_set_workspace_limit', C.c_int, [_VP, C.c_int64]),
    ('rdr_set_side_capacity', C.c_int, [_VP, C.c_int64]),
    ('rdr_trim', C.c_int, [_VP, C.c_int64, c_lp]),
    ('rdr_generic_ray_count', C.c_int64, [_VP]),
    ('rdr_profile_get', C.c_int, [_VP, C.c_int, C.POINTER(C.c_int), C.POINTER(C.c_float)]),
    ('rdr_clock_sample_begin', C.c_int, [_VP, C.c_double]),
    ('rdr_clock_sample_end', C.c_int, [_VP, c_dp]),
    ('rdr_ray_kernel_attributes', C.c_int, [_VP, _VP, C.c_int, c_ip, c_ip, c_ip, c_ip, c_ip]),
    ('rdr_cube_create', C.c_int, [_VP, _VP, C.c_int64, _VP, C.c_int64, _VP, C.c_int64, _VP, _VP, C.c_int,
                                  C.c_int64, C.c_int64, C.c_int64, C.c_int, C.POINTER(_VP)]),
    ('rdr_cube_destroy', None, [_VP]),
    ('rdr_cube_has_nan', C.c_int, [_VP]),
    ('rdr_cube_shape', C.c_int, [_VP, c_lp, c_lp, c_lp, C.POINTER(C.c_int)]),
    ('rdr_cube_axes', C.c_int, [_VP, _VP, _VP, _VP]),
    ('rdr_cube_set_projection', C.c_int, [_VP, C.c_int, _VP, C.c_int]),
    ('rdr_cube_view', C.c_int, [_VP, _VP, C.c_int, _VP, C.c_int, C.POINTER(_VP)]),
    ('rdr_project_points', C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _VP, _VP, C.c_int]),
    ('rdr_transform_tm', C.c_int, [_VP, _VP, C.c_int, C.c_int, _VP, _VP, C.c_int64, _VP, _VP, C.c_int]),
    ('rdr_transform_cone', C.c_int, [_VP, C.c_int, _VP, C.c_int, C.c_int, _VP, _VP, C.c_int64, _VP, _VP, C.c_int]),
    ('rdr_cube_blend', C.c_int, [_VP, _VP, C.c_double, _VP, C.c_double, C.POINTER(_VP)]),
    ('rdr_cube_read', C.c_int, [_VP, _VP, _VP, _VP]),
    ('rdr_cube_point_index', C.c_int, [_VP, _VP, C.c_int]),
    ('rdr_cube_point_index_bytes', C.c_int64, [_VP]),
    ('rdr_interp3', C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, _VP, C.c_int]),
    ('rdr_interp3_project', C.c_int, [_VP, _VP, _VP, _VP, _VP, C.c_int64, C.c_int, _VP, C.c_double, _VP, _VP, C.c_int]),
    ('rdr_interp3_blend', C.c_int, [_VP, _VP, C.c_double, _VP, C.c_double, _VP, _VP, _VP, C.c_int64, _VP, _VP, C.c_int]),
    ('rdr_interp3_blend_cube', C.c_int, [_VP, _VP, C.c_double, _VP, C.c_double, _VP, _VP, _VP, C.c_int64, _VP, _VP, C.c_int]),
    ('rdr_build_cube', C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, C.c_int64, _VP, C.c_int64, _VP, _VP, C.c_int]),
    ('rdr_build_cube_to_cube', C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, C.c_int64, _VP, C.c_int64, C.c_int, C.POINTER(_VP)]),
    ('rdr_last_nan_output', C.c_int, [_VP]),
    ('rdr_point_delays', C.c_int, [_VP, _VP, _VP, C.c_int64, _VP, C.c_int64, _VP, C.c_int64, _VP, _VP, _VP, C.c_int64, C.c_int, _VP, C.c_double, _VP, _VP,
                                   c_ip]),
    ('rdr_project_cosinc', C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int]),
    ('rdr_project_divide', C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, C.c_int]),
    ('rdr_ray_levels', C.c_int, [_VP, C.c_double, C.c_double, c_ip, _VP, _VP, _VP]),
    ('rdr_ray_prepass', C.c_int, [_VP, _VP, C.POINTER(RdrRays), C.c_double, C.c_double, _VP, c_ip]),
    ('rdr_nparts', C.c_int, [_VP, C.c_int32, C.c_double, _VP]),
    ('rdr_ray_march', C.c_int, [_VP, _VP, C.POINTER(RdrRays), C.c_double, C.c_double, _VP, C.c_int32, _VP, _VP]),
    ('rdr_raytrace', C.c_int, [_VP, _VP, C.POINTER(RdrRays), C.c_double, C.c_double, C.c_double, _VP, _VP, _VP, c_ip]),
    ('rdr_raytrace_slices', C.c_int, [_VP, _VP, C.POINTER(RdrRays), _VP, C.c_int32, C.c_int32, C.c_double, C.c_double, _VP, _VP, _VP, _VP, C.c_int32, _VP]),
    ('rdr_raytrace_slices_to_cube', C.c_int, [_VP, _VP, C.POINTER(RdrRays), _VP, C.c_int32, C.c_int32, C.c_double, C.c_double, _VP, _VP, C.c_int32, _VP,
                                              C.POINTER(_VP)]),
    ('rdr_top_of_atmosphere', C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_double, _VP, _VP, C.c_int]),
    ('rdr_build_ray', C.c_int, [_VP, _VP, C.c_int64, C.c_double, _VP, _VP, C.c_int64, C.c_double, c_ip, _VP, _VP, _VP, C.c_int]),
    ('rdr_lla2ecef', C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _VP, C.c_int]),
    ('rdr_ecef2lla', C.c_int, [_VP, _VP, C.c_int64, _VP, _VP, _VP, C.c_int]),
    ('rdr_look_vectors', C.c_int, [_VP, C.POINTER(RdrRays), C.c_double, _VP]),
    ('rdr_ray_prepass_device', C.c_int, [_VP, _VP, _VP, C.c_double, C.c_double, _VP]),
    ('rdr_ray_march_device', C.c_int, [_VP, _VP, _VP, C.c_double, C.c_double, C.c_double, _VP, _VP, _VP]),
    ('rdr_inverse_time_weights', C.c_int, [_VP, _VP, C.c_int64, _VP, C.c_int32, C.c_double, C.c_double, _VP, C.c_int]),
    ('rdr_cube_blend_weighted', C.c_int, [_VP, C.POINTER(_VP), C.c_int32, _VP, C.c_int, C.POINTER(_VP)]),
    ('rdr_delays_to_phase', C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int, C.c_double, _VP, _VP, C.c_int]),
    ('rdr_ecmwf_model_levels', C.c_int, [_VP, _VP, _VP, _VP, _VP, _VP, _VP, _VP, C.c_int32, C.c_int64, C.c_int64, C.c_double, _VP, _VP, C.c_int]),
    ('rdr_cubes_from_model_levels', C.c_int, [_VP, _VP, C.c_int64, _VP, C.c_int64, _VP, _VP, _VP, _VP, C.c_int, C.c_int64, _VP, C.c_int64,
                                              C.c_double, C.c_double, C.c_double, C.c_double, C.c_int, C.POINTER(_VP), C.POINTER(_VP), _VP, _VP, _VP]),
    ('rdr_orbit_look_vectors', C.c_int, [_VP, _VP, _VP, _VP, C.c_int64, _VP, C.c_int64, C.c_double, C.c_int, _VP, _VP, _VP, C.c_int]),
    ('rdr_interp_nd', C.c_int, [_VP, C.c_int32, C.POINTER(_VP), c_lp, _VP, _VP, C.c_int64, C.c_int, C.c_double, _VP, C.c_int]),
    ('rdr_interp_along_axis', C.c_int, [_VP, _VP, _VP, C.c_int64, C.c_int64, _VP, C.c_int64, C.c_int, C.c_double, _VP, C.c_int]),
    ('rdr_make_points_count', C.c_int64, [C.c_double, C.c_double]),
    ('rdr_make_points', C.c_int, [_VP, C.c_double, _VP, _VP, C.c_int64, C.c_double, _VP, C.c_int]),
]


def _preload_hip_runtime():
    """PyTorch-ROCm wheels bundle their own libamdhip64.so (SONAME libamdhip64.so.7, the same as /opt/rocm's).  Two HIP
    runtimes in one process do not work ("No HIP GPUs are available" from whichever initialises second), and which
    one libraider_hip.so binds to would depend on import order.  So when torch is installed but not imported yet,
    load ITS copy first: our NEEDED libamdhip64.so.7 then resolves to it, and a later `import torch` finds the same
    file already mapped.  RAIDER_HIP_RUNTIME=system keeps the system runtime, RAIDER_HIP_RUNTIME=/path/lib.so forces one."""
    import importlib.util
    import sys
    choice = os.environ.get('RAIDER_HIP_RUNTIME', '')
    if choice == 'system' or 'torch' in sys.modules:
        return
    if choice:
        C.CDLL(choice, mode=C.RTLD_GLOBAL)
        return
    try:
        spec = importlib.util.find_spec('torch')
    except (ImportError, ValueError):
        spec = None
    if spec and spec.submodule_search_locations:
        cand = Path(list(spec.submodule_search_locations)[0]) / 'lib' / 'libamdhip64.so'
        if cand.exists():
            C.CDLL(str(cand), mode=C.RTLD_GLOBAL)


def source_files():
    """The files libraider_hip.so is compiled from: ONE translation unit (csrc/raider_hip.hip) including every header beside it
    and the public C header."""
    return sorted(f for f in (_HERE / 'csrc').glob('*') if f.suffix in ('.h', '.hip')) + [_HERE.parent / 'include' / 'raider_hip.h']


def source_hash():
    """sha256[:16] over the library's sources as they are in the tree.  The build recipe compiles it into the binary
    (rdr_source_hash), profiles/ digests carry it: a stale binary or a stale counter digest cannot pass for the current code."""
    import hashlib
    h = hashlib.sha256()
    for f in source_files():
        h.update(f.name.encode()); h.update(f.read_bytes())
    return h.hexdigest()[:16]


def binary_source_hash(path=None):
    """The digest compiled into a built library (None: missing file / a build without rdr_source_hash)."""
    path = Path(path or LIB_PATH)
    if not path.exists():
        return None
    if _lib is not None and path == LIB_PATH:
        return _lib.rdr_source_hash().decode()
    # read it without dlopen-ing a possibly stale library into this process: the string sits in .rodata behind a marker
    data = path.read_bytes()
    k = data.find(b'rdr-source-hash:')
    if k < 0:
        return None
    return data[k + 16:k + 32].decode(errors='replace')


def load():
    """dlopen the HIP library (does not need a GPU) and declare every prototype."""
    global _lib
    with _lock:
        if _lib is not None:
            return _lib
        if not LIB_PATH.exists():
            raise RuntimeError(
                f'raider_amd: HIP library {LIB_PATH} is missing - build it with '
                f'`python -c "import __graft_entry__ as g; g.build()"` (hipcc --offload-arch=gfx950). '
                'There is no CPU fallback.')
        _preload_hip_runtime()
        lib = C.CDLL(str(LIB_PATH))
        for name, res, args in SYMBOLS:
            fn = getattr(lib, name)      # AttributeError if the .so does not export a declared symbol
            fn.restype = res
            fn.argtypes = args
        _lib = lib
        return lib


def last_error(ctx=None):
    msg = load().rdr_last_error(ctx)
    return msg.decode() if msg else ''


def check(rc, ctx=None, exc_invalid=ValueError):
    """Map C status codes onto the exception classes the reference raises."""
    if rc == RDR_OK:
        return
    msg = last_error(None) or last_error(ctx)      # (the calling thread's own message: the context's may be another thread's by now)
    if rc == RDR_ERR_INVALID:
        raise exc_invalid(msg)
    if rc == RDR_ERR_ALL_NAN:
        raise ValueError(msg)                      # delay.py:279-280
    if rc == RDR_ERR_NAN_LENGTH:
        raise ValueError(msg)                      # delay.py:283 -> np.linspace(num<0) ValueError in the reference
    if rc == RDR_ERR_NO_LEVELS:
        raise NoLevels(msg)
    if rc == RDR_ERR_OOM:
        raise DeviceOutOfMemory(f'raider_amd HIP engine: out of device memory: {msg}')
    raise RuntimeError(f'raider_amd HIP engine error {rc}: {msg}')


class _SerialisedLib:
    """The library as ONE context sees it: every call through it holds the context's lock.  A rdr_ctx is not thread-safe
    (include/raider_hip.h: scratch slots, workspace, stream are per context) and ctypes drops the GIL around a call, so two Python
    threads sharing a context - the default one, typically - would otherwise run inside the same context at once.  Distinct
    contexts never wait for each other."""

    def __init__(self, lib, lock):
        self.__dict__['_lib'] = lib
        self.__dict__['_lock'] = lock

    def __getattr__(self, name):
        fn = getattr(self._lib, name)
        lock = self._lock

        def call(*args):
            with lock:
                return fn(*args)
        call.__name__ = name
        self.__dict__[name] = call          # (next time: a plain attribute hit)
        return call


class Context:
    """One device + one stream.  `Context.default()` is the process-wide lazily created context.  Calls from several threads into one
    context are serialised (a lock per context); use one context per thread to run them side by side."""

    _default = None
    _default_lock = threading.Lock()
    _serials = __import__('itertools').count(1)

    def __init__(self, device=-1):
        lib = load()
        self.serial = next(Context._serials)      # never reused (id() of a collected context can be): what caches key contexts by
        h = C.c_void_p()
        rc = lib.rdr_create(int(device), C.byref(h))
        if rc != RDR_OK:
            raise RuntimeError(f'raider_amd: cannot create a GPU context ({last_error()}). '
                               'raider_amd needs an AMD MI355X (gfx950); there is no CPU fallback.')
        self.handle = h
        self.lock = threading.RLock()
        self.lib = _SerialisedLib(lib, self.lock)

    @classmethod
    def default(cls):
        if cls._default is None:
            with cls._default_lock:
                if cls._default is None:
                    dev = int(os.environ.get('RAIDER_HIP_DEVICE', os.environ.get('LOCAL_RANK', '-1')))
                    cls._default = cls(dev)
        return cls._default

    def set_stream(self, stream_handle):
        """Launch on an external HIP stream (0 / None = HIP's default stream, -1 = the context's private stream)."""
        h = -1 if stream_handle == -1 else (stream_handle or 0)
        check(self.lib.rdr_set_stream(self.handle, C.c_void_p(h)), self.handle)

    def forget_stream(self, stream_handle):
        """Before DESTROYING a stream once handed to set_stream: the context finishes its work on it and never records an event on it again
        (rdr_forget_stream).  Streams that live as long as the process - torch's - need no call."""
        check(self.lib.rdr_forget_stream(self.handle, C.c_void_p(stream_handle or 0)), self.handle)

    def adopt_torch_stream(self, tensor):
        """Order this context's kernels with torch work on `tensor`'s device: launch on torch's current stream."""
        import torch
        self.set_stream(torch.cuda.current_stream(tensor.device).cuda_stream)

    def synchronize(self):
        check(self.lib.rdr_synchronize(self.handle), self.handle)

    def device_info(self):
        name = C.create_string_buffer(256)
        cus = C.c_int()
        mem = C.c_int64()
        check(self.lib.rdr_device_info(self.handle, name, 256, C.byref(cus), C.byref(mem)), self.handle)
        return name.value.decode(), cus.value, mem.value

    def set_profiling(self, on=True):
        check(self.lib.rdr_set_profiling(self.handle, int(bool(on))), self.handle)

    def set_workspace_limit(self, nbytes):
        """Cap the HBM workspace that hands ray records from ray pass 1 to pass 2 (bigger batches run in chunks)."""
        check(self.lib.rdr_set_workspace_limit(self.handle, int(nbytes)), self.handle)

    def trim(self, keep_bytes=0):
        """Give device memory back (rdr_trim): scratch buffers larger than `keep_bytes` and pooled cube buffers beyond `keep_bytes` in
        total are freed after the context's streams have drained.  Returns the bytes released."""
        out = C.c_int64(0)
        check(self.lib.rdr_trim(self.handle, int(keep_bytes), C.byref(out)), self.handle)
        return int(out.value)

    def set_side_capacity(self, columns=-1):
        """Capacity (rays) of the side buffer holding the level crossings of generic-geodesy rays; -1 = automatic."""
        check(self.lib.rdr_set_side_capacity(self.handle, int(columns)), self.handle)

    def generic_ray_count(self):
        """Rays the last synchronising ray pass 1 left to the generic-geodesy kernels (diagnostics)."""
        return int(self.lib.rdr_generic_ray_count(self.handle))

    def clock_sample(self, ms):
        """Start sampling the shader clock for `ms` milliseconds of wall time (one sleeping wave on the copy stream); returns a
        function that waits for the sample and gives the clock in GHz - call it after the kernels to be observed."""
        check(self.lib.rdr_clock_sample_begin(self.handle, float(ms)), self.handle)

        def end():
            g = C.c_double()
            check(self.lib.rdr_clock_sample_end(self.handle, C.byref(g)), self.handle)
            return float(g.value)
        return end

    def profile_get(self, which):
        """(launch count, total ms) of kernel kind `which` since set_profiling(True)."""
        n = C.c_int()
        ms = C.c_float()
        check(self.lib.rdr_profile_get(self.handle, int(which), C.byref(n), C.byref(ms)), self.handle)
        return n.value, float(ms.value)

    def __del__(self):
        try:
            if getattr(self, 'handle', None) and self is not Context._default:
                self.lib.rdr_destroy(self.handle)
        except Exception:
            pass


def f64(a):
    """C-contiguous float64 view/copy (what py::array::c_style forces, module.cpp:27-29)."""
    return np.ascontiguousarray(a, dtype=np.float64)


def ptr(a):
    """void* of a NumPy array, a torch tensor (data_ptr), an int address, or None."""
    if a is None:
        return None
    if isinstance(a, np.ndarray):
        return a.ctypes.data_as(C.c_void_p)
    if isinstance(a, int):
        return C.c_void_p(a)
    if hasattr(a, 'data_ptr'):
        return C.c_void_p(a.data_ptr())
    raise TypeError(f'cannot take a pointer of {type(a)}')
