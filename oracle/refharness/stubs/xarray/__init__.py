"""Build-owned minimal stand-in for `xarray` (oracle harness only; never shipped in the product).

Only what RAiDER's delay path needs to be *importable* plus a tiny Dataset that
`getInterpolators` / `tropo_delay` can read (`.variables[name][:]`, `.z.values`, `['proj'].attrs`).
"""
import numpy as np


class _Var:
    def __init__(self, data, attrs=None):
        self._data = np.asarray(data)
        self.attrs = dict(attrs or {})

    def __getitem__(self, key):
        return self._data[key]

    def __array__(self, dtype=None, copy=None):
        return self._data if dtype is None else self._data.astype(dtype)

    @property
    def values(self):
        return self._data

    def diff(self, dim=None):
        return _Var(np.diff(self._data))

    def mean(self):
        return self._data.mean()


class Dataset:
    def __init__(self, data_vars=None, coords=None, attrs=None):
        self.variables = {}
        self.attrs = dict(attrs or {})
        for src in (coords or {}), (data_vars or {}):
            for k, v in src.items():
                if isinstance(v, tuple):
                    data = v[1]
                    a = v[2] if len(v) > 2 else None
                else:
                    data, a = v, None
                self.variables[k] = _Var(data, a)

    def __getitem__(self, k):
        return self.variables[k]

    def __setitem__(self, k, v):
        self.variables[k] = v if isinstance(v, _Var) else _Var(v)

    def __getattr__(self, k):
        try:
            return self.__dict__['variables'][k]
        except KeyError:
            raise AttributeError(k)

    def __enter__(self):
        return self

    def __exit__(self, *a):
        return False


_REGISTRY = {}


def register_dataset(path, ds):
    _REGISTRY[str(path)] = ds


def load_dataset(path, *a, **k):
    return _REGISTRY[str(path)]


open_dataset = load_dataset
DataArray = _Var
