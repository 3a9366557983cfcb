"""Orbit state vectors for ray tracing: file readers + the device look-vector solver.

Readers mirror tools/RAiDER/losreader.py: `read_txt_file` (:429-475), `read_ESA_Orbit_file` (:478-518), `get_sv`
(:319-371), `cut_times` (:617-634), `filter_ESA_orbit_file` (:536-553).  `Orbit` plays isce3.core.Orbit for the one
thing the delay path needs from it (get_orbit, :736-769: sort by time, drop duplicates) and `Orbit.look_vectors`
replaces the isce3 geo2rdr + interpolate per-pixel loop (:219-255) with one kernel launch.
"""
import datetime as dt
import os
import xml.etree.ElementTree as ET
from pathlib import Path

import numpy as np

from . import _lib as L
from ._lib import Context, check, f64, ptr


def read_txt_file(filename):
    """losreader.py:429-475: 7 columns `isotime x y z vx vy vz`; returns [t, x, y, z, vx, vy, vz]."""
    cols = [[] for _ in range(7)]
    with open(filename) as f:
        for line in f:
            try:
                parts = line.strip().split()
                vals = [dt.datetime.fromisoformat(parts[0])] + [float(p) for p in parts[1:]]
                if len(vals) != 7:
                    raise ValueError
            except (ValueError, IndexError):
                raise ValueError(f'I need {filename} to be a 7 column text file, with columns t, x, y, z, vx, vy, vz '
                                 f"(Couldn't parse line {repr(line)})")
            for c, v in zip(cols, vals):
                c.append(v)
    if len(cols[0]) < 4:
        raise ValueError(f'read_txt_file: File {filename} does not have enough statevectors')
    return [np.array(c) for c in cols]


def read_ESA_Orbit_file(filename):
    """losreader.py:478-518: ESA .EOF orbit XML -> [t (datetimes), x, y, z, vx, vy, vz]."""
    root = ET.parse(filename).getroot()
    osvs = root[1][0]
    n = len(osvs)
    t = []
    arr = np.ones((6, n))
    for i, st in enumerate(osvs):
        t.append(dt.datetime.strptime(st[1].text, 'UTC=%Y-%m-%dT%H:%M:%S.%f'))
        for k in range(6):
            arr[k, i] = float(st[4 + k].text)
    return [np.array(t)] + [arr[k] for k in range(6)]


def filter_ESA_orbit_file(orbit_xml, ref_time):
    """losreader.py:536-553: does the validity window in the file NAME contain ref_time?"""
    f = os.path.basename(orbit_xml)
    t0 = dt.datetime.strptime(f.split('_')[6].lstrip('V'), '%Y%m%dT%H%M%S')
    t1 = dt.datetime.strptime(f.split('_')[7].rstrip('.EOF'), '%Y%m%dT%H%M%S')
    return t0 < ref_time < t1


def pick_ESA_orbit_file(list_files, ref_time):
    """losreader.py:521-534: the first .EOF file of the list whose validity window (in its NAME) contains ref_time."""
    for path in list_files:
        if filter_ESA_orbit_file(path, ref_time):
            return path
    raise AssertionError('Given orbit files did not match given date/time')


def cut_times(times, ref_time, pad):
    """losreader.py:617-634."""
    diff = np.array([(x - ref_time).total_seconds() for x in times])
    return np.abs(diff) < pad


def read_shelve(filename):
    """losreader.py:399-426: the state vectors of an isce2 shelve (`shelve.open(filename)['frame'].orbit.stateVectors`, each with
    .time / .position / .velocity).  Unpickling the frame needs the classes it was pickled from - isce2's, when isce2 wrote it."""
    import shelve
    with shelve.open(str(filename), 'r') as db:
        frame = db['frame']
    vectors = list(frame.orbit.stateVectors)
    if not vectors:
        raise ValueError('read_shelve: the file has not statevectors')
    t = np.array([sv.time for sv in vectors])
    pos = np.array([[float(v) for v in sv.position[:3]] for sv in vectors], dtype=np.float64)
    vel = np.array([[float(v) for v in sv.velocity[:3]] for sv in vectors], dtype=np.float64)
    return t, pos[:, 0].copy(), pos[:, 1].copy(), pos[:, 2].copy(), vel[:, 0].copy(), vel[:, 1].copy(), vel[:, 2].copy()


def get_sv(los_file, ref_time, pad):
    """losreader.py:319-371: a 7-column text file, else ESA orbit file(s), else an isce2 shelve."""
    try:
        svs = read_txt_file(los_file)
    except (ValueError, TypeError, OSError, IsADirectoryError):
        try:
            files = [los_file] if isinstance(los_file, (str, Path)) else list(los_file)
            files = sorted(set(str(f) for f in files))
            try:
                files = [f for f in files if filter_ESA_orbit_file(f, ref_time)] or files
            except (IndexError, ValueError, TypeError):
                pass
            if not files:
                raise ValueError('There are no valid orbit files provided')
            parts = [read_ESA_Orbit_file(f) for f in files]
            svs = [np.concatenate([p[k] for p in parts]) for k in range(7)]
        except Exception:
            try:
                svs = list(read_shelve(los_file))
            except (ImportError, AttributeError) as exc:      # a shelve all right, but its pickled classes cannot be rebuilt here
                raise ValueError(f'get_sv: {los_file} looks like an isce2 shelve, but its objects cannot be unpickled without the package '
                                 f'that wrote them ({exc})')
            except Exception:
                raise ValueError(f'get_sv: I cannot parse the statevector file {los_file}')
    if ref_time:
        idx = cut_times(svs[0], ref_time, pad=pad)
        svs = [d[idx] for d in svs]
    return svs


class Orbit:
    """Time-sorted, de-duplicated state vectors (what get_orbit hands to isce3, losreader.py:736-769)."""

    def __init__(self, times, pos, vel, epoch=None):
        times = list(times)
        self.epoch = epoch or min(times)
        t = np.array([(x - self.epoch).total_seconds() if isinstance(x, dt.datetime) else float(x) for x in times])
        order = np.argsort(t, kind='stable')
        t, pos, vel = t[order], np.asarray(pos, dtype=np.float64)[order], np.asarray(vel, dtype=np.float64)[order]
        keep = np.concatenate([[True], np.diff(t) > 0])
        self.time, self.position, self.velocity = f64(t[keep]), f64(pos[keep]), f64(vel[keep])
        if self.time.size < 4:
            raise RuntimeError('state_to_los: At least 4 state vectors are required for orbit interpolation')

    @classmethod
    def from_file(cls, orbit_file, ref_time, pad=600):
        t, x, y, z, vx, vy, vz = get_sv(orbit_file, ref_time, pad)
        return cls(t, np.stack([x, y, z], -1), np.stack([vx, vy, vz], -1))

    def look_vectors(self, xyz, threshold=1.0e-7, maxiter=30, return_geometry=False, ctx=None):
        """Unit ECEF vectors target->sensor at zero Doppler for targets xyz[...,3] (NumPy, or a torch tensor on the
        GPU); NaN where the solve fails (losreader.py:253-254)."""
        ctx = ctx or Context.default()
        dev = hasattr(xyz, 'data_ptr')
        if dev:
            import torch
            ctx.adopt_torch_stream(xyz)
            shp = tuple(xyz.shape[:-1]); n = xyz.numel() // 3
            los = torch.empty(shp + (3,), dtype=torch.float64, device=xyz.device)
            az = torch.empty(shp, dtype=torch.float64, device=xyz.device) if return_geometry else None
            rg = torch.empty(shp, dtype=torch.float64, device=xyz.device) if return_geometry else None
            x = xyz
        else:
            xyz = np.asarray(xyz, dtype=np.float64)
            shp = xyz.shape[:-1]
            x = f64(xyz).reshape(-1, 3); n = x.shape[0]
            los = np.empty((n, 3)); az = np.empty(n) if return_geometry else None; rg = np.empty(n) if return_geometry else None
        check(ctx.lib.rdr_orbit_look_vectors(ctx.handle, ptr(self.time), ptr(self.position), ptr(self.velocity), self.time.size,
                                             ptr(x), n, float(threshold), int(maxiter), ptr(los), ptr(az), ptr(rg),
                                             L.RDR_DEVICE if dev else L.RDR_HOST), ctx.handle, RuntimeError)
        if not dev:
            los = los.reshape(shp + (3,))
            if return_geometry:
                az, rg = az.reshape(shp), rg.reshape(shp)
        return (los, az, rg) if return_geometry else los

    def direction(self):
        """losreader.py:200-207 getSensorDirection."""
        return 'desc' if self.position[0, 2] > self.position[-1, 2] else 'asc'
