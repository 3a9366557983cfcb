#!/usr/bin/env python3
"""Randomised parity sweep: GPU ray tracer vs the NumPy oracle over random cubes (size, extent, model top, uniform / jittered /
stretched axes), scenes (anywhere on the globe short of the poles and the date line, partly outside the cube), heights,
integration tops, incidence angles up to 70 degrees and segment lengths.  Not part of the test suite (minutes of CPU for the
oracle): a tool to shake out rare-path bugs.   usage: fuzz_parity.py [ntrials=200] [seed=0]"""
import json
import sys
from pathlib import Path

import numpy as np

sys.path.insert(0, str(Path(__file__).resolve().parent.parent))
import raider_amd as R                      # noqa: E402
from oracle import raider_oracle as O       # noqa: E402

ntrials = int(sys.argv[1]) if len(sys.argv) > 1 else 200
seed = int(sys.argv[2]) if len(sys.argv) > 2 else 0
rng = np.random.default_rng(seed)
rng2 = np.random.default_rng(seed + 1_000_003)
import os as _os
MAX_INC = float(_os.environ.get('FUZZ_MAX_INC', '70'))       # incidence range of the per-pixel look vectors (default: the 0-70 deg of rounds 1-5)
SHORT_SEG = _os.environ.get('FUZZ_SHORT_SEG', '') not in ('', '0')
NAN_CUBE = _os.environ.get('FUZZ_NAN_CUBE', '') not in ('', '0')
TINY_NZ = _os.environ.get('FUZZ_TINY_NZ', '') not in ('', '0')
DESC_ZX = _os.environ.get('FUZZ_DESC_ZX', '') not in ('', '0')
worst = dict(wet=0.0, hydro=0.0)
bad = []
stats = dict(trials=0, lcc_trials=0, stere_trials=0, dateline_trials=0, all_nan_slices=0, no_level_slices=0, nan_rays=0, rays=0)
for trial in range(ntrials):
    ny, nx, nz = int(rng.integers(6, 50)), int(rng.integers(6, 50)), int(rng.integers(5, 48))
    if TINY_NZ and rng2.random() < 0.5:          # (round 6) two to four model levels: the z windows of the marcher at their limits
        nz = int(rng2.integers(2, 5))
    lat_c = rng.uniform(-80, 80); lon_c = rng.uniform(-160, 160) if rng.random() < 0.85 else rng.choice([-1.0, 1.0]) * rng.uniform(170.0, 179.5)   # incl. the date line
    dlat = rng.uniform(1.5, 8.0); dlon = min(rng.uniform(1.5, 8.0) / max(np.cos(np.radians(lat_c)), 0.2), 30.0)
    dlon = min(dlon, 180.0 - abs(lon_c) - 1e-3)                 # a lon/lat cube's axis stays inside [-180, 180]
    ztop = float(rng.choice([15000.0, 26000.0, 41000.0, 80000.0]))
    c = O.synthetic_cube(ny, nx, nz, seed=int(rng.integers(1 << 30)), ztop=ztop, y0=lat_c - dlat, y1=lat_c + dlat, x0=lon_c - dlon, x1=lon_c + dlon)
    axes_kind = rng.choice(['exact', 'jitter', 'stretch'])
    proj = None
    u = rng.random()
    if u >= 0.3 and u < 0.38 and abs(lat_c) > 40:      # polar-stereographic model grid (HRRR-AK's spherical one, or an ellipsoidal one), either pole
        south = lat_c < 0
        proj = dict(proj='stere', lat_0=-90.0 if south else 90.0, lat_ts=(-1 if south else 1) * float(rng.choice([60.0, 70.0, 71.0])) if rng.random() < 0.8 else None,
                    k_0=0.994, lon_0=float(rng.uniform(-180, 180)), x_0=0.0, y_0=0.0, a=6371229.0 if u < 0.35 else 6378137.0, es=0.0 if u < 0.35 else 0.0066943799901413165)
        kw = {k: v for k, v in proj.items() if k != 'proj'}
        la, lo = np.meshgrid(np.linspace(lat_c - dlat, lat_c + dlat, 9), np.linspace(lon_c - dlon, lon_c + dlon, 9), indexing='ij')
        px, py = O.stere_forward(la, lo, **kw)
        c['xs'] = np.linspace(px.min(), px.max(), nx); c['ys'] = np.linspace(py.min(), py.max(), ny)
        axes_kind = 'exact'
    if u < 0.3 and abs(lat_c) < 70:      # Lambert-conformal-conic model grid (spherical cone as HRRR's, or an ellipsoidal one)
        lat1 = float(np.clip(lat_c + rng.uniform(-6, 6), -75, 75)); lat2 = lat1 if rng.random() < 0.5 else float(np.clip(lat1 + rng.uniform(2, 12) * np.sign(lat1 or 1.0), -80, 80))
        proj = dict(lat_1=lat1, lat_2=lat2, lat_0=float(lat_c + rng.uniform(-3, 3)), lon_0=float(lon_c + rng.uniform(-15, 15)), x_0=float(rng.choice([0.0, 5.0e5])), y_0=0.0,
                    a=6371229.0 if u < 0.22 else 6378137.0, es=0.0 if u < 0.22 else 0.0066943799901413165)
        la, lo = np.meshgrid(np.linspace(lat_c - dlat, lat_c + dlat, 9), np.linspace(lon_c - dlon, lon_c + dlon, 9), indexing='ij')
        px, py = O.lcc_forward(la, lo, **proj)
        proj = dict(proj, proj='lcc')
        c['xs'] = np.linspace(px.min(), px.max(), nx); c['ys'] = np.linspace(py.min(), py.max(), ny)
        axes_kind = 'exact'
    if proj is not None:
        pass
    elif axes_kind == 'jitter':
        c['xs'] = c['xs'] + 1e-7 * rng.uniform(-1, 1, nx); c['ys'] = c['ys'] + 1e-7 * rng.uniform(-1, 1, ny)
    elif axes_kind == 'stretch':
        c['xs'] = c['xs'][0] + (c['xs'] - c['xs'][0]) * (1 + 0.15 * np.linspace(0, 1, nx)); c['ys'] = c['ys'][0] + (c['ys'] - c['ys'][0]) * (1 + 0.1 * np.linspace(0, 1, ny))
    if proj is None and rng.random() < 0.25:          # latitude axis stored north to south (scipy's RGI flips descending axes, _rgi.py:280-281)
        c['ys'] = c['ys'][::-1].copy(); c['wet'] = c['wet'][:, ::-1, :].copy(); c['hydro'] = c['hydro'][:, ::-1, :].copy()
        axes_kind = str(axes_kind) + '+descending_y'
    if DESC_ZX and proj is None and rng2.random() < 0.4:       # (round 6) x and / or z axes stored descending too (scipy's RGI flips them, _rgi.py `_check_points`; rdr_cube_create does)
        if rng2.random() < 0.6:
            c['zs'] = c['zs'][::-1].copy(); c['wet'] = c['wet'][::-1].copy(); c['hydro'] = c['hydro'][::-1].copy()
            axes_kind = str(axes_kind) + '+descending_z'
        if rng2.random() < 0.6:
            c['xs'] = c['xs'][::-1].copy(); c['wet'] = c['wet'][:, :, ::-1].copy(); c['hydro'] = c['hydro'][:, :, ::-1].copy()
            axes_kind = str(axes_kind) + '+descending_x'
        stats['descending_zx_trials'] = stats.get('descending_zx_trials', 0) + 1
    if rng.random() < 0.2:           # float64 fields (the double2 instantiations of the ray kernels)
        c['wet'] = c['wet'].astype(np.float64) * (1 + 1e-9 * rng.standard_normal(c['wet'].shape)); c['hydro'] = c['hydro'].astype(np.float64) * (1 + 1e-9 * rng.standard_normal(c['hydro'].shape))
        axes_kind = str(axes_kind) + '+f64'
    gy, gx = int(rng.integers(3, 14)), int(rng.integers(3, 14))
    f = rng.uniform(0.3, 1.15)                                   # > 1: part of the scene starts outside the cube
    ypts = np.linspace(lat_c + f * dlat, lat_c - f * dlat, gy) if rng.random() < 0.7 else np.linspace(lat_c - f * dlat, lat_c + f * dlat, gy)
    xpts = np.linspace(lon_c - f * dlon, lon_c + f * dlon, gx)
    ht = float(rng.choice([0.0, -80.0, 250.0, 1234.5, 3000.0, float(c['zs'][min(3, nz - 1)]), float(c['zs'].max() + 5.0)]))
    zref = float(min(rng.choice([c['zs'].max() - 1, 0.6 * c['zs'].max(), c['zs'].max() + 500.0]), c['zs'].max() - 1))     # delay.py:86-93
    # (round 6) an origin ABOVE zref inside zref's own model interval: the reference's one reversed segment (losreader.py:785-808,821).  Drawn from
    # a second generator so that the trials of earlier seeds stay what they were.
    if rng2.random() < 0.12:
        zasc = np.sort(c['zs'])
        kt = int(np.searchsorted(zasc, zref, side='right'))
        if kt < len(zasc) and zasc[kt] - zref > 3.0:
            ht = float(zref + rng2.uniform(1.2, zasc[kt] - zref - 0.5))
            stats['above_zref_trials'] = stats.get('above_zref_trials', 0) + 1
    max_seg = float(rng.choice([1000.0, 1000.0, 400.0, 2500.0]))
    if SHORT_SEG and rng2.random() < 0.3:
        max_seg = float(rng2.choice([37.0, 90.0, 150.0]))          # many integration points per level (nParts up to ~100)
    inc = rng.uniform(0, MAX_INC, (gy, gx)) if rng.random() < 0.8 else np.full((gy, gx), rng.uniform(15, 50))
    hd = rng.uniform(-180, 180, (gy, gx)) if rng.random() < 0.5 else np.full((gy, gx), -167.9)
    c_finite = None
    nan_los = rng.random() < 0.15
    look = lambda ht_, llh, xyz, yy: O.look_vectors_from_inc_hd(inc, hd, llh[1], llh[0], llh[2])
    xx, yy = np.meshgrid(xpts, ypts)
    los = look(ht, [xx, yy, np.full(yy.shape, ht)], None, yy)
    if nan_los:
        los[rng.random((gy, gx)) < 0.2] = np.nan
    look2 = lambda ht_, llh, xyz, yy_: los
    if NAN_CUBE and rng2.random() < 0.35:        # (round 6) a block of missing values in the model: every sample touching it is NaN in the reference (scipy RGI) and here
        k0_, k1_ = sorted(rng2.integers(0, nz, 2)); j0_, j1_ = sorted(rng2.integers(0, ny, 2)); i0_, i1_ = sorted(rng2.integers(0, nx, 2))
        c_finite = dict(c)                                   # (the model before the hole was cut: the knife-edge test below)
        c['wet'] = c['wet'].copy(); c['hydro'] = c['hydro'].copy()
        if rng2.random() < 0.5:
            c['wet'][k0_:k1_ + 1, j0_:j1_ + 1, i0_:i1_ + 1] = np.nan
        else:
            c['hydro'][k0_:k1_ + 1, j0_:j1_ + 1, i0_:i1_ + 1] = np.nan
        stats['nan_cube_trials'] = stats.get('nan_cube_trials', 0) + 1
    ip = list(O.getInterpolators(c['xs'], c['ys'], c['zs'], c['wet'], c['hydro']))
    cube = R.Cube(c['ys'], c['xs'], c['zs'], c['wet'], c['hydro'], order='zyx')
    if proj is not None and proj['proj'] == 'stere':
        cube.set_projection_stere(**{k: v for k, v in proj.items() if k != 'proj'})
    elif proj is not None:
        cube.set_projection_lcc(**{k: v for k, v in proj.items() if k != 'proj'})
    tag = dict(trial=trial, proj=(None if proj is None else proj['proj'] + ('/sphere' if proj['es'] == 0 else '/ellipsoid')), cube=[ny, nx, nz], ztop=ztop, axes=str(axes_kind), lat=round(lat_c, 2), lon=round(lon_c, 2), ht=ht, zref=zref, max_seg=max_seg, scene=[gy, gx])
    stats['trials'] += 1; stats['lcc_trials'] += int(proj is not None and proj['proj'] == 'lcc'); stats['stere_trials'] += int(proj is not None and proj['proj'] == 'stere')
    stats['dateline_trials'] += int(abs(lon_c) > 169)
    try:
        o_err = None
        (ow, oh), onp = O.build_cube_ray(xpts, ypts, np.array([ht]), look2, ip, MAX_SEGMENT_LENGTH=max_seg, MAX_TROPO_HEIGHT=zref, return_nparts=True, model_proj=proj)
    except Exception as e:           # the reference's own failure modes (all-NaN lengths, no levels on a non-top slice, ...)
        o_err = type(e).__name__
    try:
        g_err = None
        wet, hyd, nparts, _ = cube.raytrace(R.Rays.grid(xpts, ypts, los=np.ascontiguousarray(los)), ht, zref, max_seg)
    except R.NoLevels:
        g_err = 'NoLevels'
    except Exception as e:
        g_err = type(e).__name__
    if o_err or g_err:
        # oracle: build_ray returning None on the (only = last) slice leaves zeros (delay.py:276-277) - the engine raises NoLevels
        if g_err == 'NoLevels' and o_err is None and np.all(ow == 0):
            stats['no_level_slices'] += 1
            continue
        if o_err == 'ValueError' and g_err == 'ValueError':
            stats['all_nan_slices'] += 1
            continue
        bad.append(dict(tag, kind='error mismatch', oracle=o_err, gpu=g_err))
        continue
    stats['rays'] += gy * gx; stats['nan_rays'] += int(np.isnan(oh[0]).sum())
    if not np.array_equal(nparts, onp[0]):
        bad.append(dict(tag, kind='nparts', gpu=nparts.tolist(), oracle=list(map(int, onp[0]))))
        continue
    if _os.environ.get('FUZZ_DUMP_TRIAL') == str(trial):       # everything needed to replay one trial elsewhere
        np.savez(_os.environ.get('FUZZ_DUMP_PATH', 'gpurun_out/fuzz_dump.npz'), wet=wet, hyd=hyd, ow=ow[0], oh=oh[0], los=los, xpts=xpts, ypts=ypts, ht=ht, zref=zref, max_seg=max_seg,
                 ys=c['ys'], xs=c['xs'], zs=c['zs'], cw=c['wet'], ch=c['hydro'], nparts=nparts)
    if not np.array_equal(np.isnan(hyd), np.isnan(oh[0])) or not np.array_equal(np.isnan(wet), np.isnan(ow[0])):
        # A hole in the model: a sample sitting ON a node within rounding (every level top of the lowest levels does - the Newton crossing
        # converges to the last bit there) touches the hole's face with weight 0 or not at all depending on the last bit of its height, and
        # scipy's NaN * 0 = NaN makes the verdict of such a ray a coin toss IN THE REFERENCE ITSELF (its height comes from PROJ's formula).  A
        # pixel whose verdicts differ is excused when, on the model WITHOUT the hole, GPU and oracle agree there and the finite one of the two
        # holed results equals that value: the hole's corners carried no weight.
        excused = False
        if c_finite is not None:
            ipf = list(O.getInterpolators(c_finite['xs'], c_finite['ys'], c_finite['zs'], c_finite['wet'], c_finite['hydro']))
            (fw, fh), _ = O.build_cube_ray(xpts, ypts, np.array([ht]), look2, ipf, MAX_SEGMENT_LENGTH=max_seg, MAX_TROPO_HEIGHT=zref, return_nparts=True, model_proj=proj)
            cf = R.Cube(c_finite['ys'], c_finite['xs'], c_finite['zs'], c_finite['wet'], c_finite['hydro'], order='zyx')
            if proj is not None and proj['proj'] == 'stere':
                cf.set_projection_stere(**{k: v for k, v in proj.items() if k != 'proj'})
            elif proj is not None:
                cf.set_projection_lcc(**{k: v for k, v in proj.items() if k != 'proj'})
            gfw, gfh, _, _ = cf.raytrace(R.Rays.grid(xpts, ypts, los=np.ascontiguousarray(los)), ht, zref, max_seg)
            excused = True
            for g_, o_, gf_, of_ in ((wet, ow[0], gfw, fw[0]), (hyd, oh[0], gfh, fh[0])):
                mm = np.isnan(g_) != np.isnan(o_)
                holed = np.where(np.isnan(g_), o_, g_)                 # the finite one of the two holed results
                ok_ = (np.abs(gf_ - of_) < 2e-8) & (np.abs(holed - of_) < 2e-8)
                excused = excused and bool(ok_[mm].all())
                stats['knife_edge_pixels'] = stats.get('knife_edge_pixels', 0) + int(mm.sum()) * int(excused)
        if excused:
            continue
        bad.append(dict(tag, kind='nan pattern', gpu_nan=int(np.isnan(hyd).sum()), oracle_nan=int(np.isnan(oh[0]).sum())))
        continue
    dw = float(np.nanmax(np.abs(wet - ow[0]))) if np.isfinite(ow[0]).any() else 0.0
    dh = float(np.nanmax(np.abs(hyd - oh[0]))) if np.isfinite(oh[0]).any() else 0.0
    worst['wet'] = max(worst['wet'], dw); worst['hydro'] = max(worst['hydro'], dh)
    # look vectors generated INSIDE the kernels from incidence / heading rasters (inc_hd_to_enu + enu2ecef on the device)
    if not nan_los and trial % 4 == 1:
        iw, ih, inp, _ = cube.raytrace(R.Rays.grid(xpts, ypts, inc=inc, hd=hd), ht, zref, max_seg)
        di = max(float(np.nanmax(np.abs(iw - wet))) if np.isfinite(wet).any() else 0.0, float(np.nanmax(np.abs(ih - hyd))) if np.isfinite(hyd).any() else 0.0)
        worst['inc_hd_mode_vs_vectors'] = max(worst.get('inc_hd_mode_vs_vectors', 0.0), di)
        if not np.array_equal(inp, nparts) or not np.array_equal(np.isnan(ih), np.isnan(hyd)) or di > 2e-9:
            bad.append(dict(tag, kind='inc/heading LOS mode vs look vectors', d=di))
    # the same rays as a POINT list (per-ray lat/lon, or per-ray ECEF origins): different tile mapping, no shared tile trigonometry
    if trial % 3 == 0:
        lo_c = np.ascontiguousarray(los).reshape(-1, 3)
        if trial % 6 == 0:
            rp = R.Rays.points(lat=yy.ravel().copy(), lon=xx.ravel().copy(), los=lo_c)
        else:
            xyz = np.stack(O.lla2ecef(yy.ravel(), xx.ravel(), np.full(yy.size, ht)), -1)
            rp = R.Rays.points(xyz=np.ascontiguousarray(xyz), lat=yy.ravel().copy(), lon=xx.ravel().copy(), los=lo_c)
        pw, ph, pn, _ = cube.raytrace(rp, ht, zref, max_seg)
        stats['point_list_trials'] = stats.get('point_list_trials', 0) + 1
        dp = max(float(np.nanmax(np.abs(pw.reshape(gy, gx) - wet))) if np.isfinite(wet).any() else 0.0,
                 float(np.nanmax(np.abs(ph.reshape(gy, gx) - hyd))) if np.isfinite(hyd).any() else 0.0)
        worst['points_vs_grid'] = max(worst.get('points_vs_grid', 0.0), dp)
        if not np.array_equal(pn, nparts) or not np.array_equal(np.isnan(ph.reshape(gy, gx)), np.isnan(hyd)) or dp > 2e-9:
            bad.append(dict(tag, kind='point list vs grid', d=dp, nparts_equal=bool(np.array_equal(pn, nparts))))
    if max(dw, dh) > 2e-8:
        bad.append(dict(tag, kind='value', d_wet=dw, d_hydro=dh, max_inc=float(inc.max())))
    # the same scene on a DEM: per-pixel origin heights (rdr_rays.hts) against the oracle's per-ray restatement (lon/lat cubes:
    # the C oracle has no projected-model branch); every fourth of these with all heights equal = the slice result, bit for bit
    if proj is None and not nan_los and trial % 2 == 0 and 'descending_z' not in str(axes_kind) and 'descending_x' not in str(axes_kind):      # (the C oracle takes ascending x / z)
        from oracle import oracle_c as OC
        equal = trial % 8 == 0
        hts = np.full((gy, gx), ht) if equal else ht + rng.uniform(0.0, rng.choice([30.0, 800.0, 4000.0]), (gy, gx))
        stats['per_pixel_trials'] = stats.get('per_pixel_trials', 0) + 1
        try:
            pw_, ph_, pnp, _ = cube.raytrace(R.Rays.grid(xpts, ypts, los=np.ascontiguousarray(los), hts=hts), None, zref, max_seg)
            g2 = None
        except R.NoLevels:
            g2 = 'NoLevels'
        except Exception as e:
            g2 = type(e).__name__
        if equal:
            if g2 or not (np.array_equal(pw_, wet, equal_nan=True) and np.array_equal(ph_, hyd, equal_nan=True) and np.array_equal(pnp, nparts)):
                bad.append(dict(tag, kind='per-pixel heights, all equal: not the slice result', gpu=g2))
        else:
            try:
                cc = c
                if c['ys'][0] > c['ys'][-1]:        # (the C oracle takes ascending axes: flip as scipy / the library do)
                    cc = dict(c, ys=c['ys'][::-1].copy(), wet=c['wet'][:, ::-1, :].copy(), hydro=c['hydro'][:, ::-1, :].copy())
                qw, qh, qnp = OC.build_cube_ray_per_pixel(cc, yy, xx, hts, los, zref, max_seg=max_seg)
                o2 = None
                import os
                if os.environ.get('FUZZ_DEBUG_TRIAL') == str(trial):
                    nw, nh, nnp = O.build_cube_ray_per_pixel(yy.ravel(), xx.ravel(), hts.ravel(), los.reshape(-1, 3), ip, MAX_SEGMENT_LENGTH=max_seg, MAX_TROPO_HEIGHT=zref)
                    print('DEBUG hts', hts.ravel()[:8], 'zs', c['zs'][:6], 'GPU', ph_.ravel()[:6], 'C', qh.ravel()[:6], 'NumPy', nh[:6], 'nparts gpu', pnp, 'C', qnp, 'np', nnp,
                          'max|GPU-NumPy|', np.nanmax(np.abs(ph_.ravel() - nh)), 'max|C-NumPy|', np.nanmax(np.abs(qh.ravel() - nh)))
            except Exception as e:
                o2 = type(e).__name__
            if g2 or o2:
                if not (g2 == 'NoLevels' and o2 is None and not qnp.any()) and not (g2 == o2 == 'ValueError'):
                    bad.append(dict(tag, kind='per-pixel heights: error mismatch', oracle=o2, gpu=g2))
            else:
                kzt = cube.ray_levels(float(hts.min()), zref)[2]
                dpp = max(float(np.nanmax(np.abs(pw_ - qw))) if np.isfinite(qw).any() else 0.0, float(np.nanmax(np.abs(ph_ - qh))) if np.isfinite(qh).any() else 0.0)
                worst['per_pixel_heights'] = max(worst.get('per_pixel_heights', 0.0), dpp)
                if not np.array_equal(pnp, qnp[kzt]) or not np.array_equal(np.isnan(ph_), np.isnan(qh)) or dpp > 2e-8:
                    bad.append(dict(tag, kind='per-pixel heights vs oracle', d=dpp, nparts_equal=bool(np.array_equal(pnp, qnp[kzt])),
                                    gpu_nan=int(np.isnan(ph_).sum()), oracle_nan=int(np.isnan(qh).sum())))
print(json.dumps(dict(stats=stats, worst_abs_m=worst, n_bad=len(bad))))
for b in bad[:40]:
    print(json.dumps(b))
