#!/usr/bin/env python
"""Randomised parity sweep: many small random files (grid shapes down to ONE column, level counts, storage dtype, plev
subsets, time stamps incl. exact records and the December -> January wrap, ps_hist above / inside the delta levels) through
the HIP file path and through the CPU oracle; reports every disagreement beyond the tolerances of tests/test_hip_parity.py.
Test infrastructure (imports oracle/).  usage: python tools/fuzz_parity.py [--cases 300] [--seed 0]"""
import argparse
import datetime as dt
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
from oracle import pgw_oracle as O, pgw_oracle_refdtype as R


def scaled(a, b):
    scale = np.nanmax(np.abs(b), axis=(2, 3), keepdims=True)
    scale = np.where(scale > 0, scale, 1.0)
    return float(np.nanmax(np.abs(a - b) / scale))


def one(rng, i, run=True, debug=False):
    nlat, nlon = int(rng.integers(1, 10)), int(rng.integers(1, 14))
    nlev = int(rng.integers(8, 45))
    dtype = np.float32 if rng.random() < 0.5 else np.float64
    mode = rng.choice(['file', 'file', 'file', 'local', 'reinterp', 'reinterp_local'])
    # plev subset that keeps p_ref = 30000 Pa and the top / bottom levels
    keep = np.ones(len(synthetic.PLEV19), dtype=bool)
    if rng.random() < 0.4:
        drop = rng.choice(np.arange(1, len(keep) - 1), size=int(rng.integers(1, 8)), replace=False)
        keep[drop] = False
        keep[np.nonzero(synthetic.PLEV19 == 30000.0)[0]] = True
    plev = synthetic.PLEV19[keep]
    month = int(rng.integers(1, 13))
    kind = rng.random()
    if kind < 0.2:
        target = None                                   # the case's own default
    elif kind < 0.4:
        target = dt.datetime(2006, month, 15, 12) if month != 2 else dt.datetime(2006, 2, 14)   # near / at a record
    elif kind < 0.6:
        target = dt.datetime(2006, 12, 31, 23) if rng.random() < 0.5 else dt.datetime(2006, 1, 1, 0)   # year wrap
    else:
        target = dt.datetime(2006, month, int(rng.integers(1, 29)), int(rng.integers(0, 24)))
    kw = dict(nlat=nlat, nlon=nlon, nlev=nlev, seed=1000 + i, dtype=dtype, plev=plev)
    if target is not None:
        kw['target_dt'] = target
    c = synthetic.make_case(**kw)
    d = c['deltas']
    times = c['delta_times']
    axes = None
    if rng.random() < 0.3 and mode in ('file', 'reinterp'):
        # every delta file on its own time axis (load_delta reads each file on its own, functions.py:195-303): a random subset of
        # the variables gets a random number of records on random days - incl. quad-group members (ta, ua, ps_hist ...) that
        # then differ from ta's axis, a leap day now and then (dropped), and single-record... no: at least two records
        day = np.timedelta64(1, 'D')
        stamps = {}
        for var in rng.choice(sorted(d), size=int(rng.integers(1, 6)), replace=False):
            n = int(rng.integers(2, 30))
            year = 1996 if rng.random() < 0.3 else 1995
            days = np.sort(rng.choice(np.arange(366 if year == 1996 else 365), size=n, replace=False))
            stamps[str(var)] = np.datetime64('%d-01-01T00:00:00' % year) + days * day + np.timedelta64(int(rng.integers(0, 24)), 'h')
        d, times = synthetic.resample_deltas(c, stamps, seed=i)
        axes = sorted(stamps)
    if rng.random() < 0.3:                              # ps_hist above every delta level somewhere
        d['ps_hist'] = d['ps_hist'].copy()
        d['ps_hist'][:, rng.integers(0, nlat), rng.integers(0, nlon)] = 104000.0
    ignore_top = True
    inject = None
    if rng.random() < 0.12 and mode == 'file':          # data errors: both sides must raise the reference's exception
        inject = str(rng.choice(['ps_hist_low', 'ps_nan', 'ps_below_p_ref', 'top_check']))
        jj, ii = int(rng.integers(0, nlat)), int(rng.integers(0, nlon))
        if inject == 'ps_hist_low':                     # ValueError() of replace_delta_sfc, functions.py:360-361
            d['ps_hist'] = d['ps_hist'].copy(); d['ps_hist'][:, jj, ii] = 50.0
        elif inject == 'ps_nan':                        # all-NaN p_diff column, functions.py:162-165
            c['era']['PS'] = c['era']['PS'].copy(); c['era']['PS'][0, jj, ii] = np.nan
        elif inject == 'ps_below_p_ref':                # surface above the reference level
            c['era']['PS'] = c['era']['PS'].copy(); c['era']['PS'][0, jj, ii] = 25000.0
        else:
            ignore_top = False                          # the model top of the synthetic levels lies above the delta top
    args = (c['era'], d, times, c['plev'], c['target_dt'], ignore_top)
    if not run:
        return None, 'skip'
    desc = dict(i=i, shape=[nlat, nlon, nlev], dtype=np.dtype(dtype).name, mode=str(mode), S=int(len(plev)), target=str(c['target_dt']),
                inject=inject, own_time_axes=axes)
    try:
        if mode in ('reinterp', 'reinterp_local'):
            pr = None if mode == 'reinterp_local' else 30000.0
            got = s3.pgw_for_era5_arrays(*args, i_reinterp=True, p_ref='local' if pr is None else None)
            if dtype == np.float32:                       # reference-dtype mode (the default on float32 files)
                want = R.pgw_for_era5_arrays_reinterp(*args, p_ref=pr)
                tol = dict(PS=2.5e-7, T=6e-8, QV=6e-7)     # T: one float32 ulp of ps_pgw times the lapse rate
            else:
                want = O.pgw_for_era5_arrays_reinterp(c['era'], {k: np.asarray(v, dtype=np.float64) for k, v in d.items()}, *args[2:], p_ref=pr)
                tol = dict(PS=1e-9, T=1e-9, QV=1e-9)
        elif mode == 'local':
            got = s3.pgw_for_era5_arrays(*args, p_ref='local', ref_dtype=False)
            f64 = lambda x: np.asarray(x, dtype=np.float64)
            era = {k: (f64(v) if isinstance(v, np.ndarray) and v.dtype == np.float32 else v) for k, v in c['era'].items()}
            dd = {k: f64(v) for k, v in d.items()}
            akm, bkm = O.full_level_coeffs(era['ak'], era['bk'])
            _, pa = O.hybrid_pressure(era['ak'], era['bk'], era['PS'], akm, bkm)
            ld = lambda k: O.load_delta_values(dd[k], c['delta_times'], c['target_dt'])
            ta = era['T'] + O.vert_interp_delta(ld('ta'), c['plev'], pa, ld('tas'), ld('ps_hist'), True)
            hur = O.specific_to_relative_humidity(era['QV'], pa, era['T']) + \
                O.vert_interp_delta(ld('hur'), c['plev'], pa, ld('hurs'), ld('ps_hist'), True)
            w = O.adjust_ps_loop_local_pref(era['ak'], era['bk'], akm, bkm, era['PS'], era['FIS'], era['T'], era['QV'],
                                            ta, hur, ld('zg'), c['plev'])
            ua = era['U'] + O.vert_interp_delta(ld('ua'), c['plev'], pa, None, None, True)
            va = era['V'] + O.vert_interp_delta(ld('va'), c['plev'], pa, None, None, True)
            want = dict(n_iter=w['n_iter'], max_err=w['max_err'], PS=w['ps_pgw'], QV=w['hus_pgw'], T=ta, U=ua, V=va)
            tol = dict(PS=1e-9, T=1e-9, QV=1e-9) if dtype == np.float64 else dict(PS=1e-6, T=2e-6, QV=3e-6)
        elif dtype == np.float32:
            got = s3.pgw_for_era5_arrays(*args)
            want = R.pgw_for_era5_arrays(*args)
            tol = dict(PS=3.2e-7, T=1e-9, QV=6e-7)      # PS: up to 4 float32 ulp (device log vs numpy)
        else:
            got = s3.pgw_for_era5_arrays(*args)
            want = O.pgw_for_era5_arrays(*args)
            tol = dict(PS=1e-9, T=1e-9, QV=1e-9)
    except Exception as e:                              # noqa: BLE001 - both sides must agree on failures too
        if mode == 'local':
            return desc, 'both raise: %s (local mode: the oracle side raised or the HIP side did)' % type(e).__name__
        try:
            if mode in ('reinterp', 'reinterp_local'):
                (R if dtype == np.float32 else O).pgw_for_era5_arrays_reinterp(*args, p_ref=None if mode == 'reinterp_local' else 30000.0)
            else:
                (R if dtype == np.float32 else O).pgw_for_era5_arrays(*args)
        except Exception as e2:                         # noqa: BLE001
            same = type(e) is type(e2) and (str(e) == str(e2) or not str(e2) or str(e).startswith(str(e2).rstrip('.!')[:40]))
            return desc, ('both raise: %s' % type(e).__name__) if same else \
                'different errors: %s: %s / %s: %s' % (type(e).__name__, str(e)[:120], type(e2).__name__, str(e2)[:120])
        return desc, 'HIP raises alone: %s: %s' % (type(e).__name__, e)
    if axes and dtype == np.float32:
        # a quad-group variable on another time axis than ta is interpolated in time before the kernel and held as ONE float32
        # field there (DeltaSet.pair_on_axis_of): one float32 rounding of a delta (6e-8 of a few K / m/s / %)
        tol = dict(tol, T=max(tol['T'], 5e-9), QV=max(tol['QV'], 1e-6))
    bad = []
    if got['n_iter'] != want['n_iter']:
        bad.append('n_iter %d vs %d (max_err %s vs %s)' % (got['n_iter'], want['n_iter'], got['max_err'][-2:], want['max_err'][-2:]))
    dps = float(np.max(np.abs(got['PS'].astype(np.float64) - want['PS']) / np.abs(want['PS'])))
    if not dps <= tol['PS']:
        bad.append('PS %.3e' % dps)
    for k in ('T', 'U', 'V'):
        dd = float(np.nanmax(np.abs(got[k] - want[k]) / np.maximum(np.abs(want[k]), 1.0)))
        # i_reinterp on a float32 file: ps_pgw is held in float32 (as in the reference), the oracle's is float64; the
        # 4e-8 relative pressure shift times the white-noise vertical gradient of the synthetic winds is ~1e-5 m/s
        lim = 1e-4 if (mode in ('reinterp', 'reinterp_local') and dtype == np.float32 and k != 'T') else max(tol['T'], 1e-9)
        if axes and dtype == np.float32 and k != 'T':
            lim = max(lim, 3e-7)                       # ua / va on another axis than ta: the delta (a few m/s) rounded to float32 once
        if not dd <= lim:
            bad.append('%s %.3e' % (k, dd))
    dq = scaled(got['QV'], want['QV'])
    if not dq <= tol['QV']:
        bad.append('QV %.3e' % dq)
    for k in ('T_SKIN', 'T_SO', 'FR_SEA_ICE'):
        if k in got and k in want and mode == 'file':
            if not np.allclose(got[k], want[k], rtol=2.5e-7 if dtype == np.float32 else 1e-9, atol=1e-12, equal_nan=True):
                bad.append(k)
    if debug:
        for k in ('U', 'V', 'FR_SEA_ICE', 'PS', 'T', 'QV'):
            if k in got and k in want:
                g, w = np.asarray(got[k], dtype=np.float64), np.asarray(want[k], dtype=np.float64)
                dd = np.abs(g - w)
                j = np.unravel_index(np.nanargmax(dd), dd.shape)
                print(k, 'max |diff| %.4e at %s: got %.9g want %.9g; dtype %s / %s' % (dd[j], j, g[j], w[j], got[k].dtype, want[k].dtype))
                with np.errstate(invalid='ignore', divide='ignore'):
                    rr = np.where(dd > 0, dd / np.abs(w), 0.0)
                j = np.unravel_index(np.nanargmax(rr), rr.shape)
                print(k, 'max rel diff %.4e at %s: got %.9g want %.9g' % (rr[j], j, g[j], w[j]))
    return desc, ('; '.join(bad) if bad else 'ok')


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--cases', type=int, default=300)
    p.add_argument('--seed', type=int, default=0)
    p.add_argument('--only', type=int, default=-1, help='replay the sweep but run (and describe) this case alone')
    a = p.parse_args()
    rng = np.random.default_rng(a.seed)
    t0 = time.time()
    counts = {}
    fails = []
    for i in range(a.cases):
        desc, res = one(rng, i, run=(a.only < 0 or i == a.only), debug=(i == a.only))
        if a.only >= 0 and i == a.only:
            print(json.dumps(dict(desc, result=res)))
        key = 'ok' if res == 'ok' else ('skip' if res == 'skip' else ('both raise' if res.startswith('both raise') else 'FAIL'))
        counts[key] = counts.get(key, 0) + 1
        if key == 'FAIL':
            fails.append(dict(desc, result=res))
            print(json.dumps(fails[-1]), flush=True)
        if (i + 1) % 50 == 0:
            print('%d cases, %.0f s: %s' % (i + 1, time.time() - t0, counts), flush=True)
    print(json.dumps(dict(cases=a.cases, seed=a.seed, counts=counts, failures=fails[:20])))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
