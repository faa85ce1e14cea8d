#!/usr/bin/env python
"""Randomised sweep of the step_03 command line over ERA5 FILE LAYOUTS: the same physics as tools/fuzz_parity.py, but
every case goes through NetCDF files whose layout varies - float32 / float64 variables, `time` unlimited or fixed,
variables in random order, extra variables the driver must pass through untouched (integer, float, scalar char), `akm` /
`bkm` stored in the file (step_03_apply_to_era.py:68-70: then they replace the half-level means), a 4-D field stored with
another dimension order (the reference's `.transpose(...)`), raw (file byte order to the GPU) or converted host I/O.
Checks the written file against the oracle and every untouched variable / attribute against the input.
Test infrastructure (imports oracle/).  usage: python tools/fuzz_files.py [--cases 60] [--seed 0]"""
import argparse
import datetime as dt
import json
import os
import shutil
import sys
import tempfile
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, ncio, step_03_apply_to_era as s3
from oracle import pgw_oracle as O, pgw_oracle_refdtype as R


def one(rng, i, root):
    nlat, nlon, nlev = int(rng.integers(2, 8)), int(rng.integers(2, 11)), int(rng.integers(8, 30))
    dtype = np.float32 if rng.random() < 0.6 else np.float64
    target = dt.datetime(2006, int(rng.integers(1, 13)), int(rng.integers(1, 29)), int(rng.integers(0, 24)))
    c = synthetic.make_case(nlat=nlat, nlon=nlon, nlev=nlev, seed=5000 + i, dtype=dtype, target_dt=target)
    d_era, d_del, d_out = (os.path.join(root, '%s%d' % (k, i)) for k in ('era', 'deltas', 'out'))
    path = synthetic.write_case_files(c, d_era, d_del)
    ds = ncio.open_dataset(path, decode_times=False)
    desc = dict(i=i, shape=[nlat, nlon, nlev], dtype=np.dtype(dtype).name, layout=[])
    F = ncio.Field
    era = dict(c['era'])
    if rng.random() < 0.4:                                       # full-level coefficients in the file (not the half-level means)
        akm = 0.5 * (era['ak'][1:] + era['ak'][:-1]) * (1 + 1e-9)
        bkm = 0.5 * (era['bk'][1:] + era['bk'][:-1])
        ds['akm'] = F(akm, ('level',)); ds['bkm'] = F(bkm, ('level',))
        era['akm'], era['bkm'] = akm, bkm
        desc['layout'].append('akm')
    if rng.random() < 0.5:
        ds['counts'] = F(rng.integers(-5, 5, size=(nlat, nlon)).astype(np.int32), ('lat', 'lon'), attrs=dict(long_name='an integer field'))
        ds['weights'] = F(rng.normal(size=nlev + 1), ('level1',), attrs=dict(units='1'))
        desc['layout'].append('extras')
    if rng.random() < 0.3:                                       # T stored (time, lat, lon, level): read through .transpose
        t = ds['T']
        ds['T'] = F(np.ascontiguousarray(np.transpose(t.values, (0, 2, 3, 1))), ('time', 'lat', 'lon', 'level'), attrs=dict(t.attrs))
        desc['layout'].append('T transposed')
    names = list(ds.variables)
    if rng.random() < 0.6:
        rng.shuffle(names)
        desc['layout'].append('shuffled')
    out = ncio.Dataset(attrs=dict(ds.attrs, history='fuzz %d' % i))
    for k in names:
        out[k] = ds[k]
    out.record_dim = 'time' if rng.random() < 0.6 else None
    desc['layout'].append('record' if out.record_dim else 'fixed')
    ncio.to_netcdf(out, path)
    raw = rng.random() < 0.6
    os.environ['PGW_IO_RAW'] = '1' if raw else '0'
    desc['layout'].append('raw' if raw else 'converted')
    stamp = '{:%Y%m%d%H}'.format(target)
    n_iters = s3._cli(['-i', d_era, '-o', d_out, '-d', d_del, '-f', stamp, '-l', stamp, '-H', '1', '-p', '1', '-t'])
    args = (era, c['deltas'], c['delta_times'], c['plev'], c['target_dt'], True)
    want = (R if dtype == np.float32 else O).pgw_for_era5_arrays(*args)
    got = ncio.open_dataset(os.path.join(d_out, os.path.basename(path)), decode_times=False)
    inp = ncio.open_dataset(path, decode_times=False)
    bad = []
    if n_iters != [want['n_iter']]:
        bad.append('n_iter %s vs %d' % (n_iters, want['n_iter']))
    f32 = dtype == np.float32
    # reference mode: PS within a few float32 ulp of the reference-dtype oracle
    for k, tol in (('PS', 2.5e-7 if f32 else 1e-9), ('T_SKIN', 1.3e-7 if f32 else 1e-9), ('T_SO', 1.3e-7 if f32 else 1e-9),
                   ('FR_SEA_ICE', 1.3e-7 if f32 else 1e-9), ('T', 1e-9), ('U', 1e-9), ('V', 1e-9)):
        g = got[k].transpose(*inp[k].dims).values if k != 'T' else got['T'].transpose('time', 'level', 'lat', 'lon').values
        if not np.allclose(g, want[k], rtol=tol, atol=1e-9 if k in 'TUV' else 0, equal_nan=True):
            bad.append(k)
    scale = np.nanmax(np.abs(want['QV']), axis=(2, 3), keepdims=True)
    if not np.nanmax(np.abs(got['QV'].values - want['QV']) / scale) < (6e-7 if f32 else 1e-9):
        bad.append('QV')
    if 'RELHUM' in got:
        bad.append('RELHUM written')
    touched = {'PS', 'T_SKIN', 'T_SO', 'FR_SEA_ICE', 'T', 'U', 'V', 'QV'}
    if list(got.variables) != [k for k in inp.variables]:
        bad.append('variable order / set: %s vs %s' % (list(got.variables), list(inp.variables)))
    for k in inp.variables:
        if k in got:
            if dict(got[k].attrs).keys() != dict(inp[k].attrs).keys():
                bad.append('attrs of ' + k)
            if k not in touched and not (got[k].dims == inp[k].dims and got[k].values.dtype == inp[k].values.dtype and
                                         np.array_equal(got[k].values, inp[k].values)):
                bad.append('pass-through ' + k)
    if got.record_dim != inp.record_dim:
        bad.append('record dimension %r vs %r' % (got.record_dim, inp.record_dim))
    if got.attrs.get('history') != 'fuzz %d' % i:
        bad.append('global attrs')
    for p in (d_era, d_del, d_out):
        shutil.rmtree(p, ignore_errors=True)
    return desc, ('; '.join(bad) if bad else 'ok')


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--cases', type=int, default=60)
    p.add_argument('--seed', type=int, default=0)
    a = p.parse_args()
    rng = np.random.default_rng(a.seed)
    root = tempfile.mkdtemp(prefix='pgw_fuzz_files')
    keep = os.environ.get('PGW_IO_RAW')
    import pgw4era5_amd.settings as S
    debug, S.i_debug = S.i_debug, -1                                 # no progress lines
    t0 = time.time()
    fails, ok = [], 0
    try:
        for i in range(a.cases):
            try:
                desc, res = one(rng, i, root)
            except Exception as e:                              # noqa: BLE001
                desc, res = dict(i=i), 'raised %s: %s' % (type(e).__name__, str(e)[:300])
            if res == 'ok':
                ok += 1
            else:
                fails.append(dict(desc, result=res))
                print(json.dumps(fails[-1]), flush=True)
    finally:
        S.i_debug = debug
        if keep is None:
            os.environ.pop('PGW_IO_RAW', None)
        else:
            os.environ['PGW_IO_RAW'] = keep
        shutil.rmtree(root, ignore_errors=True)
    print(json.dumps(dict(cases=a.cases, seed=a.seed, ok=ok, seconds=round(time.time() - t0, 1), failures=fails[:20])))
    return 1 if fails else 0


if __name__ == '__main__':
    sys.exit(main())
