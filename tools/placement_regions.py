#!/usr/bin/env python
"""Which placement of the file path's nine level arrays (T, QV, U, V in; T, QV, U, V out; the vapour-pressure workspace) over
the card's memory 'regions' is fast?  A pool of POOL field-sized arrays (default 110 = 125 GB, more than one region), classes
by copy probes against one representative per class (`Context.placement_probe`; a copy inside a region is slow), then the real
file path on arrays chosen per case.  See DESIGN.md section 4.  Usage (GPU box): POOL=110 python tools/placement_regions.py"""
import os, sys, json
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3, _lib
from pgw4era5_amd.device import default_context, DeviceArray

ctx = default_context()
dtype = np.float64
P = int(os.environ.get('POOL', 110))
case = synthetic.make_case(nlat=721, nlon=1440, nlev=137, seed=1, dtype=dtype)
coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
shape = case['era']['T'].shape
deltas = s3.DeltaSet(ctx, case['deltas'], case['delta_times'], case['plev'], dtype)
base = s3._upload_era(ctx, case['era'], dtype)
pool = [ctx.empty(shape, np.float64) for _ in range(P)]

# ---- classes: a representative per class; an array belongs to the first representative it copies SLOWLY from
unclassified = list(range(P))
classes = []
while unclassified and len(classes) < 6:
    r = unclassified[0]
    rates = {k: ctx.placement_probe([], [pool[r], pool[k]], rows=64) for k in unclassified[1:]}     # two write streams: 5.0 inside, 6.6 TB/s across
    if not rates:
        classes.append([r]); break
    lo, hi = min(rates.values()), max(rates.values())
    if hi - lo < 0.10 * hi:                      # no split left: everything that remains is one class
        classes.append(unclassified); unclassified = []
        break
    mid = 0.5 * (lo + hi)
    members = [r] + [k for k, v in rates.items() if v < mid]
    classes.append(members)
    unclassified = [k for k in unclassified if k not in members]
    print('class %d: representative %d, %d members, copy rates %.0f (inside) .. %.0f (outside) GB/s' % (len(classes) - 1, r, len(members), lo, hi), flush=True)
if unclassified:
    classes.append(unclassified)
print('classes (sizes):', [len(c) for c in classes])
print('class of each pool array:', ''.join(str(next(i for i, c in enumerate(classes) if k in c)) for k in range(P)))
free = [list(c) for c in classes]
taken_class = []


def take(ci):
    ci = ci % len(free)
    for d in range(len(free)):                      # fall back to the next class that still has arrays
        c = free[(ci + d) % len(free)]
        if c:
            taken_class.append((ci + d) % len(free))
            return c.pop(0)
    raise RuntimeError('pool exhausted')


def run(tag, cls_in, cls_out, cls_ws):
    del taken_class[:]
    idx_in, idx_out, idx_ws = [take(c) for c in cls_in], [take(c) for c in cls_out], take(cls_ws)
    actual = list(taken_class)
    era = dict(base)
    fin = int(np.prod(shape)) * 8
    for k, f in zip(idx_in, ('T', 'QV', 'U', 'V')):
        v = DeviceArray(ctx, shape, dtype, ptr=pool[k].ptr, owner=pool[k])
        ctx._check(ctx.lib.pgw_memcpy_d2d(ctx.handle, v.ptr, base[f].ptr, fin))
        era[f] = v
    out = {f: DeviceArray(ctx, shape, np.float64, ptr=pool[k].ptr, owner=pool[k]) for k, f in zip(idx_out, ('T', 'QV', 'U', 'V'))}
    ctx.ws_adopt(0, pool[idx_ws])                  # the library's from here on; index never used again (take() popped it)
    bare = ctx.placement_probe([era[f] for f in ('T', 'QV', 'U', 'V')], [out[f] for f in ('T', 'QV', 'U', 'V')], reps=3)
    ctx.sync(); ctx.profile(True)
    for i in range(2):
        s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out)
    ctx.sync(); ctx.profile_reset()
    n = 6
    ctx._check(ctx.lib.pgw_timer_start(ctx.handle))
    for i in range(n):
        s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out)
    ms = C.c_double(); ctx._check(ctx.lib.pgw_timer_stop(ctx.handle, C.byref(ms)))
    row = {'case': tag, 'classes_in_out_ws': actual, 'ms_per_file': round(ms.value / n, 3), 'bare_GBps': round(bare)}
    tot = 0.0
    for k in ('quad_delta', 'ps_loop_multi', 'finalize'):
        c, t = ctx.profile_get(k)
        row[k] = round(t / max(c, 1), 4); tot += t / n
    row['sum_kernels'] = round(tot, 3)
    print(json.dumps(row), flush=True)
    ctx.profile(False)


nc = len(classes)
if nc >= 2:
    for rep in range(2):
        run('all in class 0', [0] * 4, [0] * 4, 0)
        run('two classes alternating (what SpreadPool does)', [0, 1, 0, 1], [1, 0, 1, 0], 0)
        run('in 0 1 0 1, out 0 1 0 1, ws 0 (T_pgw and e together)', [0, 1, 0, 1], [0, 1, 0, 1], 0)
        run('in 0 0 1 1 (T, QV together), out 0 1 0 1, ws 0', [0, 0, 1, 1], [0, 1, 0, 1], 0)
        run('in 0 0 1 1, out 0 0 1 1, ws 0 (T_pgw, QV_out and e together)', [0, 0, 1, 1], [0, 0, 1, 1], 0)
        if nc >= 3 and os.environ.get('THREE'):
            run('three classes round robin', [0, 1, 2, 0], [1, 2, 0, 1], 2)
        if nc >= 4:
            run('four classes round robin', [0, 1, 2, 3], [1, 2, 3, 0], 2)
            run('four classes, in 0 1 2 3, out 2 3 0 1', [0, 1, 2, 3], [2, 3, 0, 1], 1)
else:
    print('one class only: nothing to compare')
