#!/usr/bin/env python
"""Where hipMalloc puts the level arrays decides the rate the file path's kernels get (tools/micro/pair_matrix.hip: arrays fall
into classes; a copy between arrays of one class is slow, between classes fast).  This script does it with the REAL file path:
a pool of P field-sized arrays, their classes from copy probes against array 0 (`Context.placement_probe`), then the file path
with (a) inputs and outputs + vapour-pressure workspace all in ONE class, (b) inputs in one class, outputs + workspace in
another, (c) plain fresh allocations.  Usage (GPU box): python tools/placement_classes.py [f32]"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
from pgw4era5_amd.device import default_context, DeviceArray

ctx = default_context()
f32 = len(sys.argv) > 1 and sys.argv[1] == 'f32'
dtype = np.float32 if f32 else np.float64
P = int(os.environ.get('POOL', 20))
case = synthetic.make_case(nlat=721, nlon=1440, nlev=137, seed=1, dtype=dtype)
coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
shape = case['era']['T'].shape
deltas = s3.DeltaSet(ctx, case['deltas'], case['delta_times'], case['plev'], dtype)
base = s3._upload_era(ctx, case['era'], dtype)
pool = [ctx.empty(shape, np.float64) for _ in range(P)]
fwd = [0.0] + [ctx.placement_probe([pool[0]], [pool[j]]) for j in range(1, P)]
bwd = [0.0] + [ctx.placement_probe([pool[j]], [pool[0]]) for j in range(1, P)]
print('copy 0 -> j :', ' '.join('%5.0f' % v for v in fwd))
print('copy j -> 0 :', ' '.join('%5.0f' % v for v in bwd))
hi, lo = max(fwd[1:]), min(fwd[1:])
same = [0] + [j for j in range(1, P) if fwd[j] < 0.5 * (hi + lo)]
other = [j for j in range(1, P) if fwd[j] >= 0.5 * (hi + lo)]
print('class of array 0:', same, ' other:', other, ' spread %.0f .. %.0f GB/s' % (lo, hi))


def run(tag, era_idx, out_idx, ws_idx):
    era = dict(base)
    fin = int(np.prod(shape)) * np.dtype(dtype).itemsize
    if era_idx is not None:
        for k, f in zip(era_idx, ('T', 'QV', 'U', 'V')):
            v = DeviceArray(ctx, shape, dtype, ptr=pool[k].ptr, owner=pool[k])
            ctx._check(ctx.lib.pgw_memcpy_d2d(ctx.handle, v.ptr, base[f].ptr, fin))
            era[f] = v
    out = {}
    if out_idx is not None:
        for k, f in zip(out_idx, ('T', 'QV', 'U', 'V')):
            out[f] = DeviceArray(ctx, shape, np.float64 if (not f32 or True) else dtype, ptr=pool[k].ptr, owner=pool[k])
    if ws_idx is not None:
        ctx.ws_adopt(0, pool[ws_idx])
    ctx.sync(); ctx.profile(True)
    for i in range(2):
        s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out)
    ctx.sync(); ctx.profile_reset()
    ctx._check(ctx.lib.pgw_timer_start(ctx.handle))
    n = 6
    for i in range(n):
        _, info = s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out)
    import ctypes as C
    ms = C.c_double(); ctx._check(ctx.lib.pgw_timer_stop(ctx.handle, C.byref(ms)))
    row = {'case': tag, 'ms_per_file': round(ms.value / n, 3)}
    from pgw4era5_amd import _lib
    tot = 0.0
    for k in _lib.KERNEL_IDS:
        c, t = ctx.profile_get(k)
        if c:
            row[k] = [c // n, round(t / c, 4)]
            tot += t / n
    row['sum_kernels_ms_per_file'] = round(tot, 3)
    print(json.dumps(row), flush=True)
    return out


a, b = (same, other) if len(same) >= len(other) else (other, same)
if len(b) >= 5 and len(a) >= 9:
    # an adopted workspace belongs to the library and is freed by the next adoption: a[8] and b[4] appear nowhere else
    run('fresh allocations (inputs: as uploaded; outputs, workspace: the library\'s)', None, None, None)
    run('all in ONE class', a[0:4], a[4:8], a[8])
    run('inputs class A, outputs + workspace class B', a[0:4], b[0:4], b[4])
    run('inputs class A, outputs + workspace class B (again)', a[0:4], b[0:4], None)
    run('all in ONE class (again; the workspace stays in B)', a[0:4], a[4:8], None)
else:
    print('pool did not split into two classes of >= 9 and >= 5 arrays; raise POOL')
