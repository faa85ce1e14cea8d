#!/usr/bin/env python
"""The quad kernel's time against the SPACING of its nine level arrays in HBM: T, QV, U, V in and T, QV, U, V out as views of
one arena at base + k * spacing (the vapour-pressure workspace stays the library's).  hipMalloc places 1.138 GB arrays
0x44000000 apart (17 x 64 MiB): identical low-order address bits for all of them."""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
from pgw4era5_amd.device import default_context, DeviceArray
ctx = default_context()
dtype = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else np.float64
case = synthetic.make_case(nlat=721, nlon=1440, nlev=137, seed=1, dtype=dtype)
coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
deltas = s3.DeltaSet(ctx, case['deltas'], case['delta_times'], case['plev'], dtype)
shape = case['era']['T'].shape
fb = int(np.prod(shape)) * 8                      # output fields are float64 in both modes measured here (f32: reference mode)
fin = int(np.prod(shape)) * np.dtype(dtype).itemsize
base_era = s3._upload_era(ctx, case['era'], dtype)
MB = 1 << 20
spacings = [('hipMalloc-like 0x44000000', 0x44000000), ('exact field bytes', fb), ('+4 KiB', 0x44000000 + 4096), ('+64 KiB', 0x44000000 + 65536),
            ('+256 KiB', 0x44000000 + 262144), ('+1 MiB', 0x44000000 + MB), ('+3 MiB', 0x44000000 + 3 * MB), ('+33 MiB', 0x44000000 + 33 * MB),
            ('+100 MiB + 12 KiB', 0x44000000 + 100 * MB + 12288), ('hipMalloc-like again', 0x44000000)]
arena = ctx.empty(((9 * (0x44000000 + 101 * MB)) // 8,), np.float64)
for name, sp in spacings:
    era = dict(base_era)
    for k, f in enumerate(('T', 'QV', 'U', 'V')):
        v = DeviceArray(ctx, shape, dtype, ptr=arena.ptr + k * sp, owner=arena)
        ctx._check(ctx.lib.pgw_memcpy_d2d(ctx.handle, v.ptr, base_era[f].ptr, fin))
        era[f] = v
    odt = np.float64
    out = {f: DeviceArray(ctx, shape, odt, ptr=arena.ptr + (4 + k) * sp, owner=arena) for k, f in enumerate(('T', 'QV', 'U', 'V'))}
    ctx.sync()
    ctx.profile(True)
    for i in range(2):
        s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out)
    ctx.sync(); ctx.profile_reset()
    for i in range(5):
        _, info = s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out)
    ctx.sync()
    q = ctx.profile_get('quad_delta'); lp = ctx.profile_get('ps_loop_multi'); fz = ctx.profile_get('finalize')
    print(json.dumps(dict(spacing=name, quad_ms=round(q[1] / q[0], 4), loop_ms=round(lp[1] / lp[0], 4), finalize_ms=round(fz[1] / fz[0], 4), n_iter=info['n_iter'])), flush=True)
    for k in list(out):
        if k not in ('T', 'QV', 'U', 'V'):
            out[k].free()
