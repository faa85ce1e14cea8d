#!/usr/bin/env python
"""Per-file time of settings.i_reinterp = 1 (re-interpolation in every pass) on the bench.py file, HBM-resident."""
import os, sys, time, json, datetime as dt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
from pgw4era5_amd.device import default_context
ctx = default_context()
case = synthetic.make_case(nlat=721, nlon=1440, nlev=137, seed=1, dtype=np.float64)
deltas = s3.DeltaSet(ctx, case['deltas'], case['delta_times'], case['plev'], np.float64)
era = s3._upload_era(ctx, case['era'], np.float64)
coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
out = {}
for i in range(2):
    _, info = s3.process_file_device_reinterp(ctx, era, coeffs, deltas, case['target_dt'] + dt.timedelta(hours=i), True, out=out)
ctx.sync()
ctx.profile(True); ctx.profile_reset()
t0 = time.perf_counter()
n = 4
for i in range(n):
    _, info = s3.process_file_device_reinterp(ctx, era, coeffs, deltas, case['target_dt'] + dt.timedelta(hours=2 + i), True, out=out)
ctx.sync()
el = (time.perf_counter() - t0) / n
from pgw4era5_amd import _lib
prof = {k: ctx.profile_get(k) for k in _lib.KERNEL_IDS}
print(json.dumps(dict(ms_per_file=round(el * 1e3, 2), n_iter=info['n_iter'],
                      kernels={k: dict(launches=c // n, ms_per_file=round(ms / n, 3)) for k, (c, ms) in prof.items() if c})))
