#!/usr/bin/env python
"""Per-file time of the settings.p_ref_inp = None mode (local reference level) on the bench.py file, HBM-resident:
the multi-pass kernel's LOCAL variant against one launch per pass (k_local_p_ref + two scans)."""
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
res = {}
for name, mp in (('one_launch_per_pass', 0), ('multi_pass', 1), ('one_launch_per_pass_again', 0), ('multi_pass_again', 1)):
    ctx.set_option('multipass', mp)
    for i in range(2):
        _, info = s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'] + dt.timedelta(hours=i), True, p_ref='local', out=out)
    ctx.sync()
    t0 = time.perf_counter()
    n = 6
    for i in range(n):
        _, info = s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'] + dt.timedelta(hours=2 + i), True, p_ref='local', out=out)
    ctx.sync()
    res[name] = dict(ms_per_file=round((time.perf_counter() - t0) / n * 1e3, 3), n_iter=info['n_iter'])
print(json.dumps(res))
