#!/usr/bin/env python
"""where a float32 reference-mode file's time goes beside its kernels: wall per file vs the per-kernel event sum"""
import os, sys, json, time, datetime as dt
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
from pgw4era5_amd.device import default_context
ctx = default_context()
case = synthetic.make_case(nlat=721, nlon=1440, nlev=137, seed=1, dtype=np.float32)
deltas = s3.DeltaSet(ctx, case['deltas'], case['delta_times'], case['plev'], np.float32)
era = s3._upload_era(ctx, case['era'], np.float32)
coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
for ref in (True, False, True):
    out = {}
    for prof in (True, False):
        ctx.profile(prof)
        for i in range(2):
            s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out, ref_dtype=ref)
        ctx.sync(); ctx.profile_reset()
        walls, infos = [], []
        for i in range(8):
            t0 = time.perf_counter()
            _, info = s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'] + dt.timedelta(hours=i), True, out=out, ref_dtype=ref)
            walls.append(round((time.perf_counter() - t0) * 1e3, 3)); infos.append((info['n_iter'], info['passes_launched']))
        ctx.sync()
        ks = {k: ctx.profile_get(k) for k in ('quad_delta', 'ps_loop_multi', 'finalize', 'surface')} if prof else {}
        print(json.dumps(dict(ref=ref, profile=prof, walls=walls, infos=infos[:3], kernel_sum=round(sum(ms / max(c, 1) for c, ms in ks.values()), 3) if prof else None)))
    for v in out.values():
        v.free()
