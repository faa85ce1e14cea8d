#!/usr/bin/env python
"""Does the quad kernel's time depend on where its arrays land in HBM?  Within ONE process: allocate a dummy buffer of a
random size (shifts every later allocation), upload the file, run a few files, record the kernel's event time, free all -
repeated.  (The kernel's time differs by 10 % between boxes and between processes of one box: DESIGN.md section 4.)"""
import os, sys, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
from pgw4era5_amd.device import default_context
ctx = default_context()
case = synthetic.make_case(nlat=721, nlon=1440, nlev=137, seed=1, dtype=np.float64)
coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
rng = np.random.default_rng(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
rows = []
for trial in range(7):
    pad_mb = 0 if trial == 0 else int(rng.integers(1, 4000))
    pad = ctx.empty((pad_mb * 131072 + 1,), np.float64)
    deltas = s3.DeltaSet(ctx, case['deltas'], case['delta_times'], case['plev'], np.float64)
    era = s3._upload_era(ctx, case['era'], np.float64)
    out = {}
    ctx.profile(True)
    for i in range(2):
        s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out)
    ctx.sync(); ctx.profile_reset()
    for i in range(4):
        s3.process_file_device(ctx, era, coeffs, deltas, case['target_dt'], True, out=out)
    ctx.sync()
    c, ms = ctx.profile_get('quad_delta')
    c2, ms2 = ctx.profile_get('ps_loop_multi')
    rows.append(dict(trial=trial, pad_MB=pad_mb, quad_ms=round(ms / c, 4), loop_ms=round(ms2 / c2, 4),
                     T_in=hex(era['T'].ptr), T_out=hex(out['T'].ptr), U_out=hex(out['U'].ptr)))
    print(json.dumps(rows[-1]), flush=True)
    for v in list(era.values()) + list(out.values()):
        v.free()
    deltas.free(); pad.free()
