#!/usr/bin/env python
"""Timing of the step_02 regridding kernel on BASELINE.json configs[3] (192 x 384 -> 721 x 1440, 19 levels x 12 months)."""
import os, sys, time, json
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import functions as F, synthetic
from pgw4era5_amd.device import default_context
ctx = default_context()
for dt in (np.float64, np.float32):
    g = synthetic.make_gcm_grid_case(nlat_src=192, nlon_src=384, nlat=721, nlon=1440, nplev=19, ntime=12, seed=4, dtype=dt)
    src = ctx.to_device(g['field'], dt)
    F.regrid_field(src, g['src_lat'], g['src_lon'], g['targ_lat'], g['targ_lon']).free()
    ctx.profile(True); ctx.profile_reset()
    for _ in range(5):
        F.regrid_field(src, g['src_lat'], g['src_lon'], g['targ_lat'], g['targ_lon']).free()
    ctx.sync()
    cnt, ms = ctx.profile_get('regrid')
    ctx.profile(False)
    nbytes = (12 * 19 * 721 * 1440 + g['field'].size) * np.dtype(dt).itemsize
    print(json.dumps(dict(dtype=np.dtype(dt).name, kernel_ms=round(ms / cnt, 4), GBps=round(nbytes / 1e9 / (ms / cnt / 1e3), 1))))
    src.free()
