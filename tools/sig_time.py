#!/usr/bin/env python
"""The signature-faithful kernels of bench.py's `signature_kernels` alone (integ_geopot, pressure, interp_logp), on the
bench file: a quick same-box A/B with alternative builds (`PGW_LIB=... python tools/sig_time.py [f32] [reps]`)."""
import json, os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
import bench
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
from pgw4era5_amd.device import default_context
dtype = np.float32 if (len(sys.argv) > 1 and sys.argv[1] == 'f32') else np.float64
reps = int(sys.argv[2]) if len(sys.argv) > 2 else 10
ctx = default_context()
case = synthetic.make_case(nlat=721, nlon=1440, nlev=137, seed=1, dtype=dtype)
coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
print('placement', ctx.enable_placement(138 * 721 * 1440 * 8, 12))           # PGW_PLACEMENT=plain for the A/B
era = s3._upload_era(ctx, case['era'], dtype)
class A: pass
bench.microbench(ctx, era, coeffs, A(), np, reps=30)       # clocks up (the card idles at 150 MHz)
m = bench.microbench(ctx, era, coeffs, A(), np, reps=reps)
print(os.environ.get('PGW_LIB', 'default').split('/')[-1], json.dumps({k: [v['avg_ms'], v['frac_of_peak']] for k, v in m.items()}))
