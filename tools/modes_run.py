#!/usr/bin/env python
"""The reference's two other operating modes on the bench.py file, HBM-resident, for the profiler: settings.p_ref_inp = None
(the LOCAL form of k_ps_loop_multi) and settings.i_reinterp = 1 (k_reinterp_pair in every pass), 2 warm-up + 4 timed files
each.  `bash tools/prof_run.sh <tag>` runs this under rocprofv3 for profiles/kernel_stats_<tag>_modes.csv and
pmc_summary_<tag>_modes.json; bench.py reports the same legs as extras.local_p_ref / extras.i_reinterp.

    python tools/modes_run.py [--storage f64|f32] [--f32-mode reference|fast]
"""
import argparse
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def main():
    p = argparse.ArgumentParser()
    p.add_argument('--storage', choices=['f64', 'f32'], default='f64')
    p.add_argument('--f32-mode', choices=['reference', 'fast'], default='reference')
    p.add_argument('--nlat', type=int, default=721)
    p.add_argument('--nlon', type=int, default=1440)
    p.add_argument('--nlev', type=int, default=137)
    a = p.parse_args()
    import numpy as np
    import bench
    from pgw4era5_amd import synthetic, settings as S, step_03_apply_to_era as s3
    from pgw4era5_amd.device import default_context
    dtype = np.float64 if a.storage == 'f64' else np.float32
    S.f32_file_mode = a.f32_mode
    ctx = default_context()
    case = synthetic.make_case(nlat=a.nlat, nlon=a.nlon, nlev=a.nlev, seed=1, dtype=dtype)
    placement = ctx.enable_placement(int(np.prod(case['era']['T'].shape)) * 8, 13)      # inputs 4, outputs 4 per leg, workspace
    deltas = s3.DeltaSet(ctx, case['deltas'], case['delta_times'], case['plev'], dtype)
    era = s3._upload_era(ctx, case['era'], dtype)
    coeffs = dict(ak=case['era']['ak'], bk=case['era']['bk'], soil1=case['era']['soil1'])
    out = {k: bench.mode_leg(ctx, era, coeffs, deltas, case, a, which) for k, which in (('local_p_ref', 'local'), ('i_reinterp', 'reinterp'))}
    out['placement'] = placement if placement is not None else {'mode': 'plain'}
    print(json.dumps(out))


if __name__ == '__main__':
    main()
