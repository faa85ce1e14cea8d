#!/usr/bin/env python
"""Arrays of 4 GiB and more (a 0.125 deg L137 float64 field is 4.5 GB): the file path on a grid of 2896 x 1440 = 4.17 M
columns, checked by a size-independent property - the columns are independent and the loop's stopping test is a global
maximum, so a file made of 16 copies of a 181 x 1440 file along latitude must give 16 copies of that file's result, bit for
bit, with the same pass count and max|err| history.  Exercises the 64-bit byte-offset instantiation of k_delta_quad as it is
chosen in production (no test knob), the 64-bit index arithmetic of every kernel of the path and grids of > 16 k blocks.
Needs ~90 GB of HBM and ~80 GB of host memory; one run takes a few minutes (most of it host-side array building)."""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3

COPIES = int(os.environ.get('BIG_COPIES', '16'))
NLAT, NLON, NLEV = 181, 1440, 137


def tile(arrays):
    out = {}
    for k, v in arrays.items():
        if isinstance(v, np.ndarray) and v.ndim >= 3 or (isinstance(v, np.ndarray) and v.ndim == 2 and v.shape == (NLAT, NLON)):
            reps = [1] * v.ndim
            reps[-2] = COPIES
            out[k] = np.tile(v, reps)
        else:
            out[k] = v
    return out


def main():
    t0 = time.time()
    c = synthetic.make_case(nlat=NLAT, nlon=NLON, nlev=NLEV, seed=7, dtype=np.float64)
    args = (c['delta_times'], c['plev'], c['target_dt'], True)
    small = s3.pgw_for_era5_arrays(c['era'], c['deltas'], *args)
    print('small: %d passes, %.1f s' % (small['n_iter'], time.time() - t0), flush=True)
    era, deltas = tile(c['era']), tile(c['deltas'])
    nbytes = era['T'].nbytes
    assert nbytes >= 1 << 32, nbytes
    print('big: %d columns, one 4-D field = %.2f GB, built after %.1f s' % (era['T'].shape[-2] * NLON, nbytes / 1e9, time.time() - t0),
          flush=True)
    big = s3.pgw_for_era5_arrays(era, deltas, *args)
    print('big done after %.1f s' % (time.time() - t0), flush=True)
    res = {'columns': int(era['T'].shape[-2] * NLON), 'field_GB': round(nbytes / 1e9, 3), 'copies': COPIES,
           'n_iter': [small['n_iter'], big['n_iter']], 'max_err_equal': small['max_err'] == big['max_err'], 'fields': {}}
    ok = res['n_iter'][0] == res['n_iter'][1] and res['max_err_equal']
    for k in ['PS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_SEA_ICE']:
        reps = [1] * small[k].ndim
        reps[-2] = COPIES
        same = bool(np.array_equal(np.tile(small[k], reps), big[k], equal_nan=True))
        res['fields'][k] = same
        ok = ok and same
    res['ok'] = ok
    print(json.dumps(res))
    return 0 if ok else 1


if __name__ == '__main__':
    sys.exit(main())
