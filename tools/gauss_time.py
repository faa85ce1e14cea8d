#!/usr/bin/env python
"""Time of the ocean-grid interpolation (12 months of tos, 530 x 530 ocean grid -> 0.25 deg) incl. the GPU geometry."""
import os, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import functions as F, synthetic
from pgw4era5_amd.device import default_context
oc = synthetic.make_ocean_grid_case(nj=530, ni=530, ntime=12, seed=3)
lat = np.linspace(-90, 90, 721); lon = np.arange(1440) * 0.25
land = np.zeros((721, 1440))
F.gauss_interp_fields(land[:20, :30], lat[:20], lon[:30], oc['latitude'][:50, :50], oc['longitude'][:50, :50], list(oc['values'][:, :50, :50]), 1e6, 4.0)
ctx = default_context(); ctx.profile(True); ctx.profile_reset()
t0 = time.perf_counter()
out = F.gauss_interp_fields(land, lat, lon, oc['latitude'], oc['longitude'], list(oc['values']), 1e6, 4.0)
print('gauss_interp_fields 12 months: %.3f s wall, kernel %.1f ms, finite %.4f' % (time.perf_counter() - t0, ctx.profile_get('gauss_interp')[1], np.isfinite(out).mean()))
