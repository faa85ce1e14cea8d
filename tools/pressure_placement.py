#!/usr/bin/env python
"""The hybrid-pressure kernel (two pure write streams, pa_hl and pa) with its two outputs in ONE stretch of the card's memory
and in two (device.SpreadPool classes): is the 0.33 / 0.39 ms bimodality of bench.py's `signature_kernels.pressure` the
placement of its outputs?  Usage (GPU box): python tools/pressure_placement.py"""
import os, sys, json
import ctypes as C
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import numpy as np
from pgw4era5_amd import synthetic
from pgw4era5_amd.device import default_context, SpreadPool, dtype_tag

ctx = default_context()
nlat, nlon, N = 721, 1440, 137
ncol = nlat * nlon
case = synthetic.make_case(nlat=8, nlon=8, nlev=N, seed=1)
ctx.set_levels(case['era']['ak'], case['era']['bk'])
ps = ctx.to_device(np.full((1, nlat, nlon), 1.0e5) + 0.0)
pool = SpreadPool(ctx, (N + 1) * ncol * 8, 8)
print(pool.info)
a = [pool.take((1, N + 1, nlat, nlon), np.float64, cls=0) for _ in range(3)]
b = [pool.take((1, N + 1, nlat, nlon), np.float64, cls=1) for _ in range(3)]
print('classes', [x.placement_class for x in a + b])


def run(hl, pa, reps=20):
    for _ in range(3):
        ctx._check(ctx.lib.pgw_pressure_levels(ctx.handle, dtype_tag(np.float64), 1, ncol, ps.ptr, hl.ptr, pa.ptr))
    ctx.sync(); ctx._check(ctx.lib.pgw_timer_start(ctx.handle))
    for _ in range(reps):
        ctx._check(ctx.lib.pgw_pressure_levels(ctx.handle, dtype_tag(np.float64), 1, ncol, ps.ptr, hl.ptr, pa.ptr))
    ms = C.c_double(); ctx._check(ctx.lib.pgw_timer_stop(ctx.handle, C.byref(ms)))
    return ms.value / reps


gb = (2 * N + 2) * ncol * 8 / 1e9
for rep in range(3):
    for tag, hl, pa in (('same class 0', a[0], a[1]), ('classes 0 + 1', a[0], b[0]), ('same class 1', b[1], b[2]), ('classes 1 + 0', b[1], a[2])):
        ms = run(hl, pa)
        print('%-14s %.4f ms  %.0f GB/s  %.3f of peak' % (tag, ms, gb / ms * 1e3, gb / ms / 8.0))
