"""
step_03: impose the climate deltas on ERA5 files and re-balance surface pressure, on MI355X.

Mirror of the reference's `step_03_apply_to_era.py` (same `pgw_for_era5` signature, same
command-line flags, `-p` = number of worker ranks) with the per-file compute path running as
HIP kernels on device-resident arrays:

    RELHUM = q->RH(QV, pa(PS), T)                       reference step_03:64-94
    sea ice / skin / soil temperature riders            :103-146
    ta, hur, ua, va: time-lerp + surface insert + ln-p interpolation + add      :155-173
    fixed-point loop on delta_ps until max|phi error| <= 0.15                   :182-319
    PS, T, QV, U, V, T_SKIN, T_SO, FR_SEA_ICE written back                      :369-378

The twelve monthly records of every delta live in HBM for the whole run (`DeltaSet`; ~10 GB
fp64 at 0.25 deg for ta,hur,ua,va,zg on plev19 - the reference re-reads them from disk for
every file and `zg` once per iteration, functions.py:203 / step_03:292).
"""
import ctypes as C
import datetime as _dt
import os

import numpy as np

from . import _lib
from . import settings as S
from .constants import CON_G, CON_RD   # noqa: F401
from .device import DeviceArray, default_context, dtype_tag, ptr

_dp = C.POINTER(C.c_double)


# ----------------------------------------------------------------------------------------
# time bracketing of load_delta (reference functions.py:224-283), host control logic
# ----------------------------------------------------------------------------------------
def _to_dt64(t):
    return np.datetime64(t).astype('datetime64[s]')


def _with_year(t, year):
    s = str(t)
    return np.datetime64('%04d' % year + s[4:]).astype('datetime64[s]')


def delta_time_bracket(delta_times, target):
    """Indices and (re-yeared) stamps of the records bracketing `target`, periodic in the
    year; Feb-29 dropped first.  Returns (ind_before, ind_after, x_hi, x_new, keep) with
    x_* = float nanoseconds relative to the 'before' stamp (what xarray hands to scipy)."""
    times = np.asarray(delta_times).astype('datetime64[s]')
    target = _to_dt64(target)
    leap = None
    for i, t in enumerate(times):                                  # :224-230
        s = str(t)
        if s[5:7] == '02' and s[8:10] == '29':
            leap = i
    keep = np.array([i for i in range(len(times)) if i != leap], dtype=np.int64)
    year = int(str(target)[:4])
    ty = np.array([_with_year(t, year) for t in times[keep]])       # :235-238
    before = ty <= target                                           # :242-243
    if before.sum() > 0:
        ib = int(np.argwhere(before)[-1].squeeze()); tb = ty[ib]
    else:                                                           # :253-258
        ib = len(ty) - 1; tb = _with_year(ty[ib], year - 1)
    after = ty >= target                                            # :262-263
    if after.sum() > 0:
        ia = int(np.argwhere(after)[0].squeeze()); ta = ty[ia]
    else:                                                           # :273-278
        ia = 0; ta = _with_year(ty[ia], year + 1)
    ns = 'datetime64[ns]'
    x_hi = float((ta.astype(ns) - tb.astype(ns)).astype(np.int64))
    x_new = float((target.astype(ns) - tb.astype(ns)).astype(np.int64))
    if ib == ia:                                                    # :282-283
        x_hi = 0.0; x_new = 0.0
    return ib, ia, x_hi, x_new, keep


# ----------------------------------------------------------------------------------------
# deltas resident in HBM
# ----------------------------------------------------------------------------------------
class DeltaSet:
    """All records of all climate deltas of one run, on the device.

    arrays: dict var -> host array [nrec, (nplev,) nlat, nlon] for ta,hur,ua,va,zg (4-D) and
    tas,hurs,ts,tos,siconc,ps_hist (3-D).  `plev` in file order (descending for CMIP)."""

    VARS_3D = ('ta', 'hur', 'ua', 'va', 'zg')
    VARS_2D = ('tas', 'hurs', 'ts', 'tos', 'siconc', 'ps_hist')

    def __init__(self, ctx, arrays, delta_times, plev, dtype):
        self.ctx = ctx
        self.dtype = np.dtype(dtype)
        self.times = np.asarray(delta_times).astype('datetime64[s]')
        self.plev = np.ascontiguousarray(plev, dtype=np.float64)
        self.dev = {}
        for k in self.VARS_3D + self.VARS_2D:
            if k in arrays:
                a = arrays[k]
                self.dev[k] = a if isinstance(a, DeviceArray) else ctx.to_device(np.ascontiguousarray(a, dtype=self.dtype), self.dtype)
        # annual-mean skin-temperature delta (step_03:134-136): depends on the delta file only
        if 'ts' in arrays:
            ts = arrays['ts']
            ts_h = ts.numpy() if isinstance(ts, DeviceArray) else np.asarray(ts)
            _, _, _, _, keep = delta_time_bracket(self.times, self.times[0])
            self.ts_clim = ctx.to_device(ts_h[keep].astype(np.float64).mean(axis=0).astype(self.dtype), self.dtype)
        else:
            self.ts_clim = None

    def bracket(self, target):
        return delta_time_bracket(self.times, target)

    def lerp2d(self, name, target, out=None):
        """Time-interpolated 2-D delta (load_delta, functions.py:195-303) -> (1, nlat, nlon)."""
        ib, ia, x_hi, x_new, keep = self.bracket(target)
        a = self.dev[name]
        b = a.slab(int(keep[ib]))
        if out is None:
            out = self.ctx.empty((1,) + b.shape, self.dtype)
        if x_hi == 0.0:
            self.ctx._check(self.ctx.lib.pgw_memcpy_d2d(self.ctx.handle, out.ptr, b.ptr, b.nbytes))
        else:
            aa = a.slab(int(keep[ia]))
            self.ctx._check(self.ctx.lib.pgw_time_lerp(self.ctx.handle, dtype_tag(self.dtype), b.size, b.ptr, aa.ptr,
                                                       x_hi, x_new, out.ptr))
        return out


def _upload_era(ctx, era, dtype):
    out = {}
    for k in ('PS', 'FIS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_LAND', 'FR_SEA_ICE'):
        v = era[k]
        out[k] = v if isinstance(v, DeviceArray) else ctx.to_device(np.ascontiguousarray(v, dtype=dtype), dtype)
    return out


def process_file_device(ctx, era, coeffs, deltas, target_dt, ignore_top_pressure_error=False,
                        p_ref=None, out=None, want=('PS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_SEA_ICE')):
    """The per-file compute path of pgw_for_era5 (reference step_03:62-346, i_reinterp = 0,
    fixed p_ref) on device arrays.

    era: dict of DeviceArrays PS,FIS,(T_SKIN,FR_LAND,FR_SEA_ICE) (1,nlat,nlon); T,QV,U,V
    (1,N,nlat,nlon); T_SO (1,nsoil,nlat,nlon).  coeffs: dict ak,bk,[akm,bkm],soil1 (host).
    deltas: DeltaSet.  out: optional dict of preallocated output DeviceArrays (reused
    across files).  Returns (dict of DeviceArrays, info)."""
    lib, h = ctx.lib, ctx.handle
    p_ref = S.p_ref_inp if p_ref is None else p_ref
    dt = deltas.dtype
    tag = dtype_tag(dt)
    ctx.set_levels(coeffs['ak'], coeffs['bk'], coeffs.get('akm'), coeffs.get('bkm'))
    T, QV, PS = era['T'], era['QV'], era['PS']
    nt, N, nlat, nlon = T.shape
    ncol = nlat * nlon
    out = {} if out is None else out

    def buf(name, shape):
        if name not in out or out[name].shape != tuple(shape):
            out[name] = ctx.empty(shape, dt)
        return out[name]

    # relative humidity of the ERA state (step_03:91-94), pa = akm + PS*bkm in registers
    relhum = buf('_RELHUM', T.shape)
    ctx._check(lib.pgw_specific_to_relative_humidity_hybrid(h, tag, nt, ncol, QV.ptr, PS.ptr, T.ptr, relhum.ptr))

    # surface riders (step_03:103-146)
    if 'FR_SEA_ICE' in era and 'siconc' in deltas.dev:
        s3 = era['T_SKIN'].shape
        dsic = deltas.lerp2d('siconc', target_dt, buf('_dsic', s3))
        dts = deltas.lerp2d('ts', target_dt, buf('_dts', s3))
        dtos = deltas.lerp2d('tos', target_dt, buf('_dtos', s3))
        soil = np.ascontiguousarray(coeffs['soil1'], dtype=np.float64)
        nsoil = len(soil)
        ctx._check(lib.pgw_surface_update(
            h, tag, nt, ncol, nsoil, soil.ctypes.data_as(_dp),
            era['FR_SEA_ICE'].ptr, dsic.ptr, dtos.ptr, dts.ptr, era['FR_LAND'].ptr, deltas.ts_clim.ptr,
            era['T_SKIN'].ptr, era['T_SO'].ptr,
            buf('FR_SEA_ICE', era['FR_SEA_ICE'].shape).ptr, buf('_dts_comb', era['T_SKIN'].shape).ptr,
            buf('T_SKIN', era['T_SKIN'].shape).ptr, buf('T_SO', era['T_SO'].shape).ptr))

    # 3-D deltas onto model levels + add (step_03:155-173)
    ib, ia, x_hi, x_new, keep = deltas.bracket(target_dt)
    rb, ra = int(keep[ib]), int(keep[ia])
    plev = deltas.plev
    nplev = len(plev)
    era_field = dict(ta=T, hur=relhum, ua=era['U'], va=era['V'])
    out_name = dict(ta='T', hur='_hur_pgw', ua='U', va='V')
    for var in ('ta', 'hur', 'ua', 'va'):
        d = deltas.dev[var]
        db, da = d.slab(rb), d.slab(ra)
        if var in ('ta', 'hur'):                                   # functions.py:325-332
            sfc = deltas.dev[var + 's']
            psh = deltas.dev['ps_hist']
            sb, sa, pb, pa_ = sfc.slab(rb).ptr, sfc.slab(ra).ptr, psh.slab(rb).ptr, psh.slab(ra).ptr
        else:
            sb = sa = pb = pa_ = None
        o = buf(out_name[var], T.shape)
        ctx._check(lib.pgw_vert_interp_delta(
            h, tag, nt, nplev, N, ncol, plev.ctypes.data_as(_dp), db.ptr, da.ptr, x_hi, x_new,
            sb, sa, pb, pa_, None, PS.ptr, 1 if ignore_top_pressure_error else 0,
            era_field[var].ptr, o.ptr))

    # zg delta at p_ref (step_03:292-295: .sel(plev=p_ref), exact label match)
    kref = np.nonzero(plev == p_ref)[0]
    if len(kref) != 1:
        raise KeyError(p_ref)
    zg = deltas.dev['zg']
    zb, za = zg.slab(rb).slab(int(kref[0])), zg.slab(ra).slab(int(kref[0]))
    dzg = buf('_dzg', PS.shape)
    if x_hi == 0.0:
        ctx._check(lib.pgw_memcpy_d2d(h, dzg.ptr, zb.ptr, zb.nbytes))
    else:
        ctx._check(lib.pgw_time_lerp(h, tag, zb.size, zb.ptr, za.ptr, x_hi, x_new, dzg.ptr))

    # fixed-point loop (step_03:182-319)
    n_iter = C.c_int(0)
    hist = (C.c_double * int(S.max_n_iter))()
    ctx._check(lib.pgw_adjust_ps_loop(
        h, tag, nt, ncol, PS.ptr, era['FIS'].ptr, T.ptr, QV.ptr, out['T'].ptr, out['_hur_pgw'].ptr, dzg.ptr,
        float(p_ref), float(S.adj_factor), float(S.thresh_phi_ref_max_error), int(S.max_n_iter),
        buf('PS', PS.shape).ptr, buf('QV', T.shape).ptr, C.byref(n_iter), hist))
    info = dict(n_iter=n_iter.value, max_err=[hist[i] for i in range(n_iter.value)],
                levels_touched=int(lib.pgw_last_levels_touched(h)))
    return out, info


def pgw_for_era5_arrays(era, deltas, delta_times, plev, target_dt, ignore_top_pressure_error=False,
                        p_ref=None, dtype=None):
    """Whole-file path on in-memory host arrays (upload, compute on the GPU, download)."""
    ctx = default_context()
    if dtype is None:
        dtype = np.asarray(era['T']).dtype
    dtype = np.dtype(dtype)
    ds = DeltaSet(ctx, deltas, delta_times, plev, dtype)
    e = _upload_era(ctx, era, dtype)
    coeffs = dict(ak=era['ak'], bk=era['bk'], akm=era.get('akm'), bkm=era.get('bkm'), soil1=era['soil1'])
    out, info = process_file_device(ctx, e, coeffs, ds, target_dt, ignore_top_pressure_error, p_ref)
    res = {k: v.numpy() for k, v in out.items() if not k.startswith('_')}
    res['RELHUM_pgw'] = out['_hur_pgw'].numpy()
    res.update(info)
    return res
