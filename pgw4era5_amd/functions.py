"""
Host-side mirror of the reference's `functions.py` numerical API, backed by the HIP library.

Every function keeps the reference's name, argument order and error behaviour
(reference functions.py line ranges are cited per function) and runs on the GPU through the
C-ABI in include/pgw_hip.h.  There is no CPU implementation behind these names.

Accepted array kinds, everywhere a field is expected:
  * `numpy.ndarray` (float32/float64)      -> copied to the device, result returned as ndarray
  * `pgw4era5_amd.device.DeviceArray`      -> used in place, result stays on the device
  * labelled arrays (`pgw4era5_amd.ncio.Field`, or any object with `.values`, `.dims`,
    `.coords`)                             -> like ndarray, result re-wrapped with the labels
4-D fields are C-order `(time, level, lat, lon)`.
"""
import ctypes as C
import datetime as _dt
import os

import numpy as np

from . import _lib
from .constants import CON_G, CON_RD, CON_MW_MD   # noqa: F401  (re-exported like the reference)
from .device import DeviceArray, default_context, dtype_tag, ptr
from .settings import (                      # noqa: F401
    i_debug, i_use_xesmf_regridding, file_name_bases,
    TIME_ERA, LEV_ERA, HLEV_ERA, LON_ERA, LAT_ERA,
    TIME_GCM, PLEV_GCM, LON_GCM, LAT_GCM, LON_GCM_OCEAN, LAT_GCM_OCEAN,
)

_dp = C.POINTER(C.c_double)
_ip = C.POINTER(C.c_int)


# ------------------------------------------------------------------------------- helpers
def _is_labelled(x):
    return hasattr(x, 'values') and hasattr(x, 'dims') and not isinstance(x, DeviceArray)


def _raw(x):
    """ndarray / DeviceArray behind any accepted input."""
    if isinstance(x, DeviceArray):
        return x
    if _is_labelled(x):
        return np.asarray(x.values)
    return np.asarray(x)


def _common_dtype(*xs):
    for x in xs:
        if x is None:
            continue
        r = _raw(x)
        if r.dtype == np.float64:
            return np.dtype('float64')
        if r.dtype not in (np.dtype('float32'),):
            if not isinstance(r, DeviceArray) and r.dtype.kind in 'iu':
                return np.dtype('float64')
    return np.dtype('float32')


def _dev(ctx, x, dtype, shape=None):
    """Device array of `x` in `dtype` (no copy if it already is one of that dtype)."""
    if x is None:
        return None
    r = _raw(x)
    if isinstance(r, DeviceArray):
        if r.dtype != dtype:
            raise TypeError('device arrays of mixed dtype: got %s, expected %s' % (r.dtype, dtype))
        return r if shape is None else r.view(shape)
    a = np.ascontiguousarray(r, dtype=dtype)
    if shape is not None:
        a = a.reshape(shape)
    return ctx.to_device(a, dtype)


def _out(ctx, dev, like):
    """Return `dev` in the kind of `like`: DeviceArray as it is; a labelled array re-wrapped with `like`'s dimension
    names and coordinates - `ncio.Field.like(data)`, or `.copy(data=...)` of an `xarray.DataArray` (and anything else
    that offers it), so that the reference's own next line, e.g. `.transpose(TIME_ERA, LEV_ERA, LAT_ERA, LON_ERA)`
    (step_03_apply_to_era.py:91-94), keeps working; a plain ndarray otherwise."""
    if isinstance(like, DeviceArray):
        return dev
    host = dev.numpy()
    if _is_labelled(like):
        if hasattr(like, 'like'):
            return like.like(host)
        if hasattr(like, 'copy') and tuple(getattr(like, 'shape', ())) == host.shape:
            try:
                return like.copy(data=host)
            except TypeError:
                pass
    return host


def _aligned(x, like):
    """A labelled operand whose dimensions are those of `like` in another order is transposed to `like`'s order (xarray
    aligns operands by dimension NAME; the kernels take positions).  Anything else passes through."""
    if x is None or not (_is_labelled(x) and _is_labelled(like)):
        return x
    dx, dl = tuple(x.dims), tuple(like.dims)
    if dx != dl and sorted(dx) == sorted(dl) and hasattr(x, 'transpose'):
        return x.transpose(*dl)
    return x


def _shape4(x):
    s = _raw(x).shape
    if len(s) != 4:
        raise ValueError('expected a 4-D (time, level, lat, lon) array, got shape %s' % (s,))
    return s


# ------------------------------------------------------------------------------- humidity
def specific_to_relative_humidity(hus, pa, ta):
    """RH [%] from specific humidity (IFS 7.92/7.93).  reference functions.py:107-116."""
    ctx = default_context()
    pa, ta = _aligned(pa, hus), _aligned(ta, hus)
    dt = _common_dtype(hus, pa, ta)
    shp = _raw(hus).shape
    dh, dp_, dt_ = _dev(ctx, hus, dt), _dev(ctx, np.broadcast_to(_raw(pa), shp) if not isinstance(_raw(pa), DeviceArray) else pa, dt), _dev(ctx, ta, dt)
    out = ctx.empty(shp, dt)
    ctx._check(ctx.lib.pgw_specific_to_relative_humidity(ctx.handle, dtype_tag(dt), out.size, dh.ptr, dp_.ptr, dt_.ptr, out.ptr))
    return _out(ctx, out, hus)


def relative_to_specific_humidity(hur, pa, ta):
    """Specific humidity from RH [%].  reference functions.py:118-125."""
    ctx = default_context()
    pa, ta = _aligned(pa, hur), _aligned(ta, hur)
    dt = _common_dtype(hur, pa, ta)
    shp = _raw(hur).shape
    dh, dp_, dt_ = _dev(ctx, hur, dt), _dev(ctx, np.broadcast_to(_raw(pa), shp) if not isinstance(_raw(pa), DeviceArray) else pa, dt), _dev(ctx, ta, dt)
    out = ctx.empty(shp, dt)
    ctx._check(ctx.lib.pgw_relative_to_specific_humidity(ctx.handle, dtype_tag(dt), out.size, dh.ptr, dp_.ptr, dt_.ptr, out.ptr))
    return _out(ctx, out, hur)


def _humidity_leaf(which, a, b, like):
    ctx = default_context()
    dt = _common_dtype(a) if b is None else _common_dtype(a, b)
    shp = _raw(a).shape
    da = _dev(ctx, a, dt)
    db = None
    if b is not None:
        b = _aligned(b, a)
        db = _dev(ctx, np.broadcast_to(_raw(b), shp) if not isinstance(_raw(b), DeviceArray) else b, dt)
    out = ctx.empty(shp, dt)
    ctx._check(ctx.lib.pgw_humidity_leaf(ctx.handle, dtype_tag(dt), which, out.size, da.ptr, db.ptr if db is not None else None, out.ptr))
    return _out(ctx, out, like)


def specific_humidity_to_vapor_pressure(hus, pa):
    """e = hus * pa / (0.622 + 0.378 * hus).  reference functions.py:58-64."""
    return _humidity_leaf(0, hus, pa, hus)


def vapor_pressure_to_specific_humidity(vapp, pa):
    """hus = 0.622 * vapp / (pa - 0.378 * vapp).  reference functions.py:66-72."""
    return _humidity_leaf(1, vapp, pa, vapp)


def saturation_vapor_pressure_water_or_ice(pa, ta, water=True):
    """IFS (7.93) saturation vapour pressure over water or over ice; `pa` is unused, as in the reference (functions.py:74-89)."""
    return _humidity_leaf(2 if water else 3, ta, None, ta)


def saturation_vapor_pressure_water_and_ice(pa, ta):
    """IFS (7.92) mixed-phase saturation vapour pressure.  reference functions.py:91-105."""
    return _humidity_leaf(4, ta, None, ta)


def dt64_to_dt(dt64):
    """numpy datetime64 -> python datetime (UTC).  reference functions.py:38-51."""
    import datetime as _datetime
    timestamp = (np.datetime64(dt64, 's') - np.datetime64('1970-01-01T00:00:00')) / np.timedelta64(1, 's')
    return _datetime.datetime.utcfromtimestamp(float(timestamp))


# ------------------------------------------------------------------------------- pressure
def hybrid_pressure(ak, bk, ps, akm=None, bkm=None):
    """pa_hl = ak + ps*bk, pa = akm + ps*bkm  (reference step_03_apply_to_era.py:64-88,196-199;
    this is what BASELINE.json calls "integ_pressure").  ps (time, lat, lon) -> (pa_hl, pa)."""
    ctx = default_context()
    ctx.set_levels(ak, bk, akm, bkm)
    dt = _common_dtype(ps)
    s = _raw(ps).shape
    if len(s) != 3:
        raise ValueError('ps must be (time, lat, lon)')
    nt, ncol, n = s[0], s[1] * s[2], ctx.nlev
    dps = _dev(ctx, ps, dt)
    # the kernel's two write streams in different stretches of the card's memory when the context places its level arrays
    # (settings.placement; Context.level_array falls back to plain memory): 0.34 instead of 0.41 ms at 0.25 deg L137
    pa_hl = ctx.level_array((nt, n + 1, s[1], s[2]), dt, cls=0)
    pa = ctx.level_array((nt, n, s[1], s[2]), dt, cls=1)
    ctx._check(ctx.lib.pgw_pressure_levels(ctx.handle, dtype_tag(dt), nt, ncol, dps.ptr, pa_hl.ptr, pa.ptr))
    if isinstance(ps, DeviceArray):
        return pa_hl, pa
    return pa_hl.numpy(), pa.numpy()


# ------------------------------------------------------------------------------- integ_geopot
def integ_geopot(pa_hl, zgs, ta, hus, level1, p_ref, full_column=True):
    """Geopotential at p_ref by hydrostatic integration from the surface.
    reference functions.py:128-189.  `level1` = half-level labels (its length must be N+1).
    p_ref: scalar or (time, lat, lon) field.  Returns (time, lat, lon)."""
    ctx = default_context()
    hus = _aligned(hus, ta)
    s = _shape4(pa_hl)
    st = _shape4(ta)
    if len(level1) != s[1] or st[1] != s[1] - 1 or _shape4(hus) != st:
        raise ValueError('level dimensions are inconsistent')
    dt = _common_dtype(pa_hl, zgs, ta, hus)
    nt, n, ncol = s[0], st[1], s[2] * s[3]
    d_p, d_z, d_t, d_q = _dev(ctx, pa_hl, dt), _dev(ctx, zgs, dt, (nt, s[2], s[3])), _dev(ctx, ta, dt), _dev(ctx, hus, dt)
    pref_field = None
    pref_scalar = 0.0
    pr = _raw(p_ref) if not np.isscalar(p_ref) else None
    if pr is not None and (isinstance(pr, DeviceArray) or pr.ndim > 0):
        pref_field = _dev(ctx, p_ref, dt, (nt, s[2], s[3]))
    else:
        pref_scalar = float(p_ref)
    out = ctx.empty((nt, s[2], s[3]), dt)
    ctx._check(ctx.lib.pgw_integ_geopot(ctx.handle, dtype_tag(dt), nt, n, ncol, d_p.ptr, d_z.ptr, d_t.ptr, d_q.ptr,
                                        pref_scalar, ptr(pref_field), out.ptr, 1 if full_column else 0))
    return _out(ctx, out, zgs)


# ------------------------------------------------------------------------------- interpolation
def _check_extrapolate(extrapolate):
    if extrapolate not in _lib.EXTRAP:
        raise ValueError('Invalid input value for "extrapolate"')
    return _lib.EXTRAP[extrapolate]


def interp_logp_4d(var, source_P, targ_P, extrapolate='off', time_key=None, lat_key=None, lon_key=None):
    """Column-wise linear interpolation in ln(p).  reference functions.py:434-477.
    var, source_P (time, S, lat, lon); targ_P (time, N, lat, lon) -> (time, N, lat, lon)."""
    mode = _check_extrapolate(extrapolate)
    sv, ss, st = _shape4(var), _shape4(source_P), _shape4(targ_P)
    if (sv[0] != ss[0]) or (sv[0] != st[0]):
        raise ValueError('Time dimension of input files is inconsistent!')
    if (sv[2] != ss[2]) or (sv[2] != st[2]):
        raise ValueError('Lat dimension of input files is inconsistent!')
    if (sv[3] != ss[3]) or (sv[3] != st[3]):
        raise ValueError('Lon dimension of input files is inconsistent!')
    if sv[1] != ss[1]:
        raise ValueError('Level dimension of var and source_P is inconsistent!')
    ctx = default_context()
    dt = _common_dtype(var, source_P, targ_P)
    d_v, d_s, d_t = _dev(ctx, var, dt), _dev(ctx, source_P, dt), _dev(ctx, targ_P, dt)
    out = ctx.empty(st, dt)
    ctx._check(ctx.lib.pgw_interp_logp_4d(ctx.handle, dtype_tag(dt), st[0], sv[1], st[1], st[2] * st[3],
                                          d_v.ptr, d_s.ptr, d_t.ptr, mode, 0, out.ptr))
    return _out(ctx, out, targ_P)


def interp_1d_for_timelatlon(orig_array, src_p, targ_p, interp_array, ntime, nlat, nlon, extrapolate):
    """reference functions.py:479-508: inputs already hold ln(p); fills `interp_array` in place."""
    mode = _check_extrapolate(extrapolate)
    ctx = default_context()
    dt = np.dtype('float64')
    d_v, d_s, d_t = _dev(ctx, orig_array, dt), _dev(ctx, src_p, dt), _dev(ctx, targ_p, dt)
    out = ctx.empty(d_t.shape, dt)
    ctx._check(ctx.lib.pgw_interp_logp_4d(ctx.handle, dtype_tag(dt), ntime, d_s.shape[1], d_t.shape[1], nlat * nlon,
                                          d_v.ptr, d_s.ptr, d_t.ptr, mode, 1, out.ptr))
    interp_array[...] = out.numpy()


def interp_extrap_1d(src_x, src_y, targ_x, extrapolate):
    """reference functions.py:511-580 for one column (abscissae as given, e.g. ln p)."""
    mode = _check_extrapolate(extrapolate)
    ctx = default_context()
    dt = np.dtype('float64')
    S, N = len(src_x), len(targ_x)
    d_s = _dev(ctx, np.asarray(src_x, dtype=dt).reshape(1, S, 1, 1), dt)
    d_v = _dev(ctx, np.asarray(src_y, dtype=dt).reshape(1, S, 1, 1), dt)
    d_t = _dev(ctx, np.asarray(targ_x, dtype=dt).reshape(1, N, 1, 1), dt)
    out = ctx.empty((1, N, 1, 1), dt)
    rc = ctx.lib.pgw_interp_logp_4d(ctx.handle, dtype_tag(dt), 1, S, N, 1, d_v.ptr, d_s.ptr, d_t.ptr, mode, 1, out.ptr)
    # the 1-D function has no ascending pre-check (that lives in interp_1d_for_timelatlon)
    if rc in (10, 11):
        rc = 0
    ctx._check(rc)
    return out.numpy().reshape(N)


# ------------------------------------------------------------------------------- deltas
def time_lerp(v_before, v_after, x_hi, x_new):
    """(v_after - v_before)/x_hi * x_new + v_before: the arithmetic under load_delta's
    `.interp(time=...)` (reference functions.py:288-292; scipy interp1d linear)."""
    ctx = default_context()
    dt = _common_dtype(v_before, v_after)
    d_b, d_a = _dev(ctx, v_before, dt), _dev(ctx, v_after, dt)
    out = ctx.empty(d_b.shape, dt)
    ctx._check(ctx.lib.pgw_time_lerp(ctx.handle, dtype_tag(dt), out.size, d_b.ptr, d_a.ptr, float(x_hi), float(x_new), out.ptr))
    return _out(ctx, out, v_before)


def replace_delta_sfc(source_P, ps_hist, delta, delta_sfc):
    """reference functions.py:343-366 for one ascending-pressure column."""
    ctx = default_context()
    dt = np.dtype('float64')
    P = np.ascontiguousarray(source_P, dtype=dt)
    S = len(P)
    d_d = _dev(ctx, np.asarray(delta, dtype=dt).reshape(1, S, 1), dt)
    d_s = _dev(ctx, np.asarray([[delta_sfc]], dtype=dt), dt)
    d_p = _dev(ctx, np.asarray([[ps_hist]], dtype=dt), dt)
    oP, oD = ctx.empty((1, S, 1), dt), ctx.empty((1, S, 1), dt)
    ctx._check(ctx.lib.pgw_replace_delta_sfc(ctx.handle, dtype_tag(dt), 1, S, 1, P.ctypes.data_as(_dp),
                                             d_d.ptr, d_s.ptr, d_p.ptr, oP.ptr, oD.ptr))
    return oP.numpy().reshape(S), oD.numpy().reshape(S)


def _plev_of(delta, plev):
    if plev is not None:
        return np.ascontiguousarray(plev, dtype=np.float64)
    if _is_labelled(delta) and PLEV_GCM in getattr(delta, 'coords', {}):
        return np.ascontiguousarray(delta.coords[PLEV_GCM], dtype=np.float64)
    raise ValueError('vert_interp_delta needs the plev coordinate (labelled delta or plev=...)')


def vert_interp_delta(delta, target_P, delta_sfc=None, ps_hist=None, ignore_top_pressure_error=False,
                      plev=None, add_to=None, _bracket=None):
    """Vertical interpolation of a climate delta onto model levels, with the surface delta
    inserted at the HIST surface pressure.  reference functions.py:369-431 (+ :343-366).
    delta (time, plev, lat, lon) in the file's plev order (reversed inside like :383-384);
    target_P (time, N, lat, lon); delta_sfc, ps_hist (time, lat, lon) or None."""
    ctx = default_context()
    pl = _plev_of(delta, plev)
    sd, st = _shape4(delta), _shape4(target_P)
    if sd[0] != st[0] or sd[2:] != st[2:]:
        raise ValueError()
    if (delta_sfc is None) != (ps_hist is None):
        raise ValueError('delta_sfc and ps_hist must be given together')
    dt = _common_dtype(delta, target_P, delta_sfc, ps_hist, add_to)
    nt, S, ncol, N = sd[0], sd[1], sd[2] * sd[3], st[1]
    d_d, d_t = _dev(ctx, delta, dt), _dev(ctx, target_P, dt)
    d_s = _dev(ctx, delta_sfc, dt, (nt, sd[2], sd[3])) if delta_sfc is not None else None
    d_p = _dev(ctx, ps_hist, dt, (nt, sd[2], sd[3])) if ps_hist is not None else None
    d_add = _dev(ctx, add_to, dt) if add_to is not None else None
    out = ctx.empty(st, dt)
    ctx._check(ctx.lib.pgw_vert_interp_delta(
        ctx.handle, dtype_tag(dt), nt, S, N, ncol, pl.ctypes.data_as(_dp),
        d_d.ptr, None, 0.0, 0.0, ptr(d_s), None, ptr(d_p), None,
        d_t.ptr, None, 1 if ignore_top_pressure_error else 0, ptr(d_add), out.ptr))
    return _out(ctx, out, target_P)


def determine_p_ref(p_min_era, p_min_pgw, p_ref_opts, p_ref_last=None):
    """reference functions.py:583-598 (scalar control logic; host side like the reference)."""
    for p in p_ref_opts:
        if (p_min_era > p) & (p_min_pgw > p):
            if p_ref_last is None:
                return p
            return min(p, p_ref_last)


def integrate_tos(tos_field, ts_field, land_frac, ice_frac):
    """Blend SST and skin-temperature deltas by land + sea-ice fraction.
    reference functions.py:1145-1186."""
    ctx = default_context()
    dt = _common_dtype(tos_field, ts_field, land_frac, ice_frac)
    shp = _raw(tos_field).shape
    d = [_dev(ctx, x, dt) for x in (tos_field, ts_field, land_frac, ice_frac)]
    out = ctx.empty(shp, dt)
    ctx._check(ctx.lib.pgw_integrate_tos(ctx.handle, dtype_tag(dt), out.size, d[0].ptr, d[1].ptr, d[2].ptr, d[3].ptr, out.ptr))
    return _out(ctx, out, tos_field)


# ------------------------------------------------------------------------------- ps loop
def adjust_ps_loop(ak, bk, PS, FIS, T, QV, ta_pgw, hur_pgw, dzg_pref, akm=None, bkm=None,
                   p_ref=None, adj_factor=None, thresh=None, max_n_iter=None, want_hus=True):
    """The iterative surface-pressure adjustment, reference step_03_apply_to_era.py:182-319
    (fixed p_ref).  Returns dict(ps_pgw, hus_pgw, n_iter, max_err).  Raises the reference's
    ValueError on non-convergence."""
    from . import settings as S
    p_ref = S.p_ref_inp if p_ref is None else p_ref
    adj_factor = S.adj_factor if adj_factor is None else adj_factor
    thresh = S.thresh_phi_ref_max_error if thresh is None else thresh
    max_n_iter = S.max_n_iter if max_n_iter is None else max_n_iter
    ctx = default_context()
    ctx.set_levels(ak, bk, akm, bkm)
    s = _shape4(ta_pgw)
    dt = _common_dtype(PS, FIS, T, QV, ta_pgw, hur_pgw, dzg_pref)
    nt, ncol = s[0], s[2] * s[3]
    s3 = (nt, s[2], s[3])
    d = dict(PS=_dev(ctx, PS, dt, s3), FIS=_dev(ctx, FIS, dt, s3), T=_dev(ctx, T, dt), QV=_dev(ctx, QV, dt),
             ta=_dev(ctx, ta_pgw, dt), hur=_dev(ctx, hur_pgw, dt), dzg=_dev(ctx, dzg_pref, dt, s3))
    ps_out = ctx.empty(s3, dt)
    hus_out = ctx.empty(s, dt) if want_hus else None
    n_iter = C.c_int(0)
    hist = (C.c_double * int(max_n_iter))()
    rc = ctx.lib.pgw_adjust_ps_loop(ctx.handle, dtype_tag(dt), nt, ncol, d['PS'].ptr, d['FIS'].ptr, d['T'].ptr,
                                    d['QV'].ptr, d['ta'].ptr, d['hur'].ptr, d['dzg'].ptr, float(p_ref),
                                    float(adj_factor), float(thresh), int(max_n_iter), ps_out.ptr, ptr(hus_out),
                                    C.byref(n_iter), hist)
    ctx._check(rc)
    res = dict(n_iter=n_iter.value, max_err=[hist[i] for i in range(n_iter.value)],
               levels_touched=int(ctx.lib.pgw_last_levels_touched(ctx.handle)))
    if isinstance(ta_pgw, DeviceArray):
        res.update(ps_pgw=ps_out, hus_pgw=hus_out)
    else:
        res.update(ps_pgw=ps_out.numpy(), hus_pgw=hus_out.numpy() if want_hus else None)
    return res


# ------------------------------------------------------------------------------- regridding
def regrid_tables(src_lat, src_lon, targ_lat, targ_lon):
    """Index/weight tables for the separable lat-then-lon linear interpolation of
    regrid_lat_lon's xarray branch (reference functions.py:774-789, 817-893), including its
    pole rows, periodic +-360 extension and scipy-interp1d index rule (searchsorted-left,
    clip to [1, n-1]).  Raises the reference's ValueErrors for uncovered targets."""
    src_lat = np.asarray(src_lat, dtype=np.float64)
    src_lon = np.asarray(src_lon, dtype=np.float64)
    targ_lat = np.asarray(targ_lat, dtype=np.float64)
    targ_lon = np.asarray(targ_lon, dtype=np.float64)
    nlat_s, nlon_s = len(src_lat), len(src_lon)
    dlon = np.median(np.diff(src_lon))                          # :778
    dlat = np.median(np.diff(src_lat))                          # :779 (before the flip)
    periodic = (dlon + np.max(src_lon) - np.min(src_lon)) >= 359.9     # :780-789
    rows = np.arange(nlat_s)
    lat = src_lat
    if lat[0] > lat[-1]:                                        # :822-829
        lat = lat[::-1]
        rows = rows[::-1]
    south_row = north_row = -1
    if np.max(targ_lat) + dlat > 89.9:                          # :833-837
        north_row = int(rows[-1])
        lat = np.concatenate([lat, [90.0]])
        rows = np.concatenate([rows, [nlat_s]])
    if np.min(targ_lat) - dlat < -89.9:                         # :838-842
        south_row = int(rows[0])
        lat = np.concatenate([[-90.0], lat])
        rows = np.concatenate([[-1], rows])
    if (np.max(targ_lat) > np.max(lat)) | (np.min(targ_lat) < np.min(lat)):      # :845-856
        raise ValueError('ERA5 dataset extends further North or South than GCM dataset!. Perhaps consider '
                         'using ERA5 on a subdomain only if global coverage is not required?')
    order = np.argsort(lat, kind='stable')                      # xarray sorts before interp
    lat, rows = lat[order], rows[order]
    idx = np.searchsorted(lat, targ_lat).clip(1, len(lat) - 1)
    lat_lo, lat_hi = rows[idx - 1].astype(np.int32), rows[idx].astype(np.int32)
    lat_dx = targ_lat - lat[idx - 1]
    lat_Dx = lat[idx] - lat[idx - 1]
    lat_oob = ((targ_lat < lat[0]) | (targ_lat > lat[-1])).astype(np.int32)

    lon = src_lon
    cols = np.arange(nlon_s)
    if periodic:                                                # :866-874
        if np.max(targ_lon) > np.max(lon):
            lon = np.concatenate([lon, src_lon + 360])
            cols = np.concatenate([cols, np.arange(nlon_s)])
        if np.min(targ_lon) < np.min(lon):
            lon = np.concatenate([lon - 360, lon])
            cols = np.concatenate([cols, cols])
    if (np.max(targ_lon) > np.max(lon)) | (np.min(targ_lon) < np.min(lon)):      # :877-888
        raise ValueError('ERA5 dataset extends further East or West than GCM dataset!. Perhaps consider '
                         'using ERA5 on a subdomain only if global coverage is not required?')
    order = np.argsort(lon, kind='stable')
    lon, cols = lon[order], cols[order]
    idx = np.searchsorted(lon, targ_lon).clip(1, len(lon) - 1)
    lon_lo, lon_hi = cols[idx - 1].astype(np.int32), cols[idx].astype(np.int32)
    lon_dx = targ_lon - lon[idx - 1]
    lon_Dx = lon[idx] - lon[idx - 1]
    lon_oob = ((targ_lon < lon[0]) | (targ_lon > lon[-1])).astype(np.int32)
    return dict(lat_lo=lat_lo, lat_hi=lat_hi, lat_dx=lat_dx, lat_Dx=lat_Dx, lat_oob=lat_oob,
                lon_lo=lon_lo, lon_hi=lon_hi, lon_dx=lon_dx, lon_Dx=lon_Dx, lon_oob=lon_oob,
                south_row=south_row, north_row=north_row, periodic=bool(periodic))


def regrid_field(field, src_lat, src_lon, targ_lat, targ_lon):
    """Bilinear (lat, then lon) regridding of field (..., nlat_s, nlon_s) on the GPU."""
    ctx = default_context()
    tb = regrid_tables(src_lat, src_lon, targ_lat, targ_lon)
    r = _raw(field)
    shp = r.shape
    dt = _common_dtype(field)
    nlat_s, nlon_s = shp[-2], shp[-1]
    if nlat_s != len(src_lat) or nlon_s != len(src_lon):
        raise ValueError('field shape does not match the source coordinates')
    nfield = int(np.prod(shp[:-2], dtype=np.int64)) if len(shp) > 2 else 1
    d_src = _dev(ctx, field, dt)
    out = ctx.empty(shp[:-2] + (len(targ_lat), len(targ_lon)), dt)
    c = {k: np.ascontiguousarray(v) for k, v in tb.items() if isinstance(v, np.ndarray)}
    ctx._check(ctx.lib.pgw_regrid_bilinear(
        ctx.handle, dtype_tag(dt), nfield, nlat_s, nlon_s, len(targ_lat), len(targ_lon), d_src.ptr,
        c['lat_lo'].ctypes.data_as(_ip), c['lat_hi'].ctypes.data_as(_ip), c['lat_dx'].ctypes.data_as(_dp),
        c['lat_Dx'].ctypes.data_as(_dp), c['lat_oob'].ctypes.data_as(_ip),
        c['lon_lo'].ctypes.data_as(_ip), c['lon_hi'].ctypes.data_as(_ip), c['lon_dx'].ctypes.data_as(_dp),
        c['lon_Dx'].ctypes.data_as(_dp), c['lon_oob'].ctypes.data_as(_ip),
        tb['south_row'], tb['north_row'], out.ptr))
    return _out(ctx, out, field)


# ------------------------------------------------------------------------------- delta files
_DATASET_CACHE = {}


def _open_cached(path):
    """Delta files are read once per process (the reference re-opens them per call,
    functions.py:203-204, i.e. ~14x per ERA5 file plus once per iteration)."""
    from . import ncio
    key = os.path.abspath(path)
    st = os.stat(key)
    hit = _DATASET_CACHE.get(key)
    if hit is None or hit[0] != (st.st_mtime_ns, st.st_size):
        _DATASET_CACHE[key] = ((st.st_mtime_ns, st.st_size), ncio.open_dataset(key))
    return _DATASET_CACHE[key][1]


def load_delta(delta_input_dir, var_name, era5_date_time, target_date_time=None,
               name_base=file_name_bases['SCEN-HIST']):
    """Load a climate delta and, if target_date_time is given, interpolate it linearly to that
    time of the year (periodic, Feb-29 dropped).  reference functions.py:195-303.
    Returns a labelled array (time, [plev,] lat, lon); time has length 1 when interpolated and is
    stamped with `era5_date_time` (:296)."""
    from . import ncio
    from .step_03_apply_to_era import delta_time_bracket
    ds = _open_cached(os.path.join(delta_input_dir, name_base.format(var_name)))
    fld = ds[var_name]
    times = np.asarray(ds[TIME_GCM].values)
    if fld.dims[0] != TIME_GCM:
        raise ValueError('first dimension of %s must be %s' % (var_name, TIME_GCM))
    ib, ia, x_hi, x_new, keep = delta_time_bracket(times, times[0] if target_date_time is None else target_date_time)
    if target_date_time is None:                                    # :298-301
        return ncio.Field(fld.values[keep], fld.dims, dict(fld.coords, **{TIME_GCM: times[keep]}), fld.attrs, var_name)
    vb = fld.values[keep[ib]]
    if x_hi == 0.0:                                                 # :282-283
        val = np.array(vb, copy=True)
    else:                                                           # :288-292 on the GPU
        val = time_lerp(vb, fld.values[keep[ia]], x_hi, x_new)
    t = np.asarray(getattr(era5_date_time, 'values', era5_date_time)).reshape(-1)[:1]
    coords = dict(fld.coords)
    coords[TIME_GCM] = t
    return ncio.Field(val[None], fld.dims, coords, fld.attrs, var_name)


def load_delta_interp(delta_input_dir, var_name, target_P, era5_date_time, target_date_time,
                      ignore_top_pressure_error=False):
    """load_delta + (for ta, hur) the surface delta and HIST surface pressure + vertical
    interpolation onto the model levels.  reference functions.py:306-340."""
    delta = load_delta(delta_input_dir, var_name, era5_date_time, target_date_time)
    if var_name in ['ta', 'hur']:
        delta_sfc = load_delta(delta_input_dir, var_name + 's', era5_date_time, target_date_time)
        ps_hist = load_delta(delta_input_dir, 'ps', era5_date_time, target_date_time,
                             name_base=file_name_bases['HIST'])
    else:
        delta_sfc = ps_hist = None
    return vert_interp_delta(delta, target_P, delta_sfc, ps_hist, ignore_top_pressure_error)


# ------------------------------------------------------------------------------- step_02
def regrid_lat_lon(ds_gcm, ds_era5, var_name, method='bilinear', i_use_xesmf=0):
    """Bilinear regridding of every lat/lon variable of `ds_gcm` onto the ERA5 grid of
    `ds_era5` (xarray branch of reference functions.py:748-898; the xESMF branch is not part of
    this build - SURVEY.md section 8c).  Returns a new Dataset on the target grid."""
    from . import ncio
    if i_use_xesmf:
        raise NotImplementedError('the xESMF regridding branch (functions.py:797-810) is out of scope; '
                                  'set i_use_xesmf_regridding = 0')
    targ_lon = np.asarray(ds_era5[LON_ERA].values, dtype=np.float64)
    targ_lat = np.asarray(ds_era5[LAT_ERA].values, dtype=np.float64)
    src_lon = np.asarray(ds_gcm[LON_GCM].values, dtype=np.float64)
    src_lat = np.asarray(ds_gcm[LAT_GCM].values, dtype=np.float64)
    out = ncio.Dataset(attrs=ds_gcm.attrs)
    for name, f in ds_gcm.variables.items():
        if name in (LAT_GCM, LON_GCM):
            continue
        if LAT_GCM in f.dims and LON_GCM in f.dims:
            lead = [d for d in f.dims if d not in (LAT_GCM, LON_GCM)]
            g = f.transpose(*(lead + [LAT_GCM, LON_GCM]))
            vals = g.values if g.values.dtype in (np.float32, np.float64) else g.values.astype(np.float64)
            res = regrid_field(vals, src_lat, src_lon, targ_lat, targ_lon)
            coords = {d: f.coords[d] for d in lead if d in f.coords}
            coords[LAT_GCM] = targ_lat
            coords[LON_GCM] = targ_lon
            out[name] = ncio.Field(res, tuple(lead) + (LAT_GCM, LON_GCM), coords, f.attrs, name)
        elif LAT_GCM not in f.dims and LON_GCM not in f.dims:
            out[name] = f
    out[LAT_GCM] = ncio.Field(targ_lat, (LAT_GCM,), {LAT_GCM: targ_lat}, ds_gcm[LAT_GCM].attrs, LAT_GCM)
    out[LON_GCM] = ncio.Field(targ_lon, (LON_GCM,), {LON_GCM: targ_lon}, ds_gcm[LON_GCM].attrs, LON_GCM)
    return out


# ------------------------------------------------------------------------------- step_02 smoothing
def harmonic_tables(lt):
    """cos / sin(2 pi i / lt * t), t = 1..lt, i = 1..3, evaluated as the reference does (functions.py:716, 727)
    so the table entries are the same doubles: ([3][lt], [3][lt])."""
    import math
    tv = np.arange(1, lt + 1, 1)
    arg = [2. * math.pi * i / lt * tv for i in (1, 2, 3)]
    return (np.ascontiguousarray(np.stack([np.cos(a) for a in arg])),
            np.ascontiguousarray(np.stack([np.sin(a) for a in arg])))


def smooth_annual_cycle(diff):
    """Spectral smoothing of every column of a (time, [level,] y, x) array on the GPU (`pgw_harmonic_smooth`):
    the array form of filter_data (reference functions.py:603-669).  Returns the kind of `diff` (host array, labelled
    array or DeviceArray), same dtype."""
    ctx = default_context()
    r = _raw(diff)
    if len(r.shape) not in (3, 4):
        raise ValueError('Wrong dimensions of input file should be 3 or 4-D')          # :648
    dtype = _common_dtype(diff)
    lt = int(r.shape[0])
    inner = int(np.prod(r.shape[1:], dtype=np.int64))
    cos_t, sin_t = harmonic_tables(max(lt, 1))
    d_in = _dev(ctx, diff, dtype)
    d_out = ctx.empty(r.shape, dtype)
    ctx._check(ctx.lib.pgw_harmonic_smooth(ctx.handle, dtype_tag(dtype), lt, inner, cos_t.ctypes.data_as(_lib._dp),
                                           sin_t.ctypes.data_as(_lib._dp), d_in.ptr, d_out.ptr))
    return _out(ctx, d_out, diff)


def harmonic_ac_analysis(ts):
    """Smoothed version of one series: mean + first three harmonics (reference functions.py:672-740); a series
    holding a NaN comes back all NaN in its own dtype, otherwise float64 like the reference.  Series shorter than 8
    steps: ValueError with the reference's text (the reference's `sys.exit` at :735 is a NameError, `sys` is not
    imported there)."""
    ts = np.asarray(ts)
    if ts.ndim != 1:
        raise ValueError('harmonic_ac_analysis expects a 1-D series')
    if np.isnan(ts).any():
        return np.full_like(ts, np.nan)
    out = smooth_annual_cycle(np.ascontiguousarray(ts, dtype=np.float64).reshape(-1, 1, 1))
    return out.reshape(-1)


def filter_data(annualcycleraw, variablename_to_smooth, outputpath):
    """File form (reference functions.py:603-669): read the variable, drop size-1 dimensions (`.squeeze()`), smooth
    every column along the first dimension, write the variable with its coordinates to `outputpath`."""
    from . import ncio
    ds = ncio.open_dataset(annualcycleraw)
    f = ds[variablename_to_smooth]
    keep = [i for i, n in enumerate(f.shape) if n != 1]
    vals = f.values.reshape([f.shape[i] for i in keep])
    dims = tuple(f.dims[i] for i in keep)
    print('Dimension that is assumed to be time dimension is called: ', dims[0] if dims else None)
    print('shape of data: ', vals.shape)
    if vals.dtype not in (np.float32, np.float64):
        vals = vals.astype(np.float64)
    res = smooth_annual_cycle(vals)
    print('Done with smoothing')
    out = ncio.Dataset(attrs={})
    for d in dims:
        if d in ds:
            out[d] = ds[d]
    out[variablename_to_smooth] = ncio.Field(res, dims, {d: f.coords[d] for d in dims if d in f.coords}, f.attrs)
    ncio.to_netcdf(out, outputpath)


# ------------------------------------------------------------------------------- step_02: ocean-grid deltas
def _fold_lon(lon):
    """functions.py:938-941 / 998-1001: longitudes above 180 move to the (-180, 180] range."""
    lon = np.array(lon, dtype=np.float64, copy=True)
    lon[lon > 180] -= 360
    return lon


def planar_metres(lat_deg, lon_deg):
    """The reference's point-cloud coordinates (functions.py:958-975, 1010-1023: three pyproj Geod.inv lengths per point) on
    the GPU (`pgw_planar_metres`): (lat_m, lon_m, lon_offset) in metres for latitudes / longitudes in degrees, longitudes
    already folded to (-180, 180].  The arithmetic is pgw4era5_amd/geodesy.py's (the host form, kept as the check)."""
    ctx = default_context()
    lat = np.ascontiguousarray(lat_deg, dtype=np.float64).reshape(-1)
    lon = np.ascontiguousarray(lon_deg, dtype=np.float64).reshape(-1)
    if lat.shape != lon.shape:
        raise ValueError('latitudes and longitudes must have the same number of points')
    n = len(lat)
    if n == 0:
        return np.zeros(0), np.zeros(0), np.zeros(0)
    f64 = np.dtype('float64')
    d_lat, d_lon = ctx.to_device(lat, f64), ctx.to_device(lon, f64)
    outs = [ctx.empty((n,), f64) for _ in range(3)]
    ctx._check(ctx.lib.pgw_planar_metres(ctx.handle, n, d_lat.ptr, d_lon.ptr, outs[0].ptr, outs[1].ptr, outs[2].ptr))
    return tuple(o.numpy() for o in outs)


def gauss_interp_fields(land_fr, era5_lat, era5_lon, gcm_lat, gcm_lon, fields, kernel_radius, sharpness):
    """The geometry and the GPU pass of nan_ignoring_interp for SEVERAL fields on the same source points (the twelve
    months of a variable: interp_wrapper calls nan_ignoring_interp once per month, functions.py:1102-1109, rebuilding
    the same two point clouds each time).
    land_fr (nlat, nlon); era5_lat (nlat), era5_lon (nlon); gcm_lat, gcm_lon, fields[k]: arrays of one common shape
    (curvilinear 2-D coordinates flattened like :931-933).  Returns (nfield, nlat, nlon) float64."""
    from . import geodesy
    ctx = default_context()
    vals = np.stack([np.asarray(f, dtype=np.float64).reshape(-1) for f in fields], axis=1)       # (npoint, nfield)
    glat = np.asarray(gcm_lat, dtype=np.float64).reshape(-1)
    glon = _fold_lon(np.asarray(gcm_lon).reshape(-1))
    if not (glat.shape == glon.shape == vals.shape[:1]):
        raise ValueError('ocean-grid coordinates and values must have the same number of points')
    nf = vals.shape[1]
    keep = ~np.isnan(vals).all(axis=1)                         # :944-948 (a point that is NaN in every field is in no cloud)
    glat, glon, vals = glat[keep], glon[keep], vals[keep]
    lat_m, lon_m, lon_off = planar_metres(glat, glon)                                           # :958-975
    # :977-991 the whole field once more to the left and to the right, shifted by twice the half-way-round length
    sx = np.tile(lat_m, 3)
    sy = np.concatenate([lon_m - 2 * lon_off, lon_m, lon_m + 2 * lon_off])
    sv = np.tile(vals, (3, 1))
    elat = np.asarray(era5_lat, dtype=np.float64)
    elon = _fold_lon(era5_lon)
    tlat = np.repeat(elat, len(elon)); tlon = np.tile(elon, len(elat))                         # :1004-1005
    # lat_m depends on the latitude only, lon_m on (|lat|, |lon|): evaluate the distinct values of the regular grid once
    la_u, la_i = np.unique(np.abs(elat), return_inverse=True)
    lo_u, lo_i = np.unique(np.abs(elon), return_inverse=True)
    uu_lat, uu_lon = np.repeat(la_u, len(lo_u)), np.tile(lo_u, len(la_u))
    u_arc, u_lon, _ = planar_metres(uu_lat, uu_lon)
    lat_arc = u_arc.reshape(len(la_u), len(lo_u))[:, 0]
    lon_arc = u_lon.reshape(len(la_u), len(lo_u))
    tx = (lat_arc[la_i] * np.sign(elat))[:, None] * np.ones(len(elon))[None, :]
    ty = lon_arc[np.ix_(la_i, lo_i)] * np.sign(elon)[None, :]
    # the kernel stages the source cells a BLOCK of 256 consecutive targets needs, and a wave runs the weight arithmetic of a
    # source point when ANY of its 64 lanes has the point within the radius: hand the targets over in tiles of 16 x 16 grid
    # points (one block; a wave = 4 x 16 of them) instead of row by row - a compact footprint, so few cells per block and
    # few accepted points per wave that most of its lanes reject.  Edge tiles are filled with NaN (inactive) targets.
    nlat_t, nlon_t = tx.shape
    TILE = 16
    nlat_p, nlon_p = -(-nlat_t // TILE) * TILE, -(-nlon_t // TILE) * TILE

    # tiles of the high latitudes first: the planar cloud is densest there (a parallel shrinks, the points on it do not get
    # fewer), so those blocks run longest; started first they do not form the tail of the launch
    nty, ntx = nlat_p // TILE, nlon_p // TILE
    row_lat = np.abs(np.pad(elat, (0, nlat_p - nlat_t), mode='edge').reshape(nty, TILE)).mean(axis=1)
    tile_order = np.argsort(-np.repeat(row_lat, ntx), kind='stable')

    def tiles(a):
        full = np.full((nlat_p, nlon_p), np.nan)
        full[:nlat_t, :nlon_t] = a
        return full.reshape(nty, TILE, ntx, TILE).transpose(0, 2, 1, 3).reshape(nty * ntx, TILE * TILE)[tile_order]
    tx, ty = tiles(tx), tiles(ty)
    tx, ty = np.ascontiguousarray(tx.reshape(-1)), np.ascontiguousarray(ty.reshape(-1))
    # uniform cells of one kernel radius over the source cloud
    h = float(kernel_radius)
    if len(sx):
        x0, y0 = float(sx.min()), float(sy.min())
        ncx, ncy = int((sx.max() - x0) // h) + 1, int((sy.max() - y0) // h) + 1
        cid = ((sx - x0) // h).astype(np.int64) * ncy + ((sy - y0) // h).astype(np.int64)
        order = np.argsort(cid, kind='stable')
        sx, sy, sv, cid = sx[order], sy[order], sv[order], cid[order]
        cell_start = np.searchsorted(cid, np.arange(ncx * ncy + 1)).astype(np.int32)
    else:
        x0 = y0 = 0.0; ncx = ncy = 1
        cell_start = np.zeros(2, dtype=np.int32)
    f64 = np.dtype('float64')
    d_tx, d_ty = ctx.to_device(tx, f64), ctx.to_device(ty, f64)
    d_sx, d_sy = ctx.to_device(sx if len(sx) else np.zeros(1), f64), ctx.to_device(sy if len(sy) else np.zeros(1), f64)
    d_sv = ctx.to_device(np.ascontiguousarray(sv) if len(sx) else np.zeros((1, nf)), f64)
    d_cs = ctx.empty(cell_start.shape, np.int32).copy_from(cell_start)
    ntarg = len(tx)
    out = np.empty((nf, ntarg))
    for k0 in range(0, nf, 16):                               # the kernel takes up to 16 fields per pass
        k1 = min(k0 + 16, nf)
        if k0 or k1 < nf:
            d_sub = ctx.to_device(np.ascontiguousarray(sv[:, k0:k1]), f64)
        else:
            d_sub = d_sv
        d_out = ctx.empty((k1 - k0, ntarg), f64)
        ctx._check(ctx.lib.pgw_gauss_interp(ctx.handle, ntarg, d_tx.ptr, d_ty.ptr, ncx, ncy, x0, y0, h, d_cs.ptr, len(sx),
                                            d_sx.ptr, d_sy.ptr, d_sub.ptr, k1 - k0, float(kernel_radius), float(sharpness), d_out.ptr))
        out[k0:k1] = d_out.numpy()
    back = np.empty_like(tile_order)
    back[tile_order] = np.arange(len(tile_order))
    out = out.reshape(nf, nty * ntx, TILE * TILE)[:, back]
    out = out.reshape(nf, nty, ntx, TILE, TILE).transpose(0, 1, 3, 2, 4).reshape(nf, nlat_p, nlon_p)
    out = np.ascontiguousarray(out[:, :nlat_t, :nlon_t]).reshape(nf, -1)
    land = np.asarray(land_fr, dtype=np.float64).reshape(-1)
    out[:, land > 0.7] = np.nan                               # :1032, 1055: no SST on land points
    return out.reshape(nf, len(elat), len(elon))


def _ocean_coords(da_delta):
    """Latitudes / longitudes of the ocean grid as arrays of the values' shape (functions.py:920-933).  The reference
    builds a meshgrid for 1-D coordinates and then overwrites it with the raw 1-D coordinates (:931-932), which cannot
    index the flattened values; the meshgrid (evidently intended) is used here."""
    lat = np.asarray(da_delta.coords[LAT_GCM_OCEAN])
    lon = np.asarray(da_delta.coords[LON_GCM_OCEAN])
    if lat.ndim == 2:
        return lat, lon
    if lat.ndim == 1:
        return np.meshgrid(lat, lon, indexing='ij')
    raise NotImplementedError()


def nan_ignoring_interp(da_era5_land_fr, da_delta, kernel_radius, sharpness):
    """Point-cloud interpolation of a 2-D ocean-grid field onto the ERA5 grid, ignoring NaN source points; land points
    (FR_LAND > 0.7) come back NaN.  reference functions.py:900-1060.  Labelled inputs like the reference's
    (`.values`, `.coords` with the ocean grid's `latitude` / `longitude`, ERA5 `lat` / `lon`)."""
    glat, glon = _ocean_coords(da_delta)
    res = gauss_interp_fields(np.asarray(da_era5_land_fr.values), da_era5_land_fr.coords[LAT_ERA], da_era5_land_fr.coords[LON_ERA],
                              glat, glon, [np.asarray(da_delta.values)], kernel_radius, sharpness)
    return res[0]


def interp_wrapper(origin_grid, target_grid, var_name, i_use_xesmf=0,
                   nan_interp_kernel_radius=300000, nan_interp_sharpness=3):
    """Per-variable choice of the regridding scheme (reference functions.py:1062-1141).
    Atmospheric variables: bilinear on the GPU.  `tos` / `siconc` (ocean grid, NaN over land): the Gaussian-kernel
    point-cloud interpolation, all twelve months in one pass."""
    from . import ncio
    if var_name in ['tos', 'siconc']:
        land = target_grid['FR_LAND']
        land2d = np.asarray(land.values)[0]                                        # target_grid["FR_LAND"][0,:,:]  :1097
        values = origin_grid[var_name]
        glat, glon = _ocean_coords(ncio.Field(values.values[0], values.dims[1:],
                                              {LAT_GCM_OCEAN: np.asarray(origin_grid[LAT_GCM_OCEAN].values),
                                               LON_GCM_OCEAN: np.asarray(origin_grid[LON_GCM_OCEAN].values)}))
        if values.shape[0] != 12:
            raise ValueError('could not broadcast input array: %s has %d time steps, the ocean-grid interpolation expects 12 months'
                             % (var_name, values.shape[0]))                         # result = np.empty((12, ...))  :1101
        tlat = np.asarray(target_grid[LAT_ERA].values, dtype=np.float64)
        tlon = np.asarray(target_grid[LON_ERA].values, dtype=np.float64)
        result = gauss_interp_fields(land2d, tlat, tlon, glat, glon, [values.values[i] for i in range(12)],
                                     nan_interp_kernel_radius, nan_interp_sharpness)
        ds = ncio.Dataset(attrs=dict(description=str(var_name) + " on ERA5 grid", units="K", long_name=str(var_name)))   # :1121, 1134
        ds['lat'] = ncio.Field(tlat, ('lat',), {'lat': tlat}, target_grid[LAT_ERA].attrs)
        ds['lon'] = ncio.Field(tlon, ('lon',), {'lon': tlon}, target_grid[LON_ERA].attrs)
        t = origin_grid[TIME_GCM]
        ds['time'] = ncio.Field(t.values, ('time',), {'time': t.values}, {k: v for k, v in t.attrs.items() if k not in ('units', 'calendar')}
                                if t.values.dtype.kind == 'M' else t.attrs)
        ds[var_name] = ncio.Field(result, ('time', 'lat', 'lon'), {'time': t.values, 'lat': tlat, 'lon': tlon})
        return ds
    return regrid_lat_lon(origin_grid, target_grid, var_name, method='bilinear', i_use_xesmf=i_use_xesmf)
