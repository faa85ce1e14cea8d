"""
CPU oracle for the PGW4ERA5 step_03 / step_02 hot path  --  TEST INFRASTRUCTURE ONLY.

This file is a plain numpy (fp64) restatement of the reference algorithm.  It is the
checker the HIP path is compared against; it is NOT the product.  Only `tests/`,
`__graft_entry__.smoke()` and the `cpu_baseline` leg of `bench.py` may import it.
The product path (`pgw4era5_amd/`) never imports anything from `oracle/`.

Parity pin status: the reference ships no golden vectors.  The leaf
arithmetic that is pure numpy in the reference (`interp_extrap_1d`,
`interp_1d_for_timelatlon`, `replace_delta_sfc`, `determine_p_ref`, vapor-pressure
helpers, `integrate_tos`) is pinned by fixtures in `tests/golden/` that were produced by
importing the reference itself (`oracle/make_golden.py`, run in the build container).
The xarray-bound functions (`integ_geopot`, the humidity wrappers, `vert_interp_delta`,
`load_delta`, `regrid_lat_lon`, the step_03 loop) cannot be executed here (xarray is not
installed); they are restated index-wise from the cited lines and pinned by (i) the leaf
fixtures, (ii) analytic cases and (iii) `scipy.interpolate.interp1d` (the arithmetic
under xarray's `.interp`).  Everything that only has (ii)/(iii) is "parity unpinned
against a reference run" and DESIGN.md says so.

All `file:line` citations are relative to /root/reference.

Array convention: C-order `(time, level|level1|plev, lat, lon)`; index increases with
pressure (downward) along the level axis.
"""
import numpy as np

# constants.py:3-7
CON_RD = 287.05
CON_G = 9.80665
CON_MW_MD = 0.622

# settings.py:140-150
P_REF_INP = 30000.0
ADJ_FACTOR = 0.95
THRESH_PHI_REF_MAX_ERROR = 0.15
MAX_N_ITER = 20


# ----------------------------------------------------------------------------------------
# a1  hybrid-level pressure ("integ_pressure")            step_03_apply_to_era.py:64-88,196-199
# ----------------------------------------------------------------------------------------
def full_level_coeffs(ak, bk):
    """akm/bkm from half-level coefficients, step_03:74-85.

    `0.5 * ak.diff(label='lower') + ak[:-1]`  (this exact operation order).
    """
    ak = np.asarray(ak, dtype=np.float64)
    bk = np.asarray(bk, dtype=np.float64)
    akm = 0.5 * (ak[1:] - ak[:-1]) + ak[:-1]
    bkm = 0.5 * (bk[1:] - bk[:-1]) + bk[:-1]
    return akm, bkm


def hybrid_pressure(ak, bk, ps, akm=None, bkm=None):
    """pa_hl = ak + ps*bk (step_03:64-66), pa = akm + ps*bkm (step_03:87-88).

    ps: (time, lat, lon).  Returns pa_hl (time, N+1, lat, lon), pa (time, N, lat, lon).
    """
    ps = np.asarray(ps, dtype=np.float64)
    if akm is None or bkm is None:
        akm, bkm = full_level_coeffs(ak, bk)
    ak = np.asarray(ak, dtype=np.float64)[None, :, None, None]
    bk = np.asarray(bk, dtype=np.float64)[None, :, None, None]
    akm = np.asarray(akm, dtype=np.float64)[None, :, None, None]
    bkm = np.asarray(bkm, dtype=np.float64)[None, :, None, None]
    pa_hl = ak + ps[:, None] * bk
    pa = akm + ps[:, None] * bkm
    return pa_hl, pa


# ----------------------------------------------------------------------------------------
# a2/a3  humidity thermodynamics                                    functions.py:58-125
# ----------------------------------------------------------------------------------------
def specific_humidity_to_vapor_pressure(hus, pa):
    """functions.py:58-64"""
    return hus * pa / (CON_MW_MD + 0.378 * hus)


def vapor_pressure_to_specific_humidity(vapp, pa):
    """functions.py:66-72"""
    return CON_MW_MD * vapp / (pa - (1 - CON_MW_MD) * vapp)


def saturation_vapor_pressure_water_or_ice(pa, ta, water=True):
    """IFS 7.93, functions.py:74-89"""
    T0 = 273.16
    if water:
        a1, a3, a4 = 611.21, 17.502, 32.19
    else:
        a1, a3, a4 = 611.21, 22.587, -0.7
    return a1 * np.exp(a3 * (ta - T0) / (ta - a4))


def saturation_vapor_pressure_water_and_ice(pa, ta):
    """IFS 7.92, functions.py:91-105 (xr.where chain restated with np.where)."""
    T0 = 273.16
    Ti = 250.16
    ta = np.asarray(ta, dtype=np.float64)
    alpha = np.full_like(ta, np.nan)
    alpha = np.where(ta >= T0, 1.0, alpha)
    alpha = np.where(ta <= Ti, 0.0, alpha)
    with np.errstate(invalid='ignore'):
        mixed = np.power((ta - Ti) / (T0 - Ti), 2.)
    alpha = np.where((ta < T0) & (ta > Ti), mixed, alpha)
    return (alpha * saturation_vapor_pressure_water_or_ice(pa, ta, water=True) +
            (1 - alpha) * saturation_vapor_pressure_water_or_ice(pa, ta, water=False))


def specific_to_relative_humidity(hus, pa, ta):
    """functions.py:107-116"""
    return (specific_humidity_to_vapor_pressure(hus, pa) /
            saturation_vapor_pressure_water_and_ice(pa, ta)) * 100


def relative_to_specific_humidity(hur, pa, ta):
    """functions.py:118-125"""
    vapp = hur / 100 * saturation_vapor_pressure_water_and_ice(pa, ta)
    return vapor_pressure_to_specific_humidity(vapp, pa)


# ----------------------------------------------------------------------------------------
# a4  integ_geopot                                                  functions.py:128-189
# ----------------------------------------------------------------------------------------
def integ_geopot(pa_hl, zgs, ta, hus, level1, p_ref):
    """Hydrostatic integration surface -> p_ref.

    pa_hl (time,N+1,lat,lon); zgs (time,lat,lon); ta,hus (time,N,lat,lon);
    level1 = half-level labels (only its length is used; labels are 1..N+1);
    p_ref scalar or (time,lat,lon).  Level-wise vectorised like functions.py:147-152.
    """
    pa_hl = np.asarray(pa_hl, dtype=np.float64)
    ta = np.asarray(ta, dtype=np.float64)
    hus = np.asarray(hus, dtype=np.float64)
    zgs = np.asarray(zgs, dtype=np.float64)
    nt, nhl, nlat, nlon = pa_hl.shape
    n = nhl - 1
    if len(level1) != nhl or ta.shape[1] != n:
        raise ValueError('level dimensions are inconsistent')
    # :135  NaN > 0 is False -> NaN becomes 1e-4 as well
    with np.errstate(invalid='ignore'):
        p = np.where(pa_hl > 0, pa_hl, 0.0001)
    lnp = np.log(p)
    dlnpa = lnp[:, 1:] - lnp[:, :-1]                      # :136-138
    tav = ta * (1 + 0.61 * hus)                           # :144
    phi_hl = np.empty_like(p)
    phi_hl[:, n] = zgs                                    # :141
    for l in range(n - 1, -1, -1):                        # :147-152
        phi_hl[:, l] = phi_hl[:, l + 1] + (CON_RD * tav[:, l] * dlnpa[:, l])
    p_ref_arr = np.asarray(p_ref, dtype=np.float64)
    if p_ref_arr.ndim == 3:
        p_ref_b = p_ref_arr[:, None]
    else:
        p_ref_b = p_ref_arr
    p_diff = p - p_ref_b                                  # :160
    with np.errstate(invalid='ignore'):
        p_diff = np.where(p_diff >= 0, p_diff, np.nan)    # :161
    if np.any(np.all(np.isnan(p_diff), axis=1)):          # :162-165
        raise ValueError("p_ref locally lies below the surface. Please set a lower "
                         "reference pressue (p_ref_inp) in settings.py")
    ind = np.nanargmin(p_diff, axis=1)                    # (time,lat,lon), ties -> lowest k
    if np.any(ind == 0):
        # tav.sel(level=0) raises KeyError in the reference (labels start at 1)
        raise KeyError(0)
    ind4 = ind[:, None]
    p_ref_star = np.take_along_axis(p, ind4, axis=1)[:, 0]          # :169
    phi_ref_star = np.take_along_axis(phi_hl, ind4, axis=1)[:, 0]   # :170
    tav_star = np.take_along_axis(tav, ind4 - 1, axis=1)[:, 0]      # :176
    phi_ref = phi_ref_star - (CON_RD * tav_star) * (np.log(p_ref_arr) - np.log(p_ref_star))
    return phi_ref


# ----------------------------------------------------------------------------------------
# a6  interp_logp_4d and its column kernels                         functions.py:434-580
# ----------------------------------------------------------------------------------------
_MODES = ('off', 'linear', 'constant', 'nan')


def interp_extrap_1d(src_x, src_y, targ_x, extrapolate):
    """functions.py:511-580 restated.  Pure-Python loop: small cases."""
    ns = len(src_x)
    targ_y = np.zeros(len(targ_x))
    for ti in range(len(targ_x)):
        x = targ_x[ti]
        i1 = i2 = -1
        require_extrap = False
        for si in range(ns):
            if si == 0 and src_x[si] > x:                 # :530-538
                if extrapolate == 'linear':
                    i1, i2 = 0, 1
                elif extrapolate == 'constant':
                    i1, i2 = 0, 0
                require_extrap = True
                break
            elif src_x[si] == x:                          # :540-543
                i1 = i2 = si
                break
            elif src_x[si] > x:                           # :545-548
                i1, i2 = si - 1, si
                break
        if i1 == -1:                                      # :554-561
            if extrapolate == 'linear':
                i1, i2 = ns - 2, ns - 1
            elif extrapolate == 'constant':
                i1 = i2 = ns - 1
            require_extrap = True
        if require_extrap and extrapolate == 'off':       # :564-566
            raise ValueError('Extrapolation deactivated but data out of bounds.')
        if require_extrap and extrapolate == 'nan':       # :569-570
            targ_y[ti] = np.nan
        elif i1 == i2:
            targ_y[ti] = src_y[i1]
        else:                                             # :575-578
            targ_y[ti] = (src_y[i1] + (x - src_x[i1]) *
                          (src_y[i2] - src_y[i1]) / (src_x[i2] - src_x[i1]))
    return targ_y


def interp_1d_for_timelatlon(orig_array, src_p, targ_p, interp_array,
                             ntime, nlat, nlon, extrapolate):
    """functions.py:479-508 (inputs are already ln p; writes interp_array in place)."""
    for t in range(ntime):
        for j in range(nlat):
            for i in range(nlon):
                sp = src_p[t, :, j, i]
                tp = targ_p[t, :, j, i]
                if sp[-1] < sp[0]:
                    raise ValueError('Source pressure values must be ascending!')
                if tp[-1] < tp[0]:
                    raise ValueError('Target pressure values must be ascending!')
                interp_array[t, :, j, i] = interp_extrap_1d(sp, orig_array[t, :, j, i],
                                                            tp, extrapolate)


def interp_columns_vectorised(src_x, src_y, targ_x, extrapolate):
    """Vectorised equivalent of interp_extrap_1d over many columns (fast oracle).

    src_x, src_y: (S, ncol); targ_x: (N, ncol).  Same selection rule as A3: the first
    source index s with NOT(src_x[s] < x) ... realised as the first s with
    (src_x[s] == x) or (src_x[s] > x); exact hit -> y[s]; s == 0 and '>' -> below range;
    none -> above range.  Returns (out (N,ncol), extrap_flag (N,ncol)).
    """
    S, ncol = src_x.shape
    N = targ_x.shape[0]
    out = np.empty((N, ncol))
    flag = np.zeros((N, ncol), dtype=bool)
    cols = np.arange(ncol)
    for ti in range(N):
        x = targ_x[ti]
        with np.errstate(invalid='ignore'):
            ge = (src_x == x) | (src_x > x)                # (S, ncol)
        any_hit = ge.any(axis=0)
        s = np.where(any_hit, ge.argmax(axis=0), S)        # first hit or S
        sc = np.minimum(s, S - 1)
        with np.errstate(invalid='ignore'):
            exact = any_hit & (src_x[sc, cols] == x)
        below = any_hit & ~exact & (s == 0)
        above = ~any_hit
        i1 = np.where(exact, sc, sc - 1)
        i2 = sc.copy()
        if extrapolate == 'linear':
            i1 = np.where(below, 0, i1); i2 = np.where(below, 1, i2)
            i1 = np.where(above, S - 2, i1); i2 = np.where(above, S - 1, i2)
        else:
            i1 = np.where(below, 0, i1); i2 = np.where(below, 0, i2)
            i1 = np.where(above, S - 1, i1); i2 = np.where(above, S - 1, i2)
        i1 = np.clip(i1, 0, S - 1)
        y1 = src_y[i1, cols]; y2 = src_y[i2, cols]
        x1 = src_x[i1, cols]; x2 = src_x[i2, cols]
        with np.errstate(invalid='ignore', divide='ignore'):
            lin = y1 + (x - x1) * (y2 - y1) / (x2 - x1)
        val = np.where(i1 == i2, y1, lin)
        ext = below | above
        if extrapolate == 'nan':
            val = np.where(ext, np.nan, val)
        out[ti] = val
        flag[ti] = ext
    return out, flag


def interp_logp_4d(var, source_P, targ_P, extrapolate='off', fast=True):
    """functions.py:434-477 on plain arrays (time,lev,lat,lon)."""
    if extrapolate not in _MODES:
        raise ValueError('Invalid input value for "extrapolate"')
    var = np.asarray(var, dtype=np.float64)
    source_P = np.asarray(source_P, dtype=np.float64)
    targ_P = np.asarray(targ_P, dtype=np.float64)
    if (var.shape[0] != source_P.shape[0]) or (var.shape[0] != targ_P.shape[0]):
        raise ValueError('Time dimension of input files is inconsistent!')
    if (var.shape[2] != source_P.shape[2]) or (var.shape[2] != targ_P.shape[2]):
        raise ValueError('Lat dimension of input files is inconsistent!')
    if (var.shape[3] != source_P.shape[3]) or (var.shape[3] != targ_P.shape[3]):
        raise ValueError('Lon dimension of input files is inconsistent!')
    nt, N, nlat, nlon = targ_P.shape
    with np.errstate(invalid='ignore', divide='ignore'):
        lsp = np.log(source_P)
        ltp = np.log(targ_P)
    tmp = np.zeros_like(targ_P)
    if not fast:
        interp_1d_for_timelatlon(var, lsp, ltp, tmp, nt, nlat, nlon, extrapolate)
        return tmp
    S = var.shape[1]
    for t in range(nt):
        sx = lsp[t].reshape(S, -1); sy = var[t].reshape(S, -1); tx = ltp[t].reshape(N, -1)
        if np.any(sx[-1] < sx[0]):
            raise ValueError('Source pressure values must be ascending!')
        if np.any(tx[-1] < tx[0]):
            raise ValueError('Target pressure values must be ascending!')
        out, flag = interp_columns_vectorised(sx, sy, tx, extrapolate)
        if extrapolate == 'off' and flag.any():
            raise ValueError('Extrapolation deactivated but data out of bounds.')
        tmp[t] = out.reshape(N, nlat, nlon)
    return tmp


# ----------------------------------------------------------------------------------------
# a8  replace_delta_sfc / vert_interp_delta                         functions.py:343-431
# ----------------------------------------------------------------------------------------
def replace_delta_sfc(source_P, ps_hist, delta, delta_sfc):
    """functions.py:343-366, one ascending-pressure column."""
    out_source_P = np.array(source_P, dtype=np.float64, copy=True)
    out_delta = np.array(delta, dtype=np.float64, copy=True)
    if ps_hist > np.max(source_P):
        sfc_ind = len(source_P) - 1
        out_source_P[sfc_ind] = ps_hist
        out_delta[sfc_ind] = delta_sfc
    elif ps_hist < np.min(source_P):
        raise ValueError()
    else:
        sfc_ind = np.max(np.argwhere(ps_hist > source_P))   # empty -> ValueError
        out_delta[sfc_ind:] = delta_sfc
        out_source_P[sfc_ind] = ps_hist
    return out_source_P, out_delta


def replace_delta_sfc_columns(plev_asc, ps_hist, delta, delta_sfc):
    """Vectorised replace_delta_sfc over columns.

    plev_asc (S,) ascending; delta (S,ncol); ps_hist, delta_sfc (ncol,).
    Returns source_P (S,ncol), delta (S,ncol).  Raises ValueError like the reference when
    ps_hist <= min(plev) anywhere (functions.py:360-361 and the empty-argwhere case).
    NaN ps_hist: all comparisons False -> reference takes the else branch and np.max of an
    empty argwhere raises ValueError as well.
    """
    S, ncol = delta.shape
    P = np.repeat(np.asarray(plev_asc, dtype=np.float64)[:, None], ncol, axis=1)
    D = np.array(delta, dtype=np.float64, copy=True)
    with np.errstate(invalid='ignore'):
        gt = ps_hist[None, :] > P                          # (S,ncol)
    if not np.all(gt.any(axis=0)):
        raise ValueError()
    # last index where ps_hist > P
    k = S - 1 - np.argmax(gt[::-1], axis=0)
    lev = np.arange(S)[:, None]
    D = np.where(lev >= k[None, :], delta_sfc[None, :], D)
    P[k, np.arange(ncol)] = ps_hist
    return P, D


def vert_interp_delta(delta, plev, target_P, delta_sfc=None, ps_hist=None,
                      ignore_top_pressure_error=False):
    """functions.py:369-431 on plain arrays.

    delta (time,S,lat,lon) on `plev` (S,) in the file's (descending, CMIP) order - it is
    reversed here exactly like :383-384 (a plain reversal, whatever the input order).
    delta_sfc, ps_hist: (time,lat,lon) or None.  target_P (time,N,lat,lon).
    """
    delta = np.asarray(delta, dtype=np.float64)[:, ::-1]
    plev_r = np.asarray(plev, dtype=np.float64)[::-1]
    target_P = np.asarray(target_P, dtype=np.float64)
    nt, S, nlat, nlon = delta.shape
    source_P = np.broadcast_to(plev_r[None, :, None, None], delta.shape).copy()
    if delta_sfc is not None:
        delta = delta.copy()
        for t in range(nt):
            P, D = replace_delta_sfc_columns(plev_r, np.asarray(ps_hist[t], dtype=np.float64).reshape(-1),
                                             delta[t].reshape(S, -1),
                                             np.asarray(delta_sfc[t], dtype=np.float64).reshape(-1))
            source_P[t] = P.reshape(S, nlat, nlon)
            delta[t] = D.reshape(S, nlat, nlon)
    if np.min(target_P) < np.min(source_P):                # :417-425
        if not ignore_top_pressure_error:
            raise ValueError('ERA5 top pressure is lower than climate delta top pressure.')
    return interp_logp_4d(delta, source_P, target_P, extrapolate='constant')


# ----------------------------------------------------------------------------------------
# a7  load_delta time interpolation                                 functions.py:195-303
# ----------------------------------------------------------------------------------------
def delta_time_bracket(delta_times, target):
    """functions.py:224-283 on numpy datetime64 values.

    delta_times: 1-D datetime64 array (any year, file order); target: datetime64.
    Returns (ind_before, ind_after, t_before, t_after, keep) where `keep` is the index
    array of the records that survive the Feb-29 drop (:224-230: only the LAST Feb-29
    found is dropped, like the reference) and indices refer to the kept records.
    """
    delta_times = np.asarray(delta_times).astype('datetime64[s]')
    target = np.datetime64(target).astype('datetime64[s]')
    md = [(int(str(t)[5:7]), int(str(t)[8:10])) for t in delta_times]
    leap = None
    for i, (m, d) in enumerate(md):
        if m == 2 and d == 29:
            leap = i
    keep = np.array([i for i in range(len(delta_times)) if i != leap], dtype=np.int64)
    times = delta_times[keep]
    year = int(str(target)[:4])

    def with_year(t, y):
        s = str(t)
        return np.datetime64('%04d' % y + s[4:]).astype('datetime64[s]')

    times_y = np.array([with_year(t, year) for t in times])   # :235-238
    is_before = times_y <= target                             # :242
    if is_before.sum() > 0:
        ib = int(np.argwhere(is_before)[-1].squeeze())
        tb = times_y[ib]
    else:                                                     # :253-258
        ib = len(times_y) - 1
        tb = with_year(times_y[ib], year - 1)
    is_after = times_y >= target                              # :262
    if is_after.sum() > 0:
        ia = int(np.argwhere(is_after)[0].squeeze())
        ta = times_y[ia]
    else:                                                     # :273-278
        ia = 0
        ta = with_year(times_y[ia], year + 1)
    # the reference compares -1 with the after index (:254,282); -1 only equals it
    # when there is a single record, which the comparison below also covers
    return ib, ia, tb, ta, keep


def time_lerp(v_before, v_after, t_before, t_after, target):
    """xarray .interp(time=...) == scipy interp1d linear on float ns offsets (:288-292).

    slope = (y_hi - y_lo) / (x_hi - x_lo);  y = slope * (x_new - x_lo) + y_lo
    with x floatised as nanoseconds relative to the smaller coordinate (xarray _floatize_x).
    """
    x_hi = float((np.datetime64(t_after).astype('datetime64[ns]') -
                  np.datetime64(t_before).astype('datetime64[ns]')).astype(np.int64))
    x_new = float((np.datetime64(target).astype('datetime64[ns]') -
                   np.datetime64(t_before).astype('datetime64[ns]')).astype(np.int64))
    slope = (np.asarray(v_after, dtype=np.float64) - np.asarray(v_before, dtype=np.float64)) / (x_hi - 0.0)
    return slope * (x_new - 0.0) + np.asarray(v_before, dtype=np.float64)


def load_delta_values(values, delta_times, target):
    """load_delta (functions.py:195-303) on an in-memory record array values[time,...].

    Returns array with a leading time axis of length 1.  target=None -> all kept records.
    """
    values = np.asarray(values, dtype=np.float64)
    if target is None:
        _, _, _, _, keep = delta_time_bracket(delta_times, np.asarray(delta_times)[0])
        return values[keep]
    ib, ia, tb, ta, keep = delta_time_bracket(delta_times, target)
    v = values[keep]
    if ib == ia:
        return v[ib][None]
    return time_lerp(v[ib], v[ia], tb, ta, target)[None]


# ----------------------------------------------------------------------------------------
# determine_p_ref                                                   functions.py:583-598
# ----------------------------------------------------------------------------------------
def determine_p_ref(p_min_era, p_min_pgw, p_ref_opts, p_ref_last=None):
    """functions.py:583-598 (scalar version; returns None if no candidate)."""
    for p in p_ref_opts:
        if (p_min_era > p) & (p_min_pgw > p):
            if p_ref_last is None:
                return p
            return min(p, p_ref_last)
    return None


# ----------------------------------------------------------------------------------------
# a9  surface riders                      step_03:103-146, functions.py:1145-1186
# ----------------------------------------------------------------------------------------
def integrate_tos(tos_field, ts_field, land_frac, ice_frac):
    """functions.py:1145-1186"""
    dims = tos_field.shape
    ice = np.asarray(ice_frac, dtype=np.float64).reshape(-1)
    tos = np.asarray(tos_field, dtype=np.float64).reshape(-1)
    ts = np.asarray(ts_field, dtype=np.float64).reshape(-1)
    land = np.asarray(land_frac, dtype=np.float64).reshape(-1)
    mask = ~np.isnan(ice) & ~np.isnan(tos)
    out = ts.copy()
    frac = np.clip(ice[mask] + land[mask], 0, 1)
    out[mask] = frac * ts[mask] + (1 - frac) * tos[mask]
    return out.reshape(dims)


def sea_ice_update(sic, delta_siconc):
    """step_03:105-107"""
    return np.clip(sic + delta_siconc / 100, 0, 1)


def soil_temperature_delta(delta_ts, delta_st_clim, soil_depth):
    """step_03:139-142: clim + exp(-z/2.8) * (delta_ts - clim); result (time,soil,lat,lon)."""
    z = np.asarray(soil_depth, dtype=np.float64)[None, :, None, None]
    return delta_st_clim[None, None] + np.exp(-z / 2.8) * (delta_ts[:, None] - delta_st_clim[None, None])


# ----------------------------------------------------------------------------------------
# a5  surface-pressure fixed-point loop                              step_03:182-319
# ----------------------------------------------------------------------------------------
def adjust_ps_loop(ak, bk, akm, bkm, PS, FIS, T, QV, ta_pgw, hur_pgw, dzg_pref,
                   p_ref=P_REF_INP, adj_factor=ADJ_FACTOR,
                   thresh=THRESH_PHI_REF_MAX_ERROR, max_n_iter=MAX_N_ITER, trace=None):
    """SURVEY appendix A2 with fixed p_ref.

    PS,FIS (time,lat,lon); T,QV,ta_pgw,hur_pgw (time,N,lat,lon); dzg_pref = zg delta [m]
    at plev == p_ref (time,lat,lon).  Returns dict(ps_pgw, hus_pgw, delta_ps, n_iter,
    max_err history).  n_iter = number of passes executed.
    """
    PS = np.asarray(PS, dtype=np.float64)
    level1 = np.arange(1, len(ak) + 1)
    pa_hl_era, _ = hybrid_pressure(ak, bk, PS, akm, bkm)
    delta_ps = np.zeros_like(PS)
    adj_ps = np.zeros_like(PS)
    phi_ref_max_error = np.inf
    it = 1
    hist = []
    n_lowest = ta_pgw.shape[1] - 1
    while phi_ref_max_error > thresh:
        delta_ps = delta_ps + adj_ps                               # :192
        ps_pgw = PS + delta_ps                                     # :193
        pa_hl_pgw, pa_pgw = hybrid_pressure(ak, bk, ps_pgw, akm, bkm)   # :196-199
        hus_pgw = relative_to_specific_humidity(hur_pgw, pa_pgw, ta_pgw)  # :262-266
        phi_ref_pgw = integ_geopot(pa_hl_pgw, FIS, ta_pgw, hus_pgw, level1, p_ref)  # :269-276
        phi_ref_era = integ_geopot(pa_hl_era, FIS, T, QV, level1, p_ref)            # :280-287
        delta_phi_ref = phi_ref_pgw - phi_ref_era                  # :289
        climate_delta_phi_ref = dzg_pref * CON_G                   # :292-295
        phi_ref_error = delta_phi_ref - climate_delta_phi_ref      # :298
        adj_ps = - adj_factor * ps_pgw / (CON_RD * ta_pgw[:, n_lowest]) * phi_ref_error  # :301-304
        a = np.abs(phi_ref_error)
        phi_ref_max_error = np.nanmax(a) if not np.all(np.isnan(a)) else np.nan   # :308 (xarray skipna)
        hist.append(float(phi_ref_max_error))
        if trace is not None:
            trace.append(dict(ps_pgw=ps_pgw.copy(), err=phi_ref_error.copy(), adj=adj_ps.copy()))
        it += 1
        if it > max_n_iter:                                        # :313-319
            raise ValueError('ERROR! Pressure adjustment did not converge')
    return dict(ps_pgw=ps_pgw, hus_pgw=hus_pgw, delta_ps=delta_ps, n_iter=it - 1,
                max_err=hist, phi_ref_era=phi_ref_era, phi_ref_pgw=phi_ref_pgw)


def adjust_ps_loop_local_pref(ak, bk, akm, bkm, PS, FIS, T, QV, ta_pgw, hur_pgw, dzg, plev,
                              adj_factor=ADJ_FACTOR, thresh=THRESH_PHI_REF_MAX_ERROR, max_n_iter=MAX_N_ITER):
    """The loop with p_ref_inp = None (step_03:219-253): the reference pressure is chosen per
    column and per pass as the first plev (file order) below 95 % of both surface pressures, never
    lower in altitude than in the previous pass.  dzg (time, plev, lat, lon) [m] in file order."""
    PS = np.asarray(PS, dtype=np.float64)
    plev = np.asarray(plev, dtype=np.float64)
    level1 = np.arange(1, len(ak) + 1)
    pa_hl_era, _ = hybrid_pressure(ak, bk, PS, akm, bkm)
    delta_ps = np.zeros_like(PS)
    adj_ps = np.zeros_like(PS)
    phi_ref_max_error = np.inf
    it = 1
    hist = []
    p_ref = None
    n_lowest = ta_pgw.shape[1] - 1
    while phi_ref_max_error > thresh:
        delta_ps = delta_ps + adj_ps
        ps_pgw = PS + delta_ps
        pa_hl_pgw, pa_pgw = hybrid_pressure(ak, bk, ps_pgw, akm, bkm)
        p_min_era = pa_hl_era[:, -1] * 0.95                          # :227-228
        p_min_pgw = pa_hl_pgw[:, -1] * 0.95                          # :229-230
        new = np.full(PS.shape, np.nan)
        idx = np.full(PS.shape, -1, dtype=np.int64)
        for k in range(len(plev) - 1, -1, -1):                       # first match in file order wins
            ok = (p_min_era > plev[k]) & (p_min_pgw > plev[k])
            new = np.where(ok, plev[k], new)
            idx = np.where(ok, k, idx)
        if p_ref is not None:                                        # min(p, p_ref_last), :598
            lower = p_ref < new
            new = np.where(lower, p_ref, new)
            idx = np.where(lower, idx_last, idx)
        if np.any(np.isnan(new)):                                    # :245-251
            raise ValueError('No reference pressure level above the required local minimum pressure level '
                             'could not be found everywhere.')
        p_ref, idx_last = new, idx
        hus_pgw = relative_to_specific_humidity(hur_pgw, pa_pgw, ta_pgw)
        phi_ref_pgw = integ_geopot(pa_hl_pgw, FIS, ta_pgw, hus_pgw, level1, p_ref)
        phi_ref_era = integ_geopot(pa_hl_era, FIS, T, QV, level1, p_ref)
        sel = np.take_along_axis(np.asarray(dzg, dtype=np.float64), idx[:, None], axis=1)[:, 0]   # .sel(plev=p_ref), :294
        phi_ref_error = (phi_ref_pgw - phi_ref_era) - sel * CON_G
        adj_ps = - adj_factor * ps_pgw / (CON_RD * ta_pgw[:, n_lowest]) * phi_ref_error
        a = np.abs(phi_ref_error)
        phi_ref_max_error = np.nanmax(a) if not np.all(np.isnan(a)) else np.nan
        hist.append(float(phi_ref_max_error))
        it += 1
        if it > max_n_iter:
            raise ValueError('ERROR! Pressure adjustment did not converge')
    return dict(ps_pgw=ps_pgw, hus_pgw=hus_pgw, delta_ps=delta_ps, n_iter=it - 1, max_err=hist, p_ref=p_ref)


# ----------------------------------------------------------------------------------------
# a10  regrid_lat_lon, xarray branch                                 functions.py:774-893
# ----------------------------------------------------------------------------------------
def interp1d_linear(x, y, x_new, axis):
    """scipy.interpolate.interp1d(kind='linear', bounds_error=False, fill_value=nan,
    assume_sorted=False->sorted) arithmetic, which is what xarray's 1-D `.interp` calls
    (xarray/core/missing.py ScipyInterpolator; reference call sites functions.py:859,892).

    searchsorted(left) -> clip(1, n-1) -> slope = (y_hi-y_lo)/(x_hi-x_lo);
    y = slope*(x_new-x_lo) + y_lo ; outside [x[0], x[-1]] -> NaN.
    """
    x = np.asarray(x, dtype=np.float64)
    x_new = np.asarray(x_new, dtype=np.float64)
    y = np.moveaxis(np.asarray(y, dtype=np.float64), axis, 0)
    if np.any(np.diff(x) < 0):
        # xarray sorts by coordinate before interpolating (assume_sorted=False)
        order = np.argsort(x, kind='stable')
        x = x[order]; y = y[order]
    idx = np.searchsorted(x, x_new)
    idx = idx.clip(1, len(x) - 1).astype(int)
    lo = idx - 1
    hi = idx
    shp = (-1,) + (1,) * (y.ndim - 1)
    x_lo = x[lo].reshape(shp); x_hi = x[hi].reshape(shp)
    y_lo = y[lo]; y_hi = y[hi]
    with np.errstate(invalid='ignore'):
        slope = (y_hi - y_lo) / (x_hi - x_lo)
        y_new = slope * (x_new.reshape(shp) - x_lo) + y_lo
    oob = (x_new < x[0]) | (x_new > x[-1])
    y_new[oob] = np.nan
    return np.moveaxis(y_new, 0, axis)


def regrid_lat_lon(field, src_lat, src_lon, targ_lat, targ_lon):
    """functions.py:774-789, 817-893 on plain arrays.

    field (..., nlat_src, nlon_src).  Returns (..., nlat_t, nlon_t).
    """
    field = np.asarray(field, dtype=np.float64)
    src_lat = np.asarray(src_lat, dtype=np.float64)
    src_lon = np.asarray(src_lon, dtype=np.float64)
    targ_lat = np.asarray(targ_lat, dtype=np.float64)
    targ_lon = np.asarray(targ_lon, dtype=np.float64)
    dlon = np.median(np.diff(src_lon))                     # :778
    dlat = np.median(np.diff(src_lat))                     # :779
    periodic = (dlon + np.max(src_lon) - np.min(src_lon)) >= 359.9   # :780-789
    if src_lat[0] > src_lat[-1]:                           # :822-829
        src_lat = src_lat[::-1]
        field = field[..., ::-1, :]
    if np.max(targ_lat) + dlat > 89.9:                     # :833-837
        north = _zonal_mean(field[..., -1, :])
        field = np.concatenate([field, np.broadcast_to(north[..., None, None],
                                field[..., -1:, :].shape)], axis=-2)
        src_lat = np.concatenate([src_lat, [90.0]])
    if np.min(targ_lat) - dlat < -89.9:                    # :838-842
        south = _zonal_mean(field[..., 0, :])
        field = np.concatenate([np.broadcast_to(south[..., None, None],
                                field[..., :1, :].shape), field], axis=-2)
        src_lat = np.concatenate([[-90.0], src_lat])
    if (np.max(targ_lat) > np.max(src_lat)) | (np.min(targ_lat) < np.min(src_lat)):   # :845-856
        raise ValueError('ERA5 dataset extends further North or South than GCM dataset!')
    field = interp1d_linear(src_lat, field, targ_lat, axis=-2)       # :859
    if periodic:                                            # :866-874
        lon0, f0 = src_lon, field
        if np.max(targ_lon) > np.max(src_lon):
            src_lon = np.concatenate([src_lon, lon0 + 360])
            field = np.concatenate([field, f0], axis=-1)
        if np.min(targ_lon) < np.min(src_lon):
            # the reference prepends the (possibly already extended) dataset shifted by -360
            src_lon_b = src_lon - 360
            field = np.concatenate([field, field], axis=-1)
            src_lon = np.concatenate([src_lon_b, src_lon])
    if (np.max(targ_lon) > np.max(src_lon)) | (np.min(targ_lon) < np.min(src_lon)):   # :877-888
        raise ValueError('ERA5 dataset extends further East or West than GCM dataset!')
    return interp1d_linear(src_lon, field, targ_lon, axis=-1)        # :892


def _zonal_mean(row):
    """xarray .mean(dim=lon) skips NaN (skipna default for floats); all-NaN -> NaN."""
    import warnings
    with warnings.catch_warnings():
        warnings.simplefilter('ignore')
        return np.nanmean(row, axis=-1)


# ----------------------------------------------------------------------------------------
# whole-file restatement on arrays                                   step_03:44-381
# ----------------------------------------------------------------------------------------
def pgw_for_era5_arrays_reinterp(era, deltas, delta_times, plev, target_dt,
                                 ignore_top_pressure_error=False, p_ref=P_REF_INP):
    """pgw_for_era5 with i_reinterp = 1 (step_03:202-216, 330-343) and fixed p_ref; p_ref=None (p_ref_inp = None, the local
    reference level of :219-253): the general restatement of pgw_oracle_refdtype on float64 copies of the inputs."""
    if p_ref is None:
        from . import pgw_oracle_refdtype as R
        f = lambda v: np.asarray(v, dtype=np.float64) if isinstance(v, np.ndarray) and v.dtype.kind == 'f' else v
        return R.pgw_for_era5_arrays_reinterp({k: f(v) for k, v in era.items()}, {k: f(v) for k, v in deltas.items()},
                                              delta_times, plev, target_dt, ignore_top_pressure_error, p_ref=None)
    ak, bk = era['ak'], era['bk']
    akm, bkm = era.get('akm'), era.get('bkm')
    if akm is None:
        akm, bkm = full_level_coeffs(ak, bk)
    f64 = lambda x: np.asarray(x, dtype=np.float64)
    PS, T, QV, FIS = f64(era['PS']), f64(era['T']), f64(era['QV']), f64(era['FIS'])
    pa_hl_era, pa_era = hybrid_pressure(ak, bk, PS, akm, bkm)
    relhum = specific_to_relative_humidity(QV, pa_era, T)
    ld = lambda name: load_delta_values(deltas[name], delta_times[name] if isinstance(delta_times, dict) else delta_times, target_dt)
    level1 = np.arange(1, len(ak) + 1)
    plev = np.asarray(plev, dtype=np.float64)
    kref = int(np.nonzero(plev == p_ref)[0][0])
    dzg = ld('zg')[:, kref]
    era_fields = dict(ta=T, hur=relhum, ua=f64(era['U']), va=f64(era['V']))

    def reinterp(var, pa_pgw):
        e = interp_logp_4d(era_fields[var], pa_era, pa_pgw, extrapolate='constant')         # :209-211
        dsfc, psh = (ld(var + 's'), ld('ps_hist')) if var in ('ta', 'hur') else (None, None)
        return e + vert_interp_delta(ld(var), plev, pa_pgw, dsfc, psh, ignore_top_pressure_error)   # :212-216

    delta_ps = np.zeros_like(PS); adj_ps = np.zeros_like(PS)
    err_max = np.inf
    it = 1
    hist = []
    phi_ref_era = integ_geopot(pa_hl_era, FIS, T, QV, level1, p_ref)
    while err_max > THRESH_PHI_REF_MAX_ERROR:
        delta_ps = delta_ps + adj_ps
        ps_pgw = PS + delta_ps
        pa_hl_pgw, pa_pgw = hybrid_pressure(ak, bk, ps_pgw, akm, bkm)
        ta_pgw, hur_pgw = reinterp('ta', pa_pgw), reinterp('hur', pa_pgw)
        hus_pgw = relative_to_specific_humidity(hur_pgw, pa_pgw, ta_pgw)
        err = (integ_geopot(pa_hl_pgw, FIS, ta_pgw, hus_pgw, level1, p_ref) - phi_ref_era) - dzg * CON_G
        adj_ps = - ADJ_FACTOR * ps_pgw / (CON_RD * ta_pgw[:, -1]) * err
        err_max = np.nanmax(np.abs(err))
        hist.append(float(err_max))
        it += 1
        if it > MAX_N_ITER:
            raise ValueError('ERROR! Pressure adjustment did not converge')
    return dict(PS=ps_pgw, T=ta_pgw, QV=hus_pgw, U=reinterp('ua', pa_pgw), V=reinterp('va', pa_pgw),
                RELHUM_pgw=hur_pgw, n_iter=it - 1, max_err=hist)


def pgw_for_era5_arrays(era, deltas, delta_times, plev, target_dt,
                        ignore_top_pressure_error=False, p_ref=P_REF_INP, vert_interp=None):
    """pgw_for_era5 (step_03:44-381) with i_reinterp = 0 and fixed p_ref on in-memory arrays.

    vert_interp: another implementation of vert_interp_delta with the same arguments (the serial per-column C
    loops of oracle/pgw_oracle_c.py instead of the column-vectorised numpy form; same bits).

    era: dict with ak,bk,[akm,bkm],PS,FIS,T,QV,U,V,T_SKIN,T_SO,FR_LAND,FR_SEA_ICE,soil1
    deltas: dict var -> array [12, (S,) lat, lon] for ta,hur,ua,va,zg,tas,hurs,ts,tos,siconc
            and 'ps_hist' (the HIST ps climatology).
    Returns dict of output fields + n_iter.
    """
    ak, bk = era['ak'], era['bk']
    akm, bkm = era.get('akm'), era.get('bkm')
    if akm is None:
        akm, bkm = full_level_coeffs(ak, bk)
    PS = np.asarray(era['PS'], dtype=np.float64)
    T = np.asarray(era['T'], dtype=np.float64)
    QV = np.asarray(era['QV'], dtype=np.float64)
    _, pa_era = hybrid_pressure(ak, bk, PS, akm, bkm)
    relhum = specific_to_relative_humidity(QV, pa_era, T)             # :91-94

    def ld(name, target=target_dt):
        # every delta file has its own time axis (load_delta per variable, functions.py:195-303): a dict gives them
        return load_delta_values(deltas[name], delta_times[name] if isinstance(delta_times, dict) else delta_times, target)

    out = {}
    sic = sea_ice_update(np.asarray(era['FR_SEA_ICE'], dtype=np.float64), ld('siconc'))   # :103-107
    out['FR_SEA_ICE'] = sic
    delta_ts = ld('ts'); delta_tos = ld('tos')
    comb = integrate_tos(delta_tos, delta_ts, np.asarray(era['FR_LAND'], dtype=np.float64)[0], sic[0])  # :118-123
    out['T_SKIN'] = np.asarray(era['T_SKIN'], dtype=np.float64) + comb                   # :124
    clim = ld('ts', None).mean(axis=0)                                                   # :134-136
    dsoil = soil_temperature_delta(comb, clim, era['soil1'])                             # :139-143
    out['T_SO'] = np.asarray(era['T_SO'], dtype=np.float64) + dsoil                      # :144
    pgw = {}
    era_fields = dict(ta=T, hur=relhum, ua=np.asarray(era['U'], dtype=np.float64),
                      va=np.asarray(era['V'], dtype=np.float64))
    for var in ['ta', 'hur', 'ua', 'va']:                                                # :158-173
        d = ld(var)
        if var in ('ta', 'hur'):
            dsfc = ld(var + 's'); psh = ld('ps_hist')
        else:
            dsfc = None; psh = None
        dint = (vert_interp or vert_interp_delta)(d, plev, pa_era, dsfc, psh, ignore_top_pressure_error)
        pgw[var] = era_fields[var] + dint
    plev = np.asarray(plev, dtype=np.float64)
    kref = np.nonzero(plev == p_ref)[0]
    if len(kref) != 1:
        raise KeyError(p_ref)
    dzg = ld('zg')[:, kref[0]]                                                           # :292-295
    res = adjust_ps_loop(ak, bk, akm, bkm, PS, np.asarray(era['FIS'], dtype=np.float64), T, QV,
                         pgw['ta'], pgw['hur'], dzg, p_ref=p_ref)
    out.update(PS=res['ps_pgw'], T=pgw['ta'], QV=res['hus_pgw'], U=pgw['ua'], V=pgw['va'],
               n_iter=res['n_iter'], max_err=res['max_err'], RELHUM_pgw=pgw['hur'])
    return out


# =====================================================================================
# step_02 `smoothing`: spectral smoothing of a daily annual cycle      functions.py:603-740
# =====================================================================================
def harmonic_tables(lt):
    """cos / sin of the first three harmonics at t = 1..lt (functions.py:716, 727): [3][lt] each."""
    import math
    tv = np.arange(1, lt + 1, 1)
    arg = [2. * math.pi * i / lt * tv for i in (1, 2, 3)]
    return np.stack([np.cos(a) for a in arg]), np.stack([np.sin(a) for a in arg])


def harmonic_ac_analysis(ts):
    """functions.py:672-740: mean + first three harmonics of a series (Storch & Zwiers 12.19-12.23);
    a series with a NaN comes back all NaN (:694-696).  The reference ends with `sys.exit(...)` for series shorter
    than 8 steps (:734-737) - in fact a NameError, `sys` is not imported there; this restatement raises ValueError
    with the intended text."""
    ts = np.asarray(ts)
    if np.isnan(ts).any():
        return np.full_like(ts, np.nan)
    lt = len(ts)
    if not (3 < lt // 2):
        raise ValueError('Whooops that should not be the case for a yearly timeseries! i (reconstruction grade) is '
                         'larger than the number of timeseries elements / 2.')
    mean = ts.mean()                                   # in the dtype of the series (float32 files: float32)
    cos_t, sin_t = harmonic_tables(lt)
    total = 0
    for k in range(3):
        a = 2. / lt * (ts.dot(cos_t[k]))               # :728-730
        b = 2. / lt * (ts.dot(sin_t[k]))
        total = total + (a * cos_t[k] + b * sin_t[k])  # :733, summed by the builtin sum() at :739
    return total + mean


def filter_data_array(diff):
    """filter_data (functions.py:603-669) on an in-memory array (time, [level,] y, x): every column's series is
    replaced by its smoothed version, in the array's own dtype."""
    diff = np.array(diff, copy=True)
    if diff.ndim not in (3, 4):
        raise ValueError('Wrong dimensions of input file should be 3 or 4-D')
    flat = diff.reshape(diff.shape[0], -1)
    for c in range(flat.shape[1]):
        flat[:, c] = harmonic_ac_analysis(flat[:, c])
    return flat.reshape(diff.shape)


# =====================================================================================
# step_02 NaN-ignoring interpolation of ocean-grid deltas (tos, siconc)      functions.py:900-1060
# PARITY UNPINNED: the arithmetic lives in third-party code that is not in /root/reference and not installable here -
# pyproj 3.4.0 `Geod(ellps="WGS84").inv` (Karney's geodesic inverse; environment.yml:198-203) and pyvista 0.37.0
# `PolyData.interpolate` = VTK 9.2.2 vtkPointInterpolator + vtkGaussianKernel.  Restated from their published
# algorithms: geodesic lengths by Vincenty's inverse iteration (Survey Review 1975; agrees with Karney's to < 0.1 mm
# where it converges - it does not for nearly antipodal points, which this oracle refuses), the kernel from
# vtkGaussianKernel::ComputeWeights (w = exp(-(sharpness/radius)^2 d^2) over the points within the radius,
# normalised; a coincident point, d^2 < 256 eps, takes all the weight; no point -> null value).
# =====================================================================================
WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563


def geod_inv_distance(lon1, lat1, lon2, lat2):
    """Third return value of pyproj Geod(ellps='WGS84').inv(lon1, lat1, lon2, lat2): geodesic length [m].
    Scalar Vincenty inverse iteration; exactly antipodal-in-longitude points of equal latitude (the reference's
    lon_offset call, functions.py:969) go over the pole along the meridians."""
    import math
    a, f = WGS84_A, WGS84_F
    b = a * (1 - f)
    if lat1 == lat2 and lon1 == lon2:
        return 0.0
    dl = abs(lon2 - lon1) % 360.0
    dl = 360.0 - dl if dl > 180.0 else dl
    if lat1 == lat2 and dl == 180.0:                       # meridians over the nearer pole
        return 2.0 * (geod_inv_distance(0.0, 0.0, 0.0, 90.0) - geod_inv_distance(0.0, 0.0, 0.0, abs(lat1)))
    U1 = math.atan((1 - f) * math.tan(math.radians(lat1))) if abs(lat1) < 90 else math.copysign(math.pi / 2, lat1)
    U2 = math.atan((1 - f) * math.tan(math.radians(lat2))) if abs(lat2) < 90 else math.copysign(math.pi / 2, lat2)
    L = math.radians(dl)
    lam = L
    for _ in range(500):
        sl, cl = math.sin(lam), math.cos(lam)
        ss = math.sqrt((math.cos(U2) * sl) ** 2 + (math.cos(U1) * math.sin(U2) - math.sin(U1) * math.cos(U2) * cl) ** 2)
        if ss == 0.0:
            return 0.0
        cs = math.sin(U1) * math.sin(U2) + math.cos(U1) * math.cos(U2) * cl
        sig = math.atan2(ss, cs)
        sa = math.cos(U1) * math.cos(U2) * sl / ss
        c2a = 1 - sa * sa
        c2sm = cs - 2 * math.sin(U1) * math.sin(U2) / c2a if c2a > 1e-300 else 0.0
        C = f / 16 * c2a * (4 + f * (4 - 3 * c2a))
        new = L + (1 - C) * f * sa * (sig + C * ss * (c2sm + C * cs * (-1 + 2 * c2sm * c2sm)))
        if abs(new - lam) < 1e-15:
            lam = new
            break
        lam = new
    else:
        raise ValueError('Vincenty inverse did not converge (nearly antipodal points): outside this oracle')
    u2 = c2a * (a * a - b * b) / (b * b)
    A = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)))
    B = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)))
    ds = B * ss * (c2sm + B / 4 * (cs * (-1 + 2 * c2sm ** 2) - B / 6 * c2sm * (-3 + 4 * ss ** 2) * (-3 + 4 * c2sm ** 2)))
    return b * A * (sig - ds)


def planar_metres(lat, lon):
    """functions.py:958-975: (lat_m, lon_m, lon_offset) of points with lon already in (-180, 180]."""
    lat = np.asarray(lat, dtype=np.float64); lon = np.asarray(lon, dtype=np.float64)
    lat_m = np.array([geod_inv_distance(lo, 0.0, lo, la) for la, lo in zip(lat, lon)]) * np.sign(lat)         # :966
    lon_m = np.array([geod_inv_distance(0.0, la, lo, la) for la, lo in zip(lat, lon)]) * np.sign(lon)         # :967
    off = np.array([geod_inv_distance(0.0, la, 180.0, la) for la in lat])                                      # :968
    return lat_m, lon_m, off


def nan_ignoring_interp(land_fr, era5_lat, era5_lon, gcm_lat, gcm_lon, values, kernel_radius, sharpness):
    """functions.py:900-1060 on plain arrays: land_fr (nlat, nlon), era5_lat (nlat), era5_lon (nlon); gcm_lat, gcm_lon,
    values of one common shape.  Brute force over all pairs (small cases)."""
    glat = np.asarray(gcm_lat, dtype=np.float64).reshape(-1)
    glon = np.array(gcm_lon, dtype=np.float64).reshape(-1)
    val = np.asarray(values, dtype=np.float64).reshape(-1)
    glon[glon > 180] -= 360                                                     # :938-941
    ok = ~np.isnan(val)                                                         # :944-948
    val, glon, glat = val[ok], glon[ok], glat[ok]
    lat_m, lon_m, off = planar_metres(glat, glon)
    sx = np.tile(lat_m, 3)                                                      # :977-991
    sy = np.concatenate([lon_m - 2 * off, lon_m, lon_m + 2 * off])
    sv = np.tile(val, 3)
    elat = np.asarray(era5_lat, dtype=np.float64)
    elon = np.array(era5_lon, dtype=np.float64)
    elon[elon > 180] -= 360                                                     # :998-1001
    tlat = np.repeat(elat, len(elon)); tlon = np.tile(elon, len(elat))          # :1004-1005
    tx, ty, _ = planar_metres(tlat, tlon)
    f2 = (sharpness / kernel_radius) ** 2                                       # vtkGaussianKernel: F2 = Sharpness^2 / Radius^2
    r2 = kernel_radius ** 2
    out = np.full(len(tx), np.nan)                                              # null_value = nan, :1041
    tol = 256.0 * np.finfo(np.float64).eps
    for i in range(len(tx)):
        d2 = (tx[i] - sx) ** 2 + (ty[i] - sy) ** 2
        m = d2 <= r2                                                            # FindPointsWithinRadius
        if not m.any():
            continue
        hit = np.nonzero(m & (d2 < tol))[0]
        if len(hit):
            out[i] = sv[hit[0]]
            continue
        w = np.exp(-f2 * d2[m])
        out[i] = np.sum(w / w.sum() * sv[m])                                    # NormalizeWeights, then sum w_i v_i
    out[np.asarray(land_fr, dtype=np.float64).reshape(-1) > 0.7] = np.nan       # :1032, 1055
    return out.reshape(len(elat), len(elon))
