"""
Reference-dtype CPU oracle  --  TEST INFRASTRUCTURE ONLY (same rules as pgw_oracle.py).

`pgw_oracle.py` casts every input to float64 first.  The reference does not: it hands the file's
arrays to numpy as they are (the ERA5 file is opened with decode_cf=False, step_03:60, so
float32 files stay float32), and numpy's promotion rules then decide, operation by operation,
where float32 arithmetic and float32 roundings happen.  This module restates the same lines
WITHOUT any up-front cast, so that on float32 inputs it computes what the reference computes:

  * hybrid pressures are float64 when ak/bk are float64 (step_03:64-66: float32 PS * float64 bk);
  * the denominator of q -> e, `CON_MW_MD + 0.378*hus`, is float32 for float32 QV
    (functions.py:63: python float * float32 array -> float32), and the saturation pressure of a
    float32 temperature is evaluated entirely in float32, np.exp included (functions.py:74-105);
  * `tav = ta*(1 + 0.61*hus)` and `CON_RD*tav` are float32 for the ERA state (functions.py:144,
    :150) and float64 for the PGW state (ta_pgw = float32 T + float64 delta);
  * `phi_hl` is created from `zgs` (functions.py:141) and therefore STORED in zgs' dtype: on
    float32 files every half-level geopotential is rounded to float32 when it is assigned
    (:149-152), 0.0078 m2/s2 per step at 9e4 m2/s2 - to be compared with the loop's 0.15 threshold;
  * `delta_ps`, `adj_ps` start as `zeros_like(PS)` (step_03:182-184); `delta_ps += adj_ps` is an
    in-place add on a float32 array (numpy casts the float64 sum back, same_kind), so `delta_ps`
    and `ps_pgw = PS + delta_ps` (:193) are float32 in every pass;
  * `-adj_factor * ps_pgw` (:301) is a float32 product;
  * the time interpolation of a float32 delta takes `y_hi - y_lo` in float32 and continues in
    float64 (scipy interp1d._call_linear under xarray's .interp, functions.py:290);
  * the numba column interpolation (functions.py:575-578) takes `src_y[i2] - src_y[i1]` in the
    delta's dtype and the rest in float64.

Promotion rules: the reference pins numpy 1.23.5 (environment.yml:133, value-based casting); this
container has numpy 2.2 (NEP 50).  Both give float32 for `python scalar (op) float32 array`, the
common mixed case on this path - constants enter as python floats below, as they do in the
reference (constants.py, settings.py).  They DIFFER for a 0-d integer ARRAY meeting a float32
array: `xr.where(cond, 1, alpha)` (functions.py:97-100) goes through
`as_shared_dtype([asarray(1), alpha])`; under 1.23.5's value-based casting the 0-d int64 array does
not promote a float32 `alpha` (result float32), under NEP 50 it would (float64).  The restatement
below therefore writes `np.where(cond, 1, alpha).astype(ta.dtype)` - it FORCES the reference's
(1.23.5) result instead of relying on this container's numpy.  numpy float64 SCALARS meeting
float32 arrays would differ between the two versions as well; the path has none (np.log(p_ref)
only meets float64 arrays).

What cannot be pinned bit for bit: np.exp on float32 arrays (SIMD implementation of the numpy
build) - the reference's own last-bit noise in RELHUM of the ERA state.

On float64 inputs every function here returns what pgw_oracle.py returns
(tests/test_oracle_refdtype.py).  Parity pin status: as pgw_oracle.py (leaf functions pinned by
reference-generated fixtures, the xarray-bound flow restated from the cited lines - unpinned
against a reference run).

All `file:line` citations are relative to /root/reference.
"""
import numpy as np

from . import pgw_oracle as O

CON_RD = O.CON_RD
CON_G = O.CON_G
CON_MW_MD = O.CON_MW_MD


def _a(x):
    return np.asarray(x)


# ----------------------------------------------------------------------------------------
# pressures, humidity                                   step_03:64-88 ; functions.py:58-125
# ----------------------------------------------------------------------------------------
def full_level_coeffs(ak, bk):
    """step_03:74-85 in the dtype of ak / bk."""
    ak, bk = _a(ak), _a(bk)
    return 0.5 * (ak[1:] - ak[:-1]) + ak[:-1], 0.5 * (bk[1:] - bk[:-1]) + bk[:-1]


def hybrid_pressure(ak, bk, ps, akm=None, bkm=None):
    """step_03:64-66, 87-88: numpy promotion of (coefficient dtype, ps dtype)."""
    ps = _a(ps)
    if akm is None or bkm is None:
        akm, bkm = full_level_coeffs(ak, bk)
    e = lambda c: _a(c)[None, :, None, None]
    pa_hl = e(ak) + ps[:, None] * e(bk)
    pa = e(akm) + ps[:, None] * e(bkm)
    return pa_hl, pa


def specific_humidity_to_vapor_pressure(hus, pa):
    """functions.py:58-64"""
    return hus * pa / (CON_MW_MD + 0.378 * hus)


def vapor_pressure_to_specific_humidity(vapp, pa):
    """functions.py:66-72"""
    return CON_MW_MD * vapp / (pa - (1 - CON_MW_MD) * vapp)


def saturation_vapor_pressure_water_or_ice(pa, ta, water=True):
    """functions.py:74-89 in the dtype of ta"""
    T0 = 273.16
    a1, a3, a4 = (611.21, 17.502, 32.19) if water else (611.21, 22.587, -0.7)
    with np.errstate(over='ignore', invalid='ignore', divide='ignore'):
        return a1 * np.exp(a3 * (ta - T0) / (ta - a4))


def saturation_vapor_pressure_water_and_ice(pa, ta):
    """functions.py:91-105: alpha = full_like(ta) -> the whole chain stays in ta's dtype.  xr.where(cond, 1, alpha) turns
    the 1 into a 0-d integer array; under the reference's numpy 1.23.5 (value-based casting) that does not promote a float32
    alpha, under this container's numpy 2.2 it would: the .astype(ta.dtype) below forces the reference's result (module
    docstring)."""
    T0, Ti = 273.16, 250.16
    ta = _a(ta)
    alpha = np.full_like(ta, np.nan)
    alpha = np.where(ta >= T0, 1, alpha).astype(ta.dtype, copy=False)
    alpha = np.where(ta <= Ti, 0, alpha).astype(ta.dtype, copy=False)
    with np.errstate(invalid='ignore'):
        mixed = np.power((ta - Ti) / (T0 - Ti), 2.)
    alpha = np.where((ta < T0) & (ta > Ti), mixed, alpha)
    with np.errstate(invalid='ignore', over='ignore'):
        return (alpha * saturation_vapor_pressure_water_or_ice(pa, ta, water=True) +
                (1 - alpha) * saturation_vapor_pressure_water_or_ice(pa, ta, water=False))


def specific_to_relative_humidity(hus, pa, ta):
    """functions.py:107-116"""
    with np.errstate(invalid='ignore', divide='ignore'):
        return (specific_humidity_to_vapor_pressure(_a(hus), _a(pa)) /
                saturation_vapor_pressure_water_and_ice(pa, ta)) * 100


def relative_to_specific_humidity(hur, pa, ta):
    """functions.py:118-125"""
    with np.errstate(invalid='ignore', divide='ignore'):
        vapp = _a(hur) / 100 * saturation_vapor_pressure_water_and_ice(pa, ta)
        return vapor_pressure_to_specific_humidity(vapp, _a(pa))


# ----------------------------------------------------------------------------------------
# integ_geopot                                                          functions.py:128-189
# ----------------------------------------------------------------------------------------
def integ_geopot(pa_hl, zgs, ta, hus, level1, p_ref):
    """As pgw_oracle.integ_geopot, with phi_hl stored in zgs' dtype (functions.py:141, :149) and
    tav / CON_RD*tav in the promoted dtype of (ta, hus) (:144, :150)."""
    pa_hl, zgs, ta, hus = _a(pa_hl), _a(zgs), _a(ta), _a(hus)
    nt, nhl, nlat, nlon = pa_hl.shape
    n = nhl - 1
    if len(level1) != nhl or ta.shape[1] != n:
        raise ValueError('level dimensions are inconsistent')
    with np.errstate(invalid='ignore'):
        p = np.where(pa_hl > 0, pa_hl, 0.0001).astype(pa_hl.dtype, copy=False)   # :135
    lnp = np.log(p)
    dlnpa = lnp[:, 1:] - lnp[:, :-1]                       # :136-138
    tav = ta * (1 + 0.61 * hus)                            # :144  (float32 for float32 ta, hus)
    phi_hl = np.empty(p.shape, dtype=zgs.dtype)            # :141  zgs.expand_dims(...).copy()
    phi_hl[:, n] = zgs
    for l in range(n - 1, -1, -1):                         # :147-152, assignment casts to phi_hl.dtype
        phi_hl[:, l] = phi_hl[:, l + 1] + (CON_RD * tav[:, l] * dlnpa[:, l])
    p_ref_arr = _a(p_ref)
    p_ref_b = p_ref_arr[:, None] if p_ref_arr.ndim == 3 else p_ref_arr
    p_diff = p - p_ref_b                                   # :160
    with np.errstate(invalid='ignore'):
        p_diff = np.where(p_diff >= 0, p_diff, np.nan)     # :161
    if np.any(np.all(np.isnan(p_diff), axis=1)):           # :162-165
        raise ValueError("p_ref locally lies below the surface. Please set a lower "
                         "reference pressue (p_ref_inp) in settings.py")
    ind = np.nanargmin(p_diff, axis=1)
    if np.any(ind == 0):
        raise KeyError(0)
    ind4 = ind[:, None]
    p_ref_star = np.take_along_axis(p, ind4, axis=1)[:, 0]
    phi_ref_star = np.take_along_axis(phi_hl, ind4, axis=1)[:, 0]
    tav_star = np.take_along_axis(tav, ind4 - 1, axis=1)[:, 0]
    # :174-179  python-float p_ref -> np.log gives a float64 scalar, which only meets float64 arrays
    log_pref = np.log(p_ref) if np.isscalar(p_ref) else np.log(p_ref_arr)
    return phi_ref_star - (CON_RD * tav_star) * (log_pref - np.log(p_ref_star))


# ----------------------------------------------------------------------------------------
# deltas: time interpolation and vertical interpolation          functions.py:195-431
# ----------------------------------------------------------------------------------------
def time_lerp(v_before, v_after, t_before, t_after, target):
    """scipy interp1d._call_linear as xarray calls it (functions.py:288-292): y_hi - y_lo in the
    data's dtype, slope and result float64 (x is float64 nanoseconds)."""
    x_hi = float((np.datetime64(t_after).astype('datetime64[ns]') -
                  np.datetime64(t_before).astype('datetime64[ns]')).astype(np.int64))
    x_new = float((np.datetime64(target).astype('datetime64[ns]') -
                   np.datetime64(t_before).astype('datetime64[ns]')).astype(np.int64))
    vb, va = _a(v_before), _a(v_after)
    slope = (va - vb) / np.float64(x_hi - 0.0)
    return slope * np.float64(x_new - 0.0) + vb


def load_delta_values(values, delta_times, target):
    """load_delta (functions.py:195-303) on an in-memory record array, dtype flow as the reference:
    an exact time hit keeps the file dtype (:282-283), an interpolated one is float64."""
    values = _a(values)
    if target is None:
        _, _, _, _, keep = O.delta_time_bracket(delta_times, np.asarray(delta_times)[0])
        return values[keep]
    ib, ia, tb, ta, keep = O.delta_time_bracket(delta_times, target)
    v = values[keep]
    if ib == ia:
        return v[ib][None]
    return time_lerp(v[ib], v[ia], tb, ta, target)[None]


def vert_interp_delta(delta, plev, target_P, delta_sfc=None, ps_hist=None, ignore_top_pressure_error=False):
    """functions.py:369-431: source_P is the float64 plev coordinate; replace_delta_sfc keeps the delta's dtype
    (np.vectorize allocates its outputs from the first call's dtypes); the numba interpolation takes
    `src_y[i2] - src_y[i1]` in the delta's dtype and accumulates in float64 (targ_y = np.zeros)."""
    delta = _a(delta)[:, ::-1]
    plev_r = np.asarray(plev, dtype=np.float64)[::-1]
    target_P = _a(target_P)
    nt, S, nlat, nlon = delta.shape
    source_P = np.broadcast_to(plev_r[None, :, None, None], delta.shape).copy()
    if delta_sfc is not None:
        delta = delta.copy()
        for t in range(nt):
            ps_t = _a(ps_hist[t]).reshape(-1)
            d_t = delta[t].reshape(S, -1)
            with np.errstate(invalid='ignore'):
                gt = ps_t[None, :] > plev_r[:, None]
            if not np.all(gt.any(axis=0)):
                raise ValueError()
            k = S - 1 - np.argmax(gt[::-1], axis=0)
            lev = np.arange(S)[:, None]
            D = np.where(lev >= k[None, :], _a(delta_sfc[t]).reshape(-1)[None, :], d_t).astype(delta.dtype, copy=False)
            P = np.repeat(plev_r[:, None], d_t.shape[1], axis=1)
            P[k, np.arange(d_t.shape[1])] = ps_t
            source_P[t] = P.reshape(S, nlat, nlon)
            delta[t] = D.reshape(S, nlat, nlon)
    if np.min(target_P) < np.min(source_P):                # :417-425
        if not ignore_top_pressure_error:
            raise ValueError('ERA5 top pressure is lower than climate delta top pressure.')
    with np.errstate(invalid='ignore', divide='ignore'):
        lsp, ltp = np.log(source_P), np.log(target_P)
    N = target_P.shape[1]
    out = np.zeros(target_P.shape, dtype=np.result_type(target_P.dtype, np.float32))   # xr.zeros_like(targ_P), :472-473
    for t in range(nt):
        sx = lsp[t].reshape(S, -1); sy = delta[t].reshape(S, -1); tx = ltp[t].reshape(N, -1)
        if np.any(sx[-1] < sx[0]):
            raise ValueError('Source pressure values must be ascending!')
        if np.any(tx[-1] < tx[0]):
            raise ValueError('Target pressure values must be ascending!')
        o, _ = O.interp_columns_vectorised(sx, sy, tx, 'constant')
        out[t] = o.reshape(N, nlat, nlon)
    return out


# ----------------------------------------------------------------------------------------
# surface riders                                      step_03:103-146, functions.py:1145-1186
# ----------------------------------------------------------------------------------------
def integrate_tos(tos_field, ts_field, land_frac, ice_frac):
    """functions.py:1145-1186: output is np.ones(...) (float64); the blend is computed in the inputs' dtypes."""
    tos_field, ts_field, land_frac, ice_frac = _a(tos_field), _a(ts_field), _a(land_frac), _a(ice_frac)
    dims = tos_field.shape
    ice = ice_frac.reshape(-1); tos = tos_field.reshape(-1)
    mask = ~np.isnan(ice) & ~np.isnan(tos)
    out = np.ones(len(tos))
    out[:] = ts_field.reshape(-1)
    frac = np.clip(ice[mask] + land_frac.reshape(-1)[mask], 0, 1)
    out[mask] = np.add(np.multiply(frac, ts_field.reshape(-1)[mask]), np.multiply(1 - frac, tos[mask]))
    return out.reshape(dims)


# ----------------------------------------------------------------------------------------
# the loop                                                              step_03:182-319
# ----------------------------------------------------------------------------------------
def adjust_ps_loop(ak, bk, akm, bkm, PS, FIS, T, QV, ta_pgw, hur_pgw, dzg_pref,
                   p_ref=O.P_REF_INP, adj_factor=O.ADJ_FACTOR, thresh=O.THRESH_PHI_REF_MAX_ERROR,
                   max_n_iter=O.MAX_N_ITER):
    """As pgw_oracle.adjust_ps_loop with the reference's dtype flow: delta_ps / adj_ps start as
    zeros_like(PS) and delta_ps is updated IN PLACE (step_03:182-192), ps_pgw = PS + delta_ps in
    PS' dtype (:193)."""
    PS, FIS, T, QV = _a(PS), _a(FIS), _a(T), _a(QV)
    ta_pgw, hur_pgw = _a(ta_pgw), _a(hur_pgw)
    level1 = np.arange(1, len(ak) + 1)
    pa_hl_era, _ = hybrid_pressure(ak, bk, PS, akm, bkm)
    delta_ps = np.zeros_like(PS)                                        # :182
    adj_ps = np.zeros_like(PS)                                          # :184
    phi_ref_max_error = np.inf
    it = 1
    hist = []
    n_lowest = ta_pgw.shape[1] - 1
    while phi_ref_max_error > thresh:
        np.add(delta_ps, adj_ps, out=delta_ps, casting='same_kind')     # :192  delta_ps += adj_ps
        ps_pgw = PS + delta_ps                                          # :193
        pa_hl_pgw, pa_pgw = hybrid_pressure(ak, bk, ps_pgw, akm, bkm)   # :196-199
        hus_pgw = relative_to_specific_humidity(hur_pgw, pa_pgw, ta_pgw)             # :262-266
        phi_ref_pgw = integ_geopot(pa_hl_pgw, FIS, ta_pgw, hus_pgw, level1, p_ref)   # :269-276
        phi_ref_era = integ_geopot(pa_hl_era, FIS, T, QV, level1, p_ref)             # :280-287
        delta_phi_ref = phi_ref_pgw - phi_ref_era                       # :289
        climate_delta_phi_ref = _a(dzg_pref) * CON_G                    # :292-295
        phi_ref_error = delta_phi_ref - climate_delta_phi_ref           # :298
        adj_ps = - adj_factor * ps_pgw / (CON_RD * ta_pgw[:, n_lowest]) * phi_ref_error   # :301-304
        a = np.abs(phi_ref_error)
        phi_ref_max_error = np.nanmax(a) if not np.all(np.isnan(a)) else np.nan       # :308
        hist.append(float(phi_ref_max_error))
        it += 1
        if it > max_n_iter:                                             # :313-319
            raise ValueError('ERROR! Pressure adjustment did not converge')
    return dict(ps_pgw=ps_pgw, hus_pgw=hus_pgw, delta_ps=delta_ps, n_iter=it - 1, max_err=hist,
                phi_ref_era=phi_ref_era, phi_ref_pgw=phi_ref_pgw)


# ----------------------------------------------------------------------------------------
# whole file                                                              step_03:44-381
# ----------------------------------------------------------------------------------------
def pgw_for_era5_arrays(era, deltas, delta_times, plev, target_dt, ignore_top_pressure_error=False,
                        p_ref=O.P_REF_INP):
    """pgw_oracle.pgw_for_era5_arrays with the reference's dtype flow (i_reinterp = 0, fixed p_ref).
    Outputs carry the dtypes the reference writes: PS, T_SKIN, T_SO, FR_SEA_ICE in the file dtype (in-place
    updates / float32 sums), T, QV, U, V float64 whenever a float64 operand took part (era + delta, step_03:170-173)."""
    ak, bk = _a(era['ak']), _a(era['bk'])
    akm, bkm = era.get('akm'), era.get('bkm')
    if akm is None:
        akm, bkm = full_level_coeffs(ak, bk)
    PS, T, QV = _a(era['PS']), _a(era['T']), _a(era['QV'])
    _, pa_era = hybrid_pressure(ak, bk, PS, akm, bkm)
    relhum = specific_to_relative_humidity(QV, pa_era, T)                 # :91-94

    def ld(name, target=target_dt):
        # every delta file has its own time axis (load_delta per variable, functions.py:195-303): a dict gives them
        return load_delta_values(deltas[name], delta_times[name] if isinstance(delta_times, dict) else delta_times, target)

    out = {}
    sic = np.array(era['FR_SEA_ICE'], copy=True)
    with np.errstate(invalid='ignore'):
        np.add(sic, ld('siconc') / 100, out=sic, casting='same_kind')    # :105  .values += delta/100
    sic = np.clip(sic, 0, 1)                                              # :106-107
    out['FR_SEA_ICE'] = sic
    delta_ts = ld('ts'); delta_tos = ld('tos')
    comb = integrate_tos(delta_tos, delta_ts, _a(era['FR_LAND'])[0], sic[0])   # :118-123 (2-D land / ice of time 0)
    tskin = np.array(era['T_SKIN'], copy=True)
    np.add(tskin, comb, out=tskin, casting='same_kind')                   # :124
    out['T_SKIN'] = tskin
    clim = ld('ts', None).mean(axis=0)                                    # :134-136
    soil1 = _a(era['soil1'])
    z = soil1[None, :, None, None]
    dsoil = clim[None, None] + np.exp(-z / 2.8) * (comb[:, None] - clim[None, None])   # :139-142
    tso = np.array(era['T_SO'], copy=True)
    np.add(tso, dsoil, out=tso, casting='same_kind')                      # :144
    out['T_SO'] = tso
    pgw = {}
    era_fields = dict(ta=T, hur=relhum, ua=_a(era['U']), va=_a(era['V']))
    for var in ['ta', 'hur', 'ua', 'va']:                                 # :158-173
        d = ld(var)
        dsfc, psh = (ld(var + 's'), ld('ps_hist')) if var in ('ta', 'hur') else (None, None)
        pgw[var] = era_fields[var] + vert_interp_delta(d, plev, pa_era, dsfc, psh, ignore_top_pressure_error)
    plev = np.asarray(plev, dtype=np.float64)
    kref = np.nonzero(plev == p_ref)[0]
    if len(kref) != 1:
        raise KeyError(p_ref)
    dzg = ld('zg')[:, kref[0]]                                            # :292-295
    res = adjust_ps_loop(ak, bk, akm, bkm, PS, _a(era['FIS']), T, QV, pgw['ta'], pgw['hur'], dzg, p_ref=p_ref)
    out.update(PS=res['ps_pgw'], T=pgw['ta'], QV=res['hus_pgw'], U=pgw['ua'], V=pgw['va'],
               n_iter=res['n_iter'], max_err=res['max_err'], RELHUM_pgw=pgw['hur'])
    return out


# ----------------------------------------------------------------------------------------
# settings.i_reinterp = 1, fixed or local reference level      step_03:182-343, functions.py:434-477
# ----------------------------------------------------------------------------------------
def interp_logp_4d(var, source_P, targ_P, extrapolate='off'):
    """functions.py:434-477 with numba's dtype flow (:575-578): `src_y[i2] - src_y[i1]` in var's dtype, the rest and the
    result float64 (targ = xr.zeros_like(targ_P), targ_y = np.zeros)."""
    var, source_P, targ_P = _a(var), _a(source_P), _a(targ_P)
    if (var.shape[0] != source_P.shape[0]) or (var.shape[0] != targ_P.shape[0]):
        raise ValueError('Time dimension of input files is inconsistent!')
    nt, N, nlat, nlon = targ_P.shape
    S = var.shape[1]
    with np.errstate(invalid='ignore', divide='ignore'):
        lsp, ltp = np.log(source_P), np.log(targ_P)
    out = np.zeros(targ_P.shape, dtype=targ_P.dtype)
    for t in range(nt):
        sx = lsp[t].reshape(S, -1); sy = var[t].reshape(S, -1); tx = ltp[t].reshape(N, -1)
        if np.any(sx[-1] < sx[0]):
            raise ValueError('Source pressure values must be ascending!')
        if np.any(tx[-1] < tx[0]):
            raise ValueError('Target pressure values must be ascending!')
        o, flag = O.interp_columns_vectorised(sx, sy, tx, extrapolate)
        if extrapolate == 'off' and flag.any():
            raise ValueError('Extrapolation deactivated but data out of bounds.')
        out[t] = o.reshape(N, nlat, nlon)
    return out


def local_p_ref(pa_hl_era, pa_hl_pgw, plev, p_ref_last, idx_last):
    """step_03:219-253 with determine_p_ref (functions.py:583-598) over all columns: the first plev (file order) below 95 %
    of both surface pressures, never lower in altitude than in the previous pass.  Returns (p_ref, index into plev)."""
    p_min_era = pa_hl_era[:, -1] * 0.95                              # :227-228
    p_min_pgw = pa_hl_pgw[:, -1] * 0.95                              # :229-230
    new = np.full(p_min_era.shape, np.nan)
    idx = np.full(p_min_era.shape, -1, dtype=np.int64)
    for k in range(len(plev) - 1, -1, -1):                           # first match in file order wins
        ok = (p_min_era > plev[k]) & (p_min_pgw > plev[k])
        new = np.where(ok, plev[k], new)
        idx = np.where(ok, k, idx)
    if p_ref_last is not None:                                       # min(p, p_ref_last), :598
        lower = p_ref_last < new
        new = np.where(lower, p_ref_last, new)
        idx = np.where(lower, idx_last, idx)
    if np.any(np.isnan(new)):                                        # :245-251
        raise ValueError('No reference pressure level above the required local minimum pressure level '
                         'could not be found everywhere.')
    return new, idx


def pgw_for_era5_arrays_reinterp(era, deltas, delta_times, plev, target_dt, ignore_top_pressure_error=False,
                                 p_ref=O.P_REF_INP, adj_factor=O.ADJ_FACTOR, thresh=O.THRESH_PHI_REF_MAX_ERROR,
                                 max_n_iter=O.MAX_N_ITER):
    """pgw_for_era5 with i_reinterp = 1 (step_03:202-216, 330-343), p_ref fixed or None (p_ref_inp = None: :219-253), in
    the reference's dtype flow: on float32 files interp_logp_4d of the float32 ERA temperature gives float64 with float32
    value differences, RELHUM and every delta are float64, delta_ps / ps_pgw float32, phi_hl float32.  On float64 inputs
    this is pgw_oracle.pgw_for_era5_arrays_reinterp (tests/test_oracle_refdtype.py)."""
    ak, bk = _a(era['ak']), _a(era['bk'])
    akm, bkm = era.get('akm'), era.get('bkm')
    if akm is None:
        akm, bkm = full_level_coeffs(ak, bk)
    PS, T, QV, FIS = _a(era['PS']), _a(era['T']), _a(era['QV']), _a(era['FIS'])
    pa_hl_era, pa_era = hybrid_pressure(ak, bk, PS, akm, bkm)
    relhum = specific_to_relative_humidity(QV, pa_era, T)                 # :91-94
    ld = lambda name: load_delta_values(deltas[name], delta_times[name] if isinstance(delta_times, dict) else delta_times, target_dt)
    level1 = np.arange(1, len(ak) + 1)
    plev = np.asarray(plev, dtype=np.float64)
    zg = ld('zg')
    era_fields = dict(ta=T, hur=relhum, ua=_a(era['U']), va=_a(era['V']))

    def reinterp(var, pa_pgw):
        e = interp_logp_4d(era_fields[var], pa_era, pa_pgw, extrapolate='constant')          # :209-211
        dsfc, psh = (ld(var + 's'), ld('ps_hist')) if var in ('ta', 'hur') else (None, None)
        return e + vert_interp_delta(ld(var), plev, pa_pgw, dsfc, psh, ignore_top_pressure_error)   # :212-216

    delta_ps = np.zeros_like(PS)                                          # :182
    adj_ps = np.zeros_like(PS)                                            # :184
    err_max = np.inf
    it = 1
    hist = []
    pref, idx = None, None
    while err_max > thresh:
        np.add(delta_ps, adj_ps, out=delta_ps, casting='same_kind')       # :192
        ps_pgw = PS + delta_ps                                            # :193
        pa_hl_pgw, pa_pgw = hybrid_pressure(ak, bk, ps_pgw, akm, bkm)
        ta_pgw, hur_pgw = reinterp('ta', pa_pgw), reinterp('hur', pa_pgw)
        if p_ref is None:
            pref, idx = local_p_ref(pa_hl_era, pa_hl_pgw, plev, pref, idx)
            dzg = np.take_along_axis(zg, idx[:, None], axis=1)[:, 0]      # .sel(plev=p_ref), :294
            pr = pref
        else:
            kref = np.nonzero(plev == p_ref)[0]
            if len(kref) != 1:
                raise KeyError(p_ref)
            dzg = zg[:, kref[0]]
            pr = p_ref
        hus_pgw = relative_to_specific_humidity(hur_pgw, pa_pgw, ta_pgw)                     # :262-266
        phi_ref_pgw = integ_geopot(pa_hl_pgw, FIS, ta_pgw, hus_pgw, level1, pr)              # :269-276
        phi_ref_era = integ_geopot(pa_hl_era, FIS, T, QV, level1, pr)                        # :280-287
        err = (phi_ref_pgw - phi_ref_era) - dzg * CON_G                                      # :289-298
        adj_ps = - adj_factor * ps_pgw / (CON_RD * ta_pgw[:, -1]) * err                       # :301-304
        a = np.abs(err)
        err_max = np.nanmax(a) if not np.all(np.isnan(a)) else np.nan                        # :308
        hist.append(float(err_max))
        it += 1
        if it > max_n_iter:                                                                   # :313-319
            raise ValueError('ERROR! Pressure adjustment did not converge')
    return dict(PS=ps_pgw, T=ta_pgw, QV=hus_pgw, U=reinterp('ua', pa_pgw), V=reinterp('va', pa_pgw),
                RELHUM_pgw=hur_pgw, n_iter=it - 1, max_err=hist, p_ref=pref)

