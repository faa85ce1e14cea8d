"""
ctypes face of oracle/pgw_oracle_c.c  --  TEST INFRASTRUCTURE ONLY.

The reference's serial per-column loops (numba `interp_1d_for_timelatlon` / `interp_extrap_1d`,
functions.py:479-580, and the `np.vectorize`d `replace_delta_sfc`, functions.py:343-366, 395-405) as plain C,
column by column in the reference's loop order.  Same call signatures as the vectorised numpy oracle
(oracle/pgw_oracle.py), which this checks column by column, and what bench.py's `cpu_baseline` times as the
stand-in for serial numba.  Only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline
leg may import this module.  Pin status: see the header of pgw_oracle_c.c.
"""
import ctypes
import os
import subprocess

import numpy as np

_HERE = os.path.dirname(os.path.abspath(__file__))
_SO = os.path.join(_HERE, '_build', 'libpgw_oracle_c.so')
_MODES = {'off': 0, 'linear': 1, 'constant': 2, 'nan': 3}
_lib = None


def build():
    subprocess.check_call(['make', '-C', _HERE, '--quiet'])


def lib():
    global _lib
    if _lib is None:
        if not os.path.exists(_SO):
            build()
        L = ctypes.CDLL(_SO)
        dp, ll = ctypes.POINTER(ctypes.c_double), ctypes.c_longlong
        pd, ci = ctypes.c_ssize_t, ctypes.c_int
        L.pgwc_interp_extrap_1d.argtypes = [dp, pd, dp, pd, ci, dp, pd, ci, dp, pd, ci]
        L.pgwc_interp_1d_for_timelatlon.argtypes = [dp, dp, dp, dp, ci, ci, ci, ci, ci, ci, ctypes.POINTER(ll)]
        L.pgwc_replace_delta_sfc.argtypes = [dp, pd, ci, ctypes.c_double, dp, pd, ctypes.c_double, dp, dp, pd]
        L.pgwc_replace_delta_sfc_columns.argtypes = [dp, ci, dp, dp, dp, dp, dp, ci, ci, ci, ctypes.POINTER(ll)]
        for f in (L.pgwc_interp_extrap_1d, L.pgwc_interp_1d_for_timelatlon, L.pgwc_replace_delta_sfc,
                  L.pgwc_replace_delta_sfc_columns):
            f.restype = ci
        _lib = L
    return _lib


def _p(a):
    return a.ctypes.data_as(ctypes.POINTER(ctypes.c_double))


def _c(a):
    return np.ascontiguousarray(a, dtype=np.float64)


def _raise(rc):
    if rc == 1:
        raise ValueError('Extrapolation deactivated but data out of bounds.')        # functions.py:565-566
    if rc == 2:
        raise ValueError('Source pressure values must be ascending!')                # :501
    if rc == 3:
        raise ValueError('Target pressure values must be ascending!')                # :503
    if rc == 4:
        raise ValueError()                                                           # :361 / empty np.max, :363
    if rc != 0:
        raise RuntimeError('pgw_oracle_c: status %d' % rc)


def interp_extrap_1d(src_x, src_y, targ_x, extrapolate):
    """functions.py:511-580"""
    sx, sy, tx = _c(src_x), _c(src_y), _c(targ_x)
    out = np.zeros(len(tx))
    _raise(lib().pgwc_interp_extrap_1d(_p(sx), 1, _p(sy), 1, len(sx), _p(tx), 1, len(tx), _p(out), 1,
                                       _MODES[extrapolate]))
    return out


def interp_1d_for_timelatlon(orig_array, src_p, targ_p, interp_array, ntime, nlat, nlon, extrapolate):
    """functions.py:479-508 (inputs are ln p; writes interp_array in place, which must be C-contiguous float64)."""
    o, s, t = _c(orig_array), _c(src_p), _c(targ_p)
    assert interp_array.flags.c_contiguous and interp_array.dtype == np.float64
    bad = ctypes.c_longlong(-1)
    _raise(lib().pgwc_interp_1d_for_timelatlon(_p(o), _p(s), _p(t), _p(interp_array), ntime, s.shape[1], t.shape[1],
                                               nlat, nlon, _MODES[extrapolate], ctypes.byref(bad)))


def interp_logp_4d(var, source_P, targ_P, extrapolate='off'):
    """functions.py:434-477 on plain arrays (time, lev, lat, lon), the column loop in C."""
    if extrapolate not in _MODES:
        raise ValueError('Invalid input value for "extrapolate"')
    var, source_P, targ_P = _c(var), _c(source_P), _c(targ_P)
    for ax, name in ((0, 'Time'), (2, 'Lat'), (3, 'Lon')):                         # :447-458
        if var.shape[ax] != source_P.shape[ax] or var.shape[ax] != targ_P.shape[ax]:
            raise ValueError('%s dimension of input files is inconsistent!' % name)
    with np.errstate(invalid='ignore', divide='ignore'):
        lsp, ltp = np.log(source_P), np.log(targ_P)                               # :470-471
    tmp = np.zeros_like(targ_P)                                                    # :465
    interp_1d_for_timelatlon(var, lsp, ltp, tmp, targ_P.shape[0], targ_P.shape[2], targ_P.shape[3], extrapolate)
    return tmp


def replace_delta_sfc(source_P, ps_hist, delta, delta_sfc):
    """functions.py:343-366, one ascending-pressure column."""
    sp, d = _c(source_P), _c(delta)
    oP, oD = np.empty_like(sp), np.empty_like(d)
    _raise(lib().pgwc_replace_delta_sfc(_p(sp), 1, len(sp), float(ps_hist), _p(d), 1, float(delta_sfc), _p(oP), _p(oD), 1))
    return oP, oD


def vert_interp_delta(delta, plev, target_P, delta_sfc=None, ps_hist=None, ignore_top_pressure_error=False):
    """functions.py:369-431 on plain arrays; same arguments as pgw_oracle.vert_interp_delta."""
    delta = _c(np.asarray(delta, dtype=np.float64)[:, ::-1])                      # :383-384
    plev_r = _c(np.asarray(plev, dtype=np.float64)[::-1])
    target_P = _c(target_P)
    nt, S, nlat, nlon = delta.shape
    if delta_sfc is not None:                                                      # :395-404
        oP, oD = np.empty_like(delta), np.empty_like(delta)
        bad = ctypes.c_longlong(-1)
        ps, ds = _c(ps_hist), _c(delta_sfc)
        _raise(lib().pgwc_replace_delta_sfc_columns(_p(plev_r), S, _p(ps), _p(delta), _p(ds), _p(oP), _p(oD),
                                                    nt, nlat, nlon, ctypes.byref(bad)))
        source_P, delta = oP, oD
    else:
        source_P = np.broadcast_to(plev_r[None, :, None, None], delta.shape).copy()  # :387-391
    if np.min(target_P) < np.min(source_P):                                       # :417-425
        if not ignore_top_pressure_error:
            raise ValueError('ERA5 top pressure is lower than climate delta top pressure.')
    return interp_logp_4d(delta, source_P, target_P, extrapolate='constant')       # :429-430
