"""
Generate golden vectors in tests/golden/ by running the REFERENCE's own pure-numpy leaf
functions (test infrastructure; run in the build container only, where /root/reference is
mounted).  The reference cannot be imported as-is (xarray / numba / pyvista / pyproj are
not installed - SURVEY.md section 8c), so those four modules are replaced by empty
placeholder modules (`numba.njit` -> identity decorator) purely so that `import functions`
succeeds; only functions that touch none of them are called.  Nothing of the reference
(source or bytecode) is written anywhere: the outputs are inputs + expected values.

usage:  python oracle/make_golden.py            (writes tests/golden/ref_leaf_vectors.npz)
        python oracle/make_golden.py harmonic   (writes tests/golden/ref_harmonic_vectors.npz only)
        python oracle/make_golden.py f32        (writes tests/golden/ref_leaf_f32_vectors.npz only)
"""
import os
import sys
import types
import json
import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
OUT = os.path.join(HERE, '..', 'tests', 'golden')
REF = '/root/reference'


def import_reference_functions():
    for name in ['xarray', 'pyvista', 'pyproj', 'numba']:
        if name not in sys.modules:
            sys.modules[name] = types.ModuleType(name)
    sys.modules['pyvista'].PolyData = None
    sys.modules['pyproj'].Geod = None
    sys.modules['numba'].njit = lambda *a, **k: (lambda f: f)
    sys.dont_write_bytecode = True
    sys.path.insert(0, REF)
    import functions as F
    return F


def main():
    F = import_reference_functions()
    rng = np.random.default_rng(20261004)
    out = {}
    meta = {}

    # ---- interp_extrap_1d: known answer case of SURVEY 8c + random columns -------------
    src_p = np.array([100, 500, 1e3, 5e3, 1e4, 5e4, 1e5])
    src_y = np.arange(1., 8.)
    trg_p = np.array([50, 100, 700, 99999, 1e5, 101000.])
    out['kat_src_x'] = np.log(src_p); out['kat_src_y'] = src_y; out['kat_targ_x'] = np.log(trg_p)
    for mode in ['constant', 'linear', 'nan']:
        out['kat_' + mode] = F.interp_extrap_1d(np.log(src_p), src_y, np.log(trg_p), mode)
    try:
        F.interp_extrap_1d(np.log(src_p), src_y, np.log(trg_p), 'off')
        meta['kat_off_error'] = None
    except ValueError as e:
        meta['kat_off_error'] = str(e)

    ncase, S, N = 40, 9, 23
    sx = np.empty((ncase, S)); sy = np.empty((ncase, S)); tx = np.empty((ncase, N))
    for c in range(ncase):
        p = np.sort(rng.uniform(50., 101000., S))
        sx[c] = np.log(p)
        sy[c] = rng.normal(0, 3, S)
        t = np.sort(rng.uniform(10., 108000., N))
        # force exact hits, duplicates and edge hits into some cases
        if c % 4 == 1:
            t[3] = p[2]; t[10] = p[0]; t[20] = p[-1]
            t = np.sort(t)
        tx[c] = np.log(t)
        if c % 8 == 3:          # duplicate abscissa as produced by replace_delta_sfc
            sx[c, 5] = sx[c, 6]; sy[c, 5] = sy[c, 6]
        if c % 8 == 5:          # NaN target and NaN source value
            tx[c, 7] = np.nan; sy[c, 4] = np.nan
        if c % 8 == 7:          # non monotone targets (restart of the scan)
            tx[c] = tx[c][rng.permutation(N)]
    out['rnd_src_x'] = sx; out['rnd_src_y'] = sy; out['rnd_targ_x'] = tx
    for mode in ['constant', 'linear', 'nan']:
        out['rnd_' + mode] = np.stack([F.interp_extrap_1d(sx[c], sy[c], tx[c], mode)
                                       for c in range(ncase)])
    # 'off' with in-range targets only
    tx_in = np.stack([np.sort(rng.uniform(sx[c].min(), sx[c].max(), N)) for c in range(ncase)])
    out['rnd_targ_x_inrange'] = tx_in
    ok = [c for c in range(ncase) if c % 8 not in (3,)]
    out['rnd_off_cases'] = np.array(ok)
    out['rnd_off'] = np.stack([F.interp_extrap_1d(sx[c], sy[c], tx_in[c], 'off') for c in ok])

    # ---- interp_1d_for_timelatlon on a small 4-D block ---------------------------------
    nt, S4, N4, nlat, nlon = 2, 7, 11, 3, 5
    p_src = np.sort(rng.uniform(100., 100000., (nt, S4, nlat, nlon)), axis=1)
    p_trg = np.sort(rng.uniform(50., 105000., (nt, N4, nlat, nlon)), axis=1)
    v4 = rng.normal(0, 2, (nt, S4, nlat, nlon))
    out['b4_var'] = v4; out['b4_src_lnp'] = np.log(p_src); out['b4_targ_lnp'] = np.log(p_trg)
    for mode in ['constant', 'linear', 'nan']:
        buf = np.zeros((nt, N4, nlat, nlon))
        F.interp_1d_for_timelatlon(v4, np.log(p_src), np.log(p_trg), buf, nt, nlat, nlon, mode)
        out['b4_' + mode] = buf
    bad = np.log(p_src).copy(); bad[0, :, 1, 2] = bad[0, ::-1, 1, 2]
    try:
        F.interp_1d_for_timelatlon(v4, bad, np.log(p_trg), np.zeros((nt, N4, nlat, nlon)),
                                   nt, nlat, nlon, 'constant')
        meta['b4_descending_error'] = None
    except ValueError as e:
        meta['b4_descending_error'] = str(e)

    # ---- replace_delta_sfc ---------------------------------------------------------------
    plev = np.array([100, 1000, 5000, 10000, 25000, 50000, 70000, 85000, 92500, 100000.])
    ps_cases = np.array([101300., 100000., 99999., 92500., 90000., 60000., 150., 100.0001])
    rp = []; rd = []
    d0 = rng.normal(0, 1, len(plev))
    for ps in ps_cases:
        P, D = F.replace_delta_sfc(plev, ps, d0, 9.25)
        rp.append(P); rd.append(D)
    out['rds_plev'] = plev; out['rds_delta'] = d0; out['rds_ps'] = ps_cases
    out['rds_out_P'] = np.stack(rp); out['rds_out_D'] = np.stack(rd)
    out['rds_sfc'] = np.array(9.25)
    errs = {}
    for ps in [100.0, 50.0]:
        try:
            F.replace_delta_sfc(plev, ps, d0, 9.25)
            errs[str(ps)] = None
        except ValueError:
            errs[str(ps)] = 'ValueError'
    meta['rds_errors'] = errs

    # ---- determine_p_ref -----------------------------------------------------------------
    opts = np.array([100000., 92500, 85000, 70000, 50000, 30000])
    cases = [(95000., 94000., None), (95000., 84000., None), (95000., 94000., 70000.),
             (60000., 99000., 92500.), (20000., 20000., None)]
    res = []
    for a, b, last in cases:
        r = F.determine_p_ref(a, b, opts, last)
        res.append(np.nan if r is None else float(r))
    out['dpr_opts'] = opts
    out['dpr_cases'] = np.array([[a, b, np.nan if l is None else l] for a, b, l in cases])
    out['dpr_out'] = np.array(res)

    # ---- humidity leaf functions ----------------------------------------------------------
    ta = np.concatenate([np.array([180., 250.16, 250.17, 260., 273.15, 273.16, 273.17, 300., 320.]),
                         rng.uniform(185., 315., 40)])
    pa = np.concatenate([np.array([1., 100., 5000., 30000., 50000., 70000., 85000., 101325., 105000.]),
                         rng.uniform(1., 105000., 40)])
    hus = np.concatenate([np.array([0., 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 5e-3, 1e-2, 3e-2]),
                          rng.uniform(0, 2.5e-2, 40)])
    out['hum_ta'] = ta; out['hum_pa'] = pa; out['hum_hus'] = hus
    out['hum_e'] = F.specific_humidity_to_vapor_pressure(hus, pa)
    out['hum_q_from_e'] = F.vapor_pressure_to_specific_humidity(out['hum_e'], pa)
    out['hum_esat_water'] = F.saturation_vapor_pressure_water_or_ice(pa, ta, water=True)
    out['hum_esat_ice'] = F.saturation_vapor_pressure_water_or_ice(pa, ta, water=False)

    # ---- integrate_tos ----------------------------------------------------------------------
    shp = (6, 7)
    tos = rng.normal(1.5, 0.5, shp); tos[rng.uniform(size=shp) < 0.3] = np.nan
    ts = rng.normal(2.5, 0.5, shp)
    land = np.clip(rng.uniform(-0.3, 1.3, shp), 0, 1)
    ice = np.clip(rng.uniform(-0.5, 1.0, shp), 0, 1); ice[rng.uniform(size=shp) < 0.25] = np.nan
    out['tos_tos'] = tos; out['tos_ts'] = ts; out['tos_land'] = land; out['tos_ice'] = ice
    out['tos_out'] = F.integrate_tos(tos.copy(), ts.copy(), land.copy(), ice.copy())

    os.makedirs(OUT, exist_ok=True)
    np.savez_compressed(os.path.join(OUT, 'ref_leaf_vectors.npz'), **out)
    with open(os.path.join(OUT, 'ref_leaf_vectors.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('wrote', len(out), 'arrays;', meta)


def harmonic():
    """harmonic_ac_analysis (reference functions.py:672-740), the per-column kernel of the step_02 `smoothing`
    sub-command: daily annual cycles of 365 / 366 / 360 days and the shortest legal series, float64 and float32
    inputs (the reference keeps the dtype of the file for the mean), a series with a NaN."""
    F = import_reference_functions()
    rng = np.random.default_rng(20261005)
    out, meta = {}, {}
    for lt in (365, 366, 360, 8, 9):
        t = np.arange(lt)
        ts = np.stack([2.0 + 1.5 * np.sin(2 * np.pi * (t + rng.uniform(0, lt)) / lt) + 0.4 * np.cos(4 * np.pi * t / lt)
                       + rng.normal(0, 0.6, lt) for _ in range(6)])
        out['ts64_%d' % lt] = ts
        out['sm64_%d' % lt] = np.stack([F.harmonic_ac_analysis(x.copy()) for x in ts])
        ts32 = ts.astype(np.float32)
        out['ts32_%d' % lt] = ts32
        res = [F.harmonic_ac_analysis(x.copy()) for x in ts32]
        meta['dtype32_%d' % lt] = str(res[0].dtype)
        out['sm32_%d' % lt] = np.stack(res)
    x = out['ts64_365'][0].copy(); x[17] = np.nan
    out['nan_in'] = x
    out['nan_out'] = F.harmonic_ac_analysis(x.copy())
    try:
        F.harmonic_ac_analysis(np.arange(7.0))
        meta['short_series'] = None
    except BaseException as e:       # the reference calls sys.exit(...) without importing sys -> NameError
        meta['short_series'] = '%s: %s' % (type(e).__name__, e)
    np.savez_compressed(os.path.join(OUT, 'ref_harmonic_vectors.npz'), **out)
    with open(os.path.join(OUT, 'ref_harmonic_vectors.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('wrote', len(out), 'arrays;', meta)


def leaf_f32():
    """The same leaf functions on float32 inputs mixed with float64 ones, as they meet on float32 ERA5 files
    (decode_cf=False, step_03:60): pins where numpy's promotion puts float32 arithmetic (oracle/pgw_oracle_refdtype.py).
    Values AND result dtypes are recorded.  Run with this container's numpy (2.2, NEP 50); the reference pins numpy
    1.23.5 - both give float32 for `python float (op) float32 array`, the only mixed case these functions contain."""
    F = import_reference_functions()
    rng = np.random.default_rng(20261006)
    out, meta = {}, {'numpy': np.__version__}
    f32 = np.float32
    ta = np.concatenate([np.array([180., 250.16, 250.17, 260., 273.15, 273.16, 273.17, 300., 320.]),
                         rng.uniform(185., 315., 55)]).astype(f32)
    pa = np.concatenate([np.array([1., 100., 5000., 30000., 50000., 70000., 85000., 101325., 105000.]),
                         rng.uniform(1., 105000., 55)])
    hus = np.concatenate([np.array([0., 1e-7, 1e-6, 1e-5, 1e-4, 1e-3, 5e-3, 1e-2, 3e-2]),
                          rng.uniform(0, 2.5e-2, 55)]).astype(f32)
    out['hum_ta'] = ta; out['hum_pa'] = pa; out['hum_hus'] = hus
    out['hum_e'] = F.specific_humidity_to_vapor_pressure(hus, pa)              # float32 hus, float64 pa
    out['hum_e_allf32'] = F.specific_humidity_to_vapor_pressure(hus, pa.astype(f32))
    out['hum_q_from_e'] = F.vapor_pressure_to_specific_humidity(out['hum_e'], pa)
    out['hum_esat_water'] = F.saturation_vapor_pressure_water_or_ice(pa, ta, water=True)
    out['hum_esat_ice'] = F.saturation_vapor_pressure_water_or_ice(pa, ta, water=False)
    # interp_extrap_1d: float32 values on float64 abscissae (a delta that needed no time interpolation, functions.py:282-283)
    ncase, S, N = 24, 9, 23
    sx = np.empty((ncase, S)); sy = np.empty((ncase, S), dtype=f32); tx = np.empty((ncase, N))
    for c in range(ncase):
        sx[c] = np.log(np.sort(rng.uniform(50., 101000., S)))
        sy[c] = rng.normal(0, 3, S).astype(f32)
        tx[c] = np.log(np.sort(rng.uniform(10., 108000., N)))
    out['int_src_x'] = sx; out['int_src_y'] = sy; out['int_targ_x'] = tx
    for mode in ['constant', 'linear']:
        out['int_' + mode] = np.stack([F.interp_extrap_1d(sx[c], sy[c], tx[c], mode) for c in range(ncase)])
    # replace_delta_sfc keeps the dtype of the delta
    plev = np.array([100, 1000, 5000, 10000, 25000, 50000, 70000, 85000, 92500, 100000.])
    d0 = rng.normal(0, 1, len(plev)).astype(f32)
    ps_cases = np.array([101300., 92500., 90000., 60000.], dtype=f32)
    res = [F.replace_delta_sfc(plev, ps, d0, f32(9.25)) for ps in ps_cases]
    out['rds_plev'] = plev; out['rds_delta'] = d0; out['rds_ps'] = ps_cases
    out['rds_out_P'] = np.stack([r[0] for r in res]); out['rds_out_D'] = np.stack([r[1] for r in res])
    # integrate_tos: float32 land / ice fractions of the ERA5 file with float64 (time-interpolated) deltas, and all float32
    shp = (6, 7)
    tos = rng.normal(1.5, 0.5, shp); tos[rng.uniform(size=shp) < 0.3] = np.nan
    ts = rng.normal(2.5, 0.5, shp)
    land = np.clip(rng.uniform(-0.3, 1.3, shp), 0, 1).astype(f32)
    ice = np.clip(rng.uniform(-0.5, 1.0, shp), 0, 1).astype(f32); ice[rng.uniform(size=shp) < 0.25] = np.nan
    out['tos_tos'] = tos; out['tos_ts'] = ts; out['tos_land'] = land; out['tos_ice'] = ice
    out['tos_out'] = F.integrate_tos(tos.copy(), ts.copy(), land.copy(), ice.copy())
    out['tos_out_allf32'] = F.integrate_tos(tos.astype(f32), ts.astype(f32), land.copy(), ice.copy())
    meta['dtypes'] = {k: str(v.dtype) for k, v in out.items()}
    np.savez_compressed(os.path.join(OUT, 'ref_leaf_f32_vectors.npz'), **out)
    with open(os.path.join(OUT, 'ref_leaf_f32_vectors.json'), 'w') as f:
        json.dump(meta, f, indent=1, sort_keys=True)
    print('wrote', len(out), 'arrays;', meta)


if __name__ == '__main__':
    if len(sys.argv) > 1 and sys.argv[1] == 'harmonic':
        harmonic()
    elif len(sys.argv) > 1 and sys.argv[1] == 'f32':
        leaf_f32()
    else:
        main()
