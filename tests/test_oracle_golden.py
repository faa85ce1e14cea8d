"""Oracle (oracle/pgw_oracle.py) against the vectors produced by the reference's own leaf
functions (oracle/make_golden.py) and against analytic / scipy pins."""
import numpy as np
import pytest

from oracle import pgw_oracle as O


def eq(a, b):
    np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_interp_extrap_1d_known_answers(golden):
    g, meta = golden
    for mode in ['constant', 'linear', 'nan']:
        eq(O.interp_extrap_1d(g['kat_src_x'], g['kat_src_y'], g['kat_targ_x'], mode), g['kat_' + mode])
    np.testing.assert_allclose(g['kat_constant'], [1, 1, 2.48542683, 6.99998557, 7, 7], rtol=1e-8)
    with pytest.raises(ValueError) as e:
        O.interp_extrap_1d(g['kat_src_x'], g['kat_src_y'], g['kat_targ_x'], 'off')
    assert str(e.value) == meta['kat_off_error']


@pytest.mark.parametrize('mode', ['constant', 'linear', 'nan'])
def test_interp_extrap_1d_random_columns(golden, mode):
    g, _ = golden
    sx, sy, tx = g['rnd_src_x'], g['rnd_src_y'], g['rnd_targ_x']
    for c in range(sx.shape[0]):
        eq(O.interp_extrap_1d(sx[c], sy[c], tx[c], mode), g['rnd_' + mode][c])
    # vectorised form used for big cases: same numbers, bit for bit
    out, _ = O.interp_columns_vectorised(sx.T.copy(), sy.T.copy(), tx.T.copy(), mode)
    eq(out.T, g['rnd_' + mode])


def test_interp_off_inrange(golden):
    g, _ = golden
    for k, c in enumerate(g['rnd_off_cases']):
        eq(O.interp_extrap_1d(g['rnd_src_x'][c], g['rnd_src_y'][c], g['rnd_targ_x_inrange'][c], 'off'),
           g['rnd_off'][k])


@pytest.mark.parametrize('mode', ['constant', 'linear', 'nan'])
def test_interp_4d_block(golden, mode):
    g, meta = golden
    v, s, t = g['b4_var'], g['b4_src_lnp'], g['b4_targ_lnp']
    buf = np.zeros_like(g['b4_' + mode])
    O.interp_1d_for_timelatlon(v, s, t, buf, v.shape[0], v.shape[2], v.shape[3], mode)
    eq(buf, g['b4_' + mode])
    # wrapper takes pressures, logs inside (functions.py:470-471)
    for fast in (True, False):
        w = O.interp_logp_4d(v, np.exp(s), np.exp(t), mode, fast=fast)
        np.testing.assert_allclose(w, g['b4_' + mode], rtol=1e-9, atol=1e-12, equal_nan=True)
    bad = s.copy(); bad[0, :, 1, 2] = bad[0, ::-1, 1, 2]
    with pytest.raises(ValueError) as e:
        O.interp_1d_for_timelatlon(v, bad, t, buf, v.shape[0], v.shape[2], v.shape[3], 'constant')
    assert str(e.value) == meta['b4_descending_error']


def test_replace_delta_sfc(golden):
    g, meta = golden
    for i, ps in enumerate(g['rds_ps']):
        P, D = O.replace_delta_sfc(g['rds_plev'], ps, g['rds_delta'], float(g['rds_sfc']))
        eq(P, g['rds_out_P'][i]); eq(D, g['rds_out_D'][i])
    # vectorised column form
    n = len(g['rds_ps'])
    P, D = O.replace_delta_sfc_columns(g['rds_plev'], g['rds_ps'].copy(),
                                       np.repeat(g['rds_delta'][:, None], n, 1),
                                       np.full(n, float(g['rds_sfc'])))
    eq(P.T, g['rds_out_P']); eq(D.T, g['rds_out_D'])
    for ps in meta['rds_errors']:
        with pytest.raises(ValueError):
            O.replace_delta_sfc(g['rds_plev'], float(ps), g['rds_delta'], 9.25)
        with pytest.raises(ValueError):
            O.replace_delta_sfc_columns(g['rds_plev'], np.array([float(ps)]), g['rds_delta'][:, None],
                                        np.array([9.25]))
    # SURVEY 8c known answers
    P, D = O.replace_delta_sfc(np.array([100, 5e4, 8.5e4, 1e5]), 9e4, np.array([1., 2, 3, 4]), 9)
    eq(P, [100, 5e4, 9e4, 1e5]); eq(D, [1, 2, 9, 9])
    P, D = O.replace_delta_sfc(np.array([100, 5e4, 8.5e4, 1e5]), 101000, np.array([1., 2, 3, 4]), 9)
    eq(P, [100, 5e4, 8.5e4, 1.01e5]); eq(D, [1, 2, 3, 9])


def test_determine_p_ref(golden):
    g, _ = golden
    for (a, b, last), want in zip(g['dpr_cases'], g['dpr_out']):
        r = O.determine_p_ref(a, b, g['dpr_opts'], None if np.isnan(last) else last)
        if np.isnan(want):
            assert r is None
        else:
            assert r == want
    assert O.determine_p_ref(95000, 94000, [1e5, 8.5e4, 5e4], None) == 85000


def test_humidity_leaf(golden):
    g, _ = golden
    eq(O.specific_humidity_to_vapor_pressure(g['hum_hus'], g['hum_pa']), g['hum_e'])
    eq(O.vapor_pressure_to_specific_humidity(g['hum_e'], g['hum_pa']), g['hum_q_from_e'])
    eq(O.saturation_vapor_pressure_water_or_ice(g['hum_pa'], g['hum_ta'], True), g['hum_esat_water'])
    eq(O.saturation_vapor_pressure_water_or_ice(g['hum_pa'], g['hum_ta'], False), g['hum_esat_ice'])
    np.testing.assert_allclose(O.saturation_vapor_pressure_water_or_ice(None, np.array([250, 273.16, 300.])),
                               [95.05273377, 611.21, 3531.56496587], rtol=1e-9)


def test_humidity_mixed_phase_and_roundtrip(golden):
    g, _ = golden
    ta, pa, q = g['hum_ta'], g['hum_pa'], g['hum_hus']
    es = O.saturation_vapor_pressure_water_and_ice(pa, ta)
    T0, Ti = 273.16, 250.16
    a = np.where(ta >= T0, 1.0, np.where(ta <= Ti, 0.0, ((ta - Ti) / (T0 - Ti)) ** 2))
    np.testing.assert_allclose(es, a * g['hum_esat_water'] + (1 - a) * g['hum_esat_ice'], rtol=1e-15)
    rh = O.specific_to_relative_humidity(q, pa, ta)
    ok = pa > 10 * O.specific_humidity_to_vapor_pressure(q, pa)
    np.testing.assert_allclose(O.relative_to_specific_humidity(rh, pa, ta)[ok], q[ok], rtol=1e-12, atol=1e-18)
    assert np.isnan(O.saturation_vapor_pressure_water_and_ice(1e5, np.array([np.nan])))[0]


def test_integrate_tos(golden):
    g, _ = golden
    eq(O.integrate_tos(g['tos_tos'], g['tos_ts'], g['tos_land'], g['tos_ice']), g['tos_out'])
    eq(O.integrate_tos(np.array([[1, np.nan]]), np.array([[2., 3]]), np.array([[.25, 0]]),
                       np.array([[.25, 0]])), [[1.5, 3]])


# ---------------- analytic pins for the xarray-bound restatements (parity unpinned) ----------
def _levels(n):
    eta = np.linspace(0, 1, n + 1)
    bk = eta ** 2
    ak = 101325.0 * (eta - bk)
    return ak, bk


def test_integ_geopot_isothermal_dry_is_exact():
    ak, bk = _levels(30)
    rng = np.random.default_rng(3)
    ps = rng.uniform(60000, 104000, (1, 4, 5)); fis = rng.uniform(0, 30000, (1, 4, 5))
    pa_hl, pa = O.hybrid_pressure(ak, bk, ps)
    T = np.full(pa.shape, 250.0); q = np.zeros(pa.shape)
    phi = O.integ_geopot(pa_hl, fis, T, q, np.arange(1, 32), 30000.0)
    np.testing.assert_allclose(phi, fis + O.CON_RD * 250.0 * np.log(ps / 30000.0), rtol=1e-12)
    # per-column p_ref field gives the same as the scalar
    phi2 = O.integ_geopot(pa_hl, fis, T, q, np.arange(1, 32), np.full((1, 4, 5), 30000.0))
    eq(phi, phi2)
    with pytest.raises(ValueError):
        O.integ_geopot(pa_hl, fis, T, q, np.arange(1, 32), 200000.0)


def test_hybrid_pressure_and_coeffs():
    ak, bk = _levels(10)
    akm, bkm = O.full_level_coeffs(ak, bk)
    np.testing.assert_allclose(akm, 0.5 * (ak[1:] + ak[:-1]), rtol=1e-15)
    ps = np.array([[[1e5, 9e4]]])
    pa_hl, pa = O.hybrid_pressure(ak, bk, ps)
    assert pa_hl.shape == (1, 11, 1, 2) and pa.shape == (1, 10, 1, 2)
    eq(pa_hl[0, :, 0, 1], ak + 9e4 * bk)
    eq(pa[0, :, 0, 0], akm + 1e5 * bkm)


def test_interp_linear_profile_reproduced():
    rng = np.random.default_rng(5)
    ps = np.sort(rng.uniform(100, 1e5, (1, 12, 2, 3)), axis=1)
    pt = np.sort(rng.uniform(ps.min(1, keepdims=True), ps.max(1, keepdims=True), (1, 20, 2, 3)), axis=1)
    f = lambda p: 3.0 + 2.0 * np.log(p)
    np.testing.assert_allclose(O.interp_logp_4d(f(ps), ps, pt, 'off'), f(pt), rtol=1e-12)


def test_interp1d_linear_matches_scipy():
    from scipy.interpolate import interp1d
    rng = np.random.default_rng(7)
    x = np.sort(rng.uniform(-90, 90, 17)); y = rng.normal(size=(3, 17, 5)); y[1, 4, 2] = np.nan
    xn = np.concatenate([rng.uniform(-95, 95, 30), x[[0, 5, -1]]])
    want = interp1d(x, y, kind='linear', axis=1, bounds_error=False, fill_value=np.nan)(xn)
    eq(O.interp1d_linear(x, y, xn, axis=1), want)


def test_regrid_linear_field_exact_and_poles():
    slat = np.linspace(-88.5, 88.5, 60)
    slon = np.arange(0, 360, 3.0)
    tlat = np.linspace(-90, 90, 37); tlon = np.arange(0, 360, 2.5)
    lon2, lat2 = np.meshgrid(slon, slat)
    fld = (2.0 + 0.1 * lat2)[None]                       # zonally constant: pole mean exact
    out = O.regrid_lat_lon(fld, slat, slon, tlat, tlon)
    want = 2.0 + 0.1 * np.clip(tlat, -88.5, 88.5)
    np.testing.assert_allclose(out[0], np.repeat(want[:, None], len(tlon), 1), rtol=1e-12, atol=1e-14)
    with pytest.raises(ValueError):
        O.regrid_lat_lon(fld[..., :50], slat, slon[:50], tlat, tlon)   # not periodic, target exceeds
    # reference quirk: dlat is the median diff BEFORE the flip (functions.py:779 vs :822), so
    # a descending source never gets pole rows and a target reaching +-90 raises
    with pytest.raises(ValueError):
        O.regrid_lat_lon(fld[..., ::-1, :], slat[::-1], slon, tlat, tlon)
    inner = O.regrid_lat_lon(fld[..., ::-1, :], slat[::-1], slon, tlat[1:-1], tlon)
    np.testing.assert_allclose(inner[0], out[0, 1:-1], rtol=1e-12, atol=1e-14)
    # periodic wrap: value at lon 358.75 interpolates between 357 and 0(+360)
    fl = (np.cos(np.deg2rad(lon2)))[None]
    o2 = O.regrid_lat_lon(fl, slat, slon, np.array([0.0]), np.array([358.5]))
    f357 = np.cos(np.deg2rad(357.0)); f0 = 1.0
    np.testing.assert_allclose(o2[0, 0, 0], f357 + (f0 - f357) * 0.5, rtol=1e-12)


def test_time_bracket_and_lerp():
    times = np.array(['1995-%02d-15T12:00:00' % m for m in range(1, 13)], dtype='datetime64[s]')
    vals = np.arange(12.0)[:, None] * np.ones((12, 3))
    # inside the year
    v = O.load_delta_values(vals, times, np.datetime64('2006-08-01T00:00:00'))
    ib, ia, tb, ta, _ = O.delta_time_bracket(times, np.datetime64('2006-08-01T00:00:00'))
    assert (ib, ia) == (6, 7) and str(tb).startswith('2006-07-15') and str(ta).startswith('2006-08-15')
    w = (np.datetime64('2006-08-01T00:00:00') - tb) / (ta - tb)
    np.testing.assert_allclose(v[0], 6 + w, rtol=1e-14)
    # wrap before January record / after December record
    ib, ia, tb, ta, _ = O.delta_time_bracket(times, np.datetime64('2006-01-03T00:00:00'))
    assert (ib, ia) == (11, 0) and str(tb).startswith('2005-12-15')
    ib, ia, tb, ta, _ = O.delta_time_bracket(times, np.datetime64('2006-12-31T00:00:00'))
    assert (ib, ia) == (11, 0) and str(ta).startswith('2007-01-15')
    # exact hit
    v = O.load_delta_values(vals, times, np.datetime64('2006-03-15T12:00:00'))
    eq(v[0], vals[2])
    # Feb-29 in a daily file is dropped
    d = np.arange(np.datetime64('1996-02-27'), np.datetime64('1996-03-03')).astype('datetime64[s]')
    _, _, _, _, keep = O.delta_time_bracket(d, np.datetime64('2006-02-28T00:00:00'))
    assert len(keep) == len(d) - 1 and 2 not in keep


def test_loop_converges_and_is_self_consistent():
    from pgw4era5_amd import synthetic as S
    case = S.make_case(nlat=6, nlon=8, nlev=20, seed=0)
    out = O.pgw_for_era5_arrays(case['era'], case['deltas'], case['delta_times'], case['plev'],
                                case['target_dt'], ignore_top_pressure_error=True)
    assert 2 <= out['n_iter'] <= 19
    assert out['max_err'][-1] <= 0.15 < out['max_err'][-2]
    # re-evaluate with the standalone pieces
    era = case['era']
    akm, bkm = O.full_level_coeffs(era['ak'], era['bk'])
    pa_hl, pa = O.hybrid_pressure(era['ak'], era['bk'], out['PS'], akm, bkm)
    np.testing.assert_allclose(O.relative_to_specific_humidity(out['RELHUM_pgw'], pa, out['T']), out['QV'],
                               rtol=1e-14)


# ---------------- independent numerical cross-checks of the xarray-bound restatements ----------------
def test_interp_constant_mode_equals_numpy_interp():
    """'constant' extrapolation == np.interp in ln p (np.interp clamps to the edge values): an independent
    implementation of the same piecewise-linear rule (functions.py:434-580 with extrapolate='constant')."""
    rng = np.random.default_rng(17)
    ps = np.sort(rng.uniform(100, 1e5, (1, 19, 3, 4)), axis=1)
    pt = np.sort(rng.uniform(20, 1.08e5, (1, 60, 3, 4)), axis=1)
    v = rng.normal(size=ps.shape)
    got = O.interp_logp_4d(v, ps, pt, 'constant')
    for j in range(3):
        for i in range(4):
            want = np.interp(np.log(pt[0, :, j, i]), np.log(ps[0, :, j, i]), v[0, :, j, i])
            np.testing.assert_allclose(got[0, :, j, i], want, rtol=1e-12, atol=1e-13)


def test_regrid_equals_scipy_bilinear_in_the_interior():
    """Separable lat-then-lon linear interpolation == joint bilinear interpolation (scipy
    RegularGridInterpolator) wherever no pole row / periodic copy is involved."""
    from scipy.interpolate import RegularGridInterpolator
    rng = np.random.default_rng(19)
    slat = np.linspace(-88.5, 88.5, 60); slon = np.arange(0, 360, 3.0)
    f = rng.normal(size=(60, 120))
    tlat = np.linspace(-80, 80, 33); tlon = np.linspace(1.0, 355.0, 71)
    got = O.regrid_lat_lon(f[None], slat, slon, tlat, tlon)[0]
    rgi = RegularGridInterpolator((slat, slon), f, method='linear')
    la, lo = np.meshgrid(tlat, tlon, indexing='ij')
    want = rgi(np.stack([la.ravel(), lo.ravel()], axis=1)).reshape(la.shape)
    np.testing.assert_allclose(got, want, rtol=1e-11, atol=1e-12)


def test_integ_geopot_against_literal_column_loop():
    """The level-wise vectorised oracle against a literal per-column transcription of SURVEY appendix A1
    (scalar Python, no numpy broadcasting): guards the vectorisation, not the reading of the reference."""
    import math
    rng = np.random.default_rng(23)
    ak, bk = _levels(25)
    ps = rng.uniform(55000, 104000, (1, 3, 4)); fis = rng.uniform(0, 40000, (1, 3, 4))
    pa_hl, pa = O.hybrid_pressure(ak, bk, ps)
    T = 220 + 70 * (pa / 1e5) + rng.normal(size=pa.shape); q = rng.uniform(0, 0.02, pa.shape)
    p_ref = 30000.0
    got = O.integ_geopot(pa_hl, fis, T, q, np.arange(1, 27), p_ref)
    n = 25
    for j in range(3):
        for i in range(4):
            p = [x if x > 0 else 1e-4 for x in pa_hl[0, :, j, i]]
            tv = [T[0, l, j, i] * (1 + 0.61 * q[0, l, j, i]) for l in range(n)]
            phi = [0.0] * (n + 1)
            phi[n] = fis[0, j, i]
            for l in range(n - 1, -1, -1):
                phi[l] = phi[l + 1] + (O.CON_RD * tv[l] * (math.log(p[l + 1]) - math.log(p[l])))
            d = [(p[k] - p_ref) if (p[k] - p_ref) >= 0 else float('nan') for k in range(n + 1)]
            ks = min((k for k in range(n + 1) if d[k] == d[k]), key=lambda k: d[k])
            want = phi[ks] - (O.CON_RD * tv[ks - 1]) * (math.log(p_ref) - math.log(p[ks]))
            assert abs(got[0, j, i] - want) <= 1e-9 * abs(want)


def test_harmonic_ac_analysis_golden(golden_harmonic):
    """step_02 smoothing: the restatement reproduces the reference's outputs bit for bit (float64 and float32
    series of 365 / 366 / 360 / 8 / 9 steps, NaN series); the array form equals the per-column form."""
    g, meta = golden_harmonic
    for lt in (365, 366, 360, 8, 9):
        for k in ('64', '32'):
            ts, want = g['ts%s_%d' % (k, lt)], g['sm%s_%d' % (k, lt)]
            got = np.stack([O.harmonic_ac_analysis(x) for x in ts])
            assert got.dtype == np.dtype(meta['dtype32_%d' % lt]) == np.float64
            eq(got, want)
    np.testing.assert_array_equal(O.harmonic_ac_analysis(g['nan_in']), g['nan_out'])
    assert meta['short_series'].startswith('NameError')          # the reference's sys.exit without `import sys`
    with pytest.raises(ValueError):
        O.harmonic_ac_analysis(np.arange(7.0))
    cube = np.moveaxis(g['ts32_365'].reshape(2, 3, 1, 365), -1, 0).copy()        # (time, level, y, x) float32
    sm = O.filter_data_array(cube)
    assert sm.dtype == np.float32 and sm.shape == cube.shape
    eq(sm[:, 1, 2, 0], g['sm32_365'][5].astype(np.float32))
    with pytest.raises(ValueError):
        O.filter_data_array(np.zeros((8, 3)))
    # a smooth series made of the mean and the first three harmonics only is a fixed point
    t = np.arange(1, 366)
    x = 1.5 + 0.7 * np.cos(2 * np.pi * t / 365) - 0.2 * np.sin(6 * np.pi * t / 365)
    np.testing.assert_allclose(O.harmonic_ac_analysis(x), x, atol=1e-13)
