"""The oracle's C part (oracle/pgw_oracle_c.c: the reference's serial per-column loops) against the vectors produced by
the reference's own functions (tests/golden/ref_leaf_vectors.npz, oracle/make_golden.py), and against the
column-vectorised numpy oracle it cross-checks.  CPU only."""
import numpy as np
import pytest

from oracle import pgw_oracle as O, pgw_oracle_c as C
from pgw4era5_amd import synthetic


def eq(a, b):
    np.testing.assert_array_equal(np.asarray(a), np.asarray(b))


def test_c_interp_extrap_1d_reference_vectors(golden):
    g, meta = golden
    for mode in ['constant', 'linear', 'nan']:
        eq(C.interp_extrap_1d(g['kat_src_x'], g['kat_src_y'], g['kat_targ_x'], mode), g['kat_' + mode])
        for c in range(g['rnd_src_x'].shape[0]):
            eq(C.interp_extrap_1d(g['rnd_src_x'][c], g['rnd_src_y'][c], g['rnd_targ_x'][c], mode), g['rnd_' + mode][c])
    with pytest.raises(ValueError) as e:
        C.interp_extrap_1d(g['kat_src_x'], g['kat_src_y'], g['kat_targ_x'], 'off')
    assert str(e.value) == meta['kat_off_error']
    for k, c in enumerate(g['rnd_off_cases']):
        eq(C.interp_extrap_1d(g['rnd_src_x'][c], g['rnd_src_y'][c], g['rnd_targ_x_inrange'][c], 'off'), g['rnd_off'][k])


@pytest.mark.parametrize('mode', ['constant', 'linear', 'nan'])
def test_c_interp_4d_block_reference_vectors(golden, mode):
    g, meta = golden
    v, s, t = g['b4_var'], g['b4_src_lnp'], g['b4_targ_lnp']
    buf = np.zeros_like(g['b4_' + mode])
    C.interp_1d_for_timelatlon(v, s, t, buf, v.shape[0], v.shape[2], v.shape[3], mode)
    eq(buf, g['b4_' + mode])
    bad = s.copy(); bad[0, :, 1, 2] = bad[0, ::-1, 1, 2]
    with pytest.raises(ValueError) as e:
        C.interp_1d_for_timelatlon(v, bad, t, buf, v.shape[0], v.shape[2], v.shape[3], 'constant')
    assert str(e.value) == meta['b4_descending_error']
    badt = t.copy(); badt[1, :, 2, 4] = badt[1, ::-1, 2, 4]
    with pytest.raises(ValueError, match='Target pressure values must be ascending!'):
        C.interp_1d_for_timelatlon(v, s, badt, buf, v.shape[0], v.shape[2], v.shape[3], 'constant')


def test_c_replace_delta_sfc_reference_vectors(golden):
    g, meta = golden
    for i, ps in enumerate(g['rds_ps']):
        P, D = C.replace_delta_sfc(g['rds_plev'], ps, g['rds_delta'], float(g['rds_sfc']))
        eq(P, g['rds_out_P'][i]); eq(D, g['rds_out_D'][i])
    for ps in list(meta['rds_errors']) + [np.nan]:
        with pytest.raises(ValueError):
            C.replace_delta_sfc(g['rds_plev'], float(ps), g['rds_delta'], 9.25)
    P, D = C.replace_delta_sfc(np.array([100, 5e4, 8.5e4, 1e5]), 9e4, np.array([1., 2, 3, 4]), 9)
    eq(P, [100, 5e4, 9e4, 1e5]); eq(D, [1, 2, 9, 9])


def test_c_column_loops_equal_the_vectorised_oracle_bit_for_bit():
    """interp_logp_4d (all modes, NaNs, exact hits, unsorted targets inside a column) and vert_interp_delta with the
    surface insertion: column-by-column C loops == column-vectorised numpy."""
    rng = np.random.default_rng(5)
    nt, S, N, nlat, nlon = 2, 9, 31, 7, 11
    sp = np.sort(rng.uniform(200., 1.0e5, (nt, S, nlat, nlon)), axis=1)
    tp = np.sort(rng.uniform(50., 1.08e5, (nt, N, nlat, nlon)), axis=1)
    tp[0, 5, 2, 3] = sp[0, 4, 2, 3]                      # exact hit
    tp[1, 3:6, 1, 1] = tp[1, 5:2:-1, 1, 1].copy()        # locally unsorted targets (first <= last still holds)
    var = rng.normal(size=(nt, S, nlat, nlon))
    var[0, 2, 0, 0] = np.nan
    for mode in ['constant', 'linear', 'nan']:
        eq(C.interp_logp_4d(var, sp, tp, mode), O.interp_logp_4d(var, sp, tp, mode))
    inr = np.clip(tp, sp[:, :1], sp[:, -1:])
    eq(C.interp_logp_4d(var, sp, inr, 'off'), O.interp_logp_4d(var, sp, inr, 'off'))
    with pytest.raises(ValueError, match='Extrapolation deactivated'):
        C.interp_logp_4d(var, sp, tp, 'off')
    with pytest.raises(ValueError, match='Invalid input value'):
        C.interp_logp_4d(var, sp, tp, 'cubic')
    with pytest.raises(ValueError, match='Lat dimension'):
        C.interp_logp_4d(var, sp[:, :, :-1], tp, 'nan')
    plev = np.array([1.0e5, 9.25e4, 8.5e4, 7.0e4, 5.0e4, 3.0e4, 1.0e4, 1.0e3, 100.])       # file order (descending)
    delta = rng.normal(size=(nt, len(plev), nlat, nlon))
    ps_hist = rng.uniform(5.2e4, 1.04e5, (nt, nlat, nlon))
    dsfc = rng.normal(size=(nt, nlat, nlon))
    for args in [(None, None), (dsfc, ps_hist)]:
        eq(C.vert_interp_delta(delta, plev, tp, args[0], args[1], True),
           O.vert_interp_delta(delta, plev, tp, args[0], args[1], True))
    with pytest.raises(ValueError, match='ERA5 top pressure is lower'):
        C.vert_interp_delta(delta, plev, tp, dsfc, ps_hist, False)
    ps_bad = ps_hist.copy(); ps_bad[1, 3, 4] = 50.0
    with pytest.raises(ValueError):
        C.vert_interp_delta(delta, plev, tp, dsfc, ps_bad, True)


def test_whole_file_with_c_column_loops_equals_the_numpy_oracle():
    case = synthetic.make_case(nlat=6, nlon=9, nlev=20, seed=3)
    args = (case['era'], case['deltas'], case['delta_times'], case['plev'], case['target_dt'])
    a = O.pgw_for_era5_arrays(*args, ignore_top_pressure_error=True)
    b = O.pgw_for_era5_arrays(*args, ignore_top_pressure_error=True, vert_interp=C.vert_interp_delta)
    assert a['n_iter'] == b['n_iter'] and a['max_err'] == b['max_err']
    for k in ['PS', 'T', 'QV', 'U', 'V']:
        eq(a[k], b[k])
