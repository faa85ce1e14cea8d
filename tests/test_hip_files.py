"""GPU tests of the file-level drivers (step_03 / step_02 command lines, load_delta*) on
synthetic NetCDF-3 files, against the oracle evaluated on the same arrays."""
import datetime as dt
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pgw_oracle as O
from oracle import pgw_oracle_refdtype as R

pytestmark = pytest.mark.gpu
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


@pytest.fixture(scope='module')
def files(tmp_path_factory):
    from pgw4era5_amd import synthetic
    root = tmp_path_factory.mktemp('pgw')
    cases = []
    for h in (0, 3):
        c = synthetic.make_case(6, 8, 12, seed=11, dtype=np.float32, target_dt=dt.datetime(2006, 8, 2, h))
        synthetic.write_case_files(c, str(root / 'era'), str(root / 'deltas'))
        cases.append(c)
    return root, cases


def _oracle(c):
    """float32 files in the driver's default mode (settings.f32_file_mode = 'reference'): the oracle that follows numpy's
    promotion through the reference's lines, fed the float32 arrays as they are."""
    return R.pgw_for_era5_arrays(c['era'], c['deltas'], c['delta_times'], c['plev'], c['target_dt'], True)


def test_step03_cli_end_to_end(files):
    from pgw4era5_amd import step_03_apply_to_era as s3, ncio
    root, cases = files
    out_dir = str(root / 'out')
    n_iters = s3._cli(['-i', str(root / 'era'), '-o', out_dir, '-d', str(root / 'deltas'),
                       '-f', '2006080200', '-l', '2006080203', '-H', '3', '-p', '1', '-t'])
    assert len(n_iters) == 2
    for c, n in zip(cases, n_iters):
        want = _oracle(c)
        assert n == want['n_iter']
        ds = ncio.open_dataset(os.path.join(out_dir, 'cas{:%Y%m%d%H}0000.nc'.format(c['target_dt'])), decode_times=False)
        assert 'RELHUM' not in ds
        for name in ['PS', 'T_SKIN', 'T_SO', 'FR_SEA_ICE']:
            assert ds[name].values.dtype == np.float32                        # updated in place in the file's dtype
            np.testing.assert_allclose(ds[name].values, want[name], rtol=1.3e-7, atol=0, equal_nan=True, err_msg=name)
        for name in ['T', 'U', 'V']:
            assert ds[name].values.dtype == np.float64                        # era + delta: the reference writes float64
            np.testing.assert_allclose(ds[name].values, want[name], rtol=1e-9, atol=1e-9, err_msg=name)
        scale = np.nanmax(np.abs(want['QV']), axis=(2, 3), keepdims=True)
        assert ds['QV'].values.dtype == np.float64 and np.nanmax(np.abs(ds['QV'].values - want['QV']) / scale) < 6e-7
        np.testing.assert_array_equal(ds['FIS'].values, c['era']['FIS'])      # untouched variables pass through
        np.testing.assert_array_equal(ds['ak'].values, c['era']['ak'])
        # the attribute-only scalar char variable of COSMO-style files survives, and T still points at it
        assert ds['rotated_pole'].values.dtype == np.dtype('S1') and ds['rotated_pole'].dims == ()
        assert ds['rotated_pole'].attrs['grid_mapping_name'] == 'rotated_latitude_longitude'
        assert ds['rotated_pole'].attrs['grid_north_pole_latitude'] == np.float32(43.0)
        assert ds['T'].attrs['grid_mapping'] == 'rotated_pole'


def test_step03_fill_value_encoded_deltas(tmp_path):
    """delta files as xarray re-encodes CMIP sources - tos over land and ta below ground carry _FillValue
    1e20 instead of NaN.  The driver must decode them like the reference's xr.open_dataset (functions.py:203): same
    output as the same run with NaN-valued files, = the oracle fed NaNs."""
    from pgw4era5_amd import synthetic, step_03_apply_to_era as s3, ncio
    c = synthetic.make_case(6, 8, 12, seed=21, dtype=np.float32, target_dt=dt.datetime(2006, 8, 2, 6))
    assert np.isnan(c['deltas']['tos']).any()
    # "below ground": NaN in the two lowest plev of some columns of the ua delta (ua has no surface insertion)
    low = c['era']['PS'][0] > 95000.0                         # sea-level columns: model levels reach below 925 hPa there
    assert low.sum() > 4
    c['deltas']['ua'][:, :2, low] = np.nan
    synthetic.write_case_files(c, str(tmp_path / 'era'), str(tmp_path / 'deltas'))
    for var in ('tos', 'ua'):                                   # re-encode NaN as a 1e20 fill value with the CF attribute
        path = str(tmp_path / 'deltas' / ('%s_delta.nc' % var))
        ds = ncio.open_dataset(path, decode_times=False)
        v = ds[var].values.copy()
        v[np.isnan(v)] = np.float32(1e20)
        ds[var] = ncio.Field(v, ds[var].dims, ds[var].coords, dict(ds[var].attrs, _FillValue=np.float32(1e20)))
        ncio.to_netcdf(ds, path)
        assert not np.isnan(ncio.open_dataset(path, decode_times=False)[var].values).any()
    s3._DELTASETS.clear()
    out_dir = str(tmp_path / 'out')
    s3._cli(['-i', str(tmp_path / 'era'), '-o', out_dir, '-d', str(tmp_path / 'deltas'),
             '-f', '2006080206', '-l', '2006080206', '-H', '3', '-p', '1', '-t'])
    s3._DELTASETS.clear()
    want = _oracle(c)
    ds = ncio.open_dataset(os.path.join(out_dir, 'cas2006080206' + '0000.nc'), decode_times=False)
    np.testing.assert_allclose(ds['T_SKIN'].values, want['T_SKIN'], rtol=1.3e-7, equal_nan=True)
    assert np.nanmax(ds['T_SKIN'].values) < 400.0                 # no 1e20 blended into the skin temperature
    np.testing.assert_allclose(ds['T_SO'].values, want['T_SO'], rtol=1.3e-7, equal_nan=True)
    np.testing.assert_allclose(ds['U'].values, want['U'], rtol=1e-9, atol=1e-9, equal_nan=True)
    assert np.isnan(ds['U'].values).sum() == np.isnan(want['U']).sum() > 0


def test_step03_daily_deltas_through_the_record_window(tmp_path, monkeypatch):
    """Daily deltas - what step_02 `smoothing` produces - through the command line: 366 float32 records (a leap year's
    calendar: Feb 29 is dropped, functions.py:224-230) for ta, hur, ua, va, zg, tas, hurs beside MONTHLY tos / siconc / ts /
    ps_historical files (every delta file has its own time axis, functions.py:195-303), the record window forced
    (PGW_DELTA_RESIDENT=0: at 0.25 deg these files are 144 GB and cannot stay resident).  ERA5 steps around the dropped
    day and across the end of the year, each against the reference-dtype oracle with its own pass count; the resident
    set writes the same bytes."""
    from pgw4era5_amd import synthetic, step_03_apply_to_era as s3, ncio
    base = synthetic.make_case(6, 8, 12, seed=17, dtype=np.float32)
    day = np.timedelta64(1, 'D')
    daily = np.datetime64('1996-01-01T12:00:00') + np.arange(366) * day
    deltas, times = synthetic.resample_deltas(base, {k: daily for k in ('ta', 'hur', 'ua', 'va', 'zg', 'tas', 'hurs')}, seed=3)
    steps = [dt.datetime(2007, 2, 28, 6), dt.datetime(2007, 2, 28, 18), dt.datetime(2007, 3, 1, 6),      # bracket (Feb 28, Mar 1)
             dt.datetime(2007, 12, 31, 18), dt.datetime(2008, 1, 1, 6), dt.datetime(2008, 1, 1, 12)]      # year wrap; exact record
    cases = []
    for t in steps:
        c = dict(base, deltas=deltas, delta_times=times, target_dt=t)
        synthetic.write_case_files(c, str(tmp_path / 'era'), str(tmp_path / 'deltas'))
        cases.append(c)
    outs = {}
    for resident in ('0', '1'):
        monkeypatch.setenv('PGW_DELTA_RESIDENT', resident)
        s3._DELTASETS.clear()
        out_dir = str(tmp_path / ('out' + resident))
        n_iters = []
        for a, b in (('2007022806', '2007022818'), ('2007030106', '2007030106'), ('2007123118', '2007123118'),
                     ('2008010106', '2008010112')):
            n_iters += s3._cli(['-i', str(tmp_path / 'era'), '-o', out_dir, '-d', str(tmp_path / 'deltas'),
                                '-f', a, '-l', b, '-H', '12' if a != '2008010106' else '6', '-p', '1', '-t'])
        dset = list(s3._DELTASETS.values())[0]
        assert dset.resident == (resident == '1')
        if resident == '0':
            assert not dset.dev and all(len(cache) <= s3.DeltaSet.WINDOW for cache in dset._cache.values())
        outs[resident] = (out_dir, n_iters)
    s3._DELTASETS.clear()
    out_dir, n_iters = outs['0']
    assert len(n_iters) == len(cases)
    for c, n in zip(cases, n_iters):
        want = R.pgw_for_era5_arrays(c['era'], deltas, times, c['plev'], c['target_dt'], True)
        assert n == want['n_iter'], c['target_dt']
        name = 'cas{:%Y%m%d%H}0000.nc'.format(c['target_dt'])
        ds = ncio.open_dataset(os.path.join(out_dir, name), decode_times=False)
        for k in ['PS', 'T_SKIN', 'T_SO', 'FR_SEA_ICE']:
            np.testing.assert_allclose(ds[k].values, want[k], rtol=2.5e-7, atol=0, equal_nan=True, err_msg='%s %s' % (k, name))
        for k in ['T', 'U', 'V']:
            np.testing.assert_allclose(ds[k].values, want[k], rtol=1e-9, atol=1e-9, err_msg='%s %s' % (k, name))
        assert open(os.path.join(out_dir, name), 'rb').read() == open(os.path.join(outs['1'][0], name), 'rb').read()


@pytest.mark.parametrize('p_ref_inp', [30000, None])
def test_step03_cli_with_i_reinterp_on_float32_files(files, monkeypatch, p_ref_inp):
    """settings.i_reinterp = 1 through the command line on float32 files - with the fixed reference level and with
    settings.p_ref_inp = None, independent switches in the reference (step_03_apply_to_era.py:202-253; settings.py:134-150) -
    in the driver's default reference-dtype mode: T, QV, U, V written as float64 like the reference's `era + delta`, every
    file against the reference-dtype oracle's re-interpolating loop with its own pass count."""
    from pgw4era5_amd import step_03_apply_to_era as s3, ncio, settings as S
    root, cases = files
    monkeypatch.setattr(S, 'i_reinterp', 1)
    monkeypatch.setattr(S, 'p_ref_inp', p_ref_inp)
    out_dir = str(root / ('out_reinterp_%s' % p_ref_inp))
    n_iters = s3._cli(['-i', str(root / 'era'), '-o', out_dir, '-d', str(root / 'deltas'),
                       '-f', '2006080200', '-l', '2006080203', '-H', '3', '-p', '1', '-t'])
    for c, n in zip(cases, n_iters):
        want = R.pgw_for_era5_arrays_reinterp(c['era'], c['deltas'], c['delta_times'], c['plev'], c['target_dt'], True,
                                              p_ref=None if p_ref_inp is None else float(p_ref_inp))
        base = _oracle(c)
        assert n == want['n_iter']
        ds = ncio.open_dataset(os.path.join(out_dir, 'cas{:%Y%m%d%H}0000.nc'.format(c['target_dt'])), decode_times=False)
        assert ds['PS'].values.dtype == np.float32 and ds['T'].values.dtype == np.float64 and ds['U'].values.dtype == np.float64
        np.testing.assert_allclose(ds['PS'].values, want['PS'], rtol=2.5e-7, err_msg='PS')
        np.testing.assert_allclose(ds['T'].values, want['T'], rtol=6e-8, err_msg='T')
        for k in ('U', 'V'):
            np.testing.assert_allclose(ds[k].values, want[k], rtol=0, atol=2e-5, err_msg=k)
        scale = np.nanmax(np.abs(want['QV']), axis=(2, 3), keepdims=True)
        assert np.nanmax(np.abs(ds['QV'].values - want['QV']) / scale) < 6e-7
        for k in ('T_SKIN', 'T_SO', 'FR_SEA_ICE'):                              # the surface riders do not depend on the mode
            np.testing.assert_allclose(ds[k].values, base[k], rtol=1.3e-7, atol=0, equal_nan=True, err_msg=k)
        assert np.abs(ds['T'].values - base['T']).max() > 1e-6                  # not the default mode's field


def test_step03_non_convergence_names_the_file(files, monkeypatch):
    """step_03_apply_to_era.py:315-319: the error names the input file and the setting to raise."""
    from pgw4era5_amd import step_03_apply_to_era as s3, settings as S
    root, cases = files
    monkeypatch.setattr(S, 'max_n_iter', 2)
    with pytest.raises(ValueError) as e:
        s3._cli(['-i', str(root / 'era'), '-o', str(root / 'out_nc'), '-d', str(root / 'deltas'),
                 '-f', '2006080200', '-l', '2006080200', '-H', '3', '-t'])
    inp = os.path.join(str(root / 'era'), 'cas20060802000000.nc')
    assert str(e.value) == ('ERROR! Pressure adjustment did not converge for file {}. Consider increasing the value for '
                            '"max_n_iter" in settings.py'.format(inp))


def test_step03_top_pressure_error_without_t(files):
    from pgw4era5_amd import step_03_apply_to_era as s3
    root, cases = files
    with pytest.raises(ValueError) as e:
        s3._cli(['-i', str(root / 'era'), '-o', str(root / 'out2'), '-d', str(root / 'deltas'),
                 '-f', '2006080200', '-l', '2006080200', '-H', '3'])
    assert 'ERA5 top pressure is lower than climate delta top pressure' in str(e.value)


def test_load_delta_and_interp(files):
    from pgw4era5_amd import functions as F
    root, cases = files
    c = cases[1]
    ddir = str(root / 'deltas')
    era_time = np.array([0.0])
    got = F.load_delta(ddir, 'ta', era_time, c['target_dt'])
    want = O.load_delta_values(np.asarray(c['deltas']['ta'], dtype=np.float64), c['delta_times'], c['target_dt'])
    assert got.dims == ('time', 'plev', 'lat', 'lon') and got.shape == want.shape
    np.testing.assert_allclose(got.values, want, rtol=1e-6)
    full = F.load_delta(ddir, 'ts', era_time, None)
    assert full.shape[0] == 12
    exact = F.load_delta(ddir, 'tas', era_time, dt.datetime(2006, 3, 15, 12))
    np.testing.assert_array_equal(exact.values[0], c['deltas']['tas'][2])
    ps_h = F.load_delta(ddir, 'ps', era_time, c['target_dt'], name_base='{}_historical.nc')
    assert ps_h.shape == (1, 6, 8)
    # load_delta_interp == oracle vert_interp_delta of the time-interpolated pieces
    era = c['era']
    _, pa = O.hybrid_pressure(era['ak'], era['bk'], np.asarray(era['PS'], dtype=np.float64))
    gi = F.load_delta_interp(ddir, 'hur', pa, era_time, c['target_dt'], ignore_top_pressure_error=True)
    ld = lambda k: O.load_delta_values(np.asarray(c['deltas'][k], dtype=np.float64), c['delta_times'], c['target_dt'])
    wi = O.vert_interp_delta(ld('hur'), c['plev'], pa, ld('hurs'), ld('ps_hist'), ignore_top_pressure_error=True)
    np.testing.assert_allclose(gi, wi, rtol=1e-6, atol=1e-6)


def test_step02_cli_regridding(tmp_path):
    from pgw4era5_amd import synthetic, ncio, step_02_preproc_deltas as s2
    g = synthetic.make_gcm_grid_case(nlat_src=24, nlon_src=48, nlat=19, nlon=36, nplev=4, ntime=12, seed=5)
    inp, out = tmp_path / 'gcm', tmp_path / 'regridded'
    os.makedirs(inp)
    F = ncio.Field
    times = np.array(['1995-%02d-15T12:00:00' % m for m in range(1, 13)], dtype='datetime64[s]')
    plev = np.array([100000., 85000., 50000., 30000.])
    for base in ('ta_delta.nc', 'ta_historical.nc'):
        ds = ncio.Dataset()
        ds['time'] = F(times, ('time',)); ds['plev'] = F(plev, ('plev',))
        ds['lat'] = F(g['src_lat'], ('lat',)); ds['lon'] = F(g['src_lon'], ('lon',))
        ds['ta'] = F(g['field'], ('time', 'plev', 'lat', 'lon'), attrs=dict(units='K'))
        ncio.to_netcdf(ds, str(inp / base))
    for base in ('tas_delta.nc', 'tas_historical.nc'):           # a 3-D (time, lat, lon) variable, descending-free
        ds = ncio.Dataset()
        ds['time'] = F(times, ('time',))
        ds['lat'] = F(g['src_lat'], ('lat',)); ds['lon'] = F(g['src_lon'], ('lon',))
        ds['tas'] = F(g['field'][:, 0].astype(np.float32), ('time', 'lat', 'lon'))
        ncio.to_netcdf(ds, str(inp / base))
    era = ncio.Dataset()
    era['lat'] = F(g['targ_lat'], ('lat',)); era['lon'] = F(g['targ_lon'], ('lon',))
    era['FR_LAND'] = F(np.zeros((1, 19, 36)), ('time', 'lat', 'lon'))
    era['time'] = F(np.array([0.0]), ('time',))
    ncio.to_netcdf(era, str(tmp_path / 'era.nc'))
    done = s2.main(['regridding', '-i', str(inp), '-o', str(out), '-e', str(tmp_path / 'era.nc'), '-v', 'ta'])
    assert len(done) == 2
    res = ncio.open_dataset(str(out / 'ta_delta.nc'))
    want = O.regrid_lat_lon(g['field'], g['src_lat'], g['src_lon'], g['targ_lat'], g['targ_lon'])
    np.testing.assert_allclose(res['ta'].values, want, rtol=1e-12, atol=1e-14)
    np.testing.assert_array_equal(res['lat'].values, g['targ_lat'])
    np.testing.assert_array_equal(res['plev'].values, plev)
    assert res['ta'].attrs['units'] == 'K'
    done = s2.main(['regridding', '-i', str(inp), '-o', str(out), '-e', str(tmp_path / 'era.nc'), '-v', 'tas'])
    res = ncio.open_dataset(str(out / 'tas_historical.nc'))
    assert res['tas'].dims == ('time', 'lat', 'lon') and res['tas'].dtype == np.float32
    want = O.regrid_lat_lon(g['field'][:, 0].astype(np.float32), g['src_lat'], g['src_lon'], g['targ_lat'], g['targ_lon'])
    np.testing.assert_allclose(res['tas'].values, want, rtol=1e-6, atol=1e-6)
    with pytest.raises(ValueError) as e:                       # no tos files in the input directory (step_02:137-140)
        s2.main(['regridding', '-i', str(inp), '-o', str(out), '-e', str(tmp_path / 'era.nc'), '-v', 'tos'])
    assert 'Files for variable tos are missing' in str(e.value)


def test_step03_cli_two_worker_ranks(files):
    """-p 2: two self-spawned worker processes (one per rank; on a 1-GPU box both bind to GPU 0),
    files dealt round-robin, same outputs as the serial run."""
    from pgw4era5_amd import step_03_apply_to_era as s3, ncio
    root, cases = files
    out_dir = str(root / 'out_p2')
    n_iters = s3._cli(['-i', str(root / 'era'), '-o', out_dir, '-d', str(root / 'deltas'),
                       '-f', '2006080200', '-l', '2006080203', '-H', '3', '-p', '2', '-t'])
    assert len(n_iters) == 2 and all(isinstance(n, int) for n in n_iters)
    serial = str(root / 'out')
    if not os.path.exists(os.path.join(serial, 'cas20060802000000.nc')):
        s3._cli(['-i', str(root / 'era'), '-o', serial, '-d', str(root / 'deltas'),
                 '-f', '2006080200', '-l', '2006080203', '-H', '3', '-p', '1', '-t'])
    for c in cases:
        name = 'cas{:%Y%m%d%H}0000.nc'.format(c['target_dt'])
        a = ncio.open_dataset(os.path.join(out_dir, name), decode_times=False)
        b = ncio.open_dataset(os.path.join(serial, name), decode_times=False)
        for v in ['PS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_SEA_ICE']:
            np.testing.assert_array_equal(a[v].values, b[v].values, err_msg=v)


def test_bench_rank_under_torch_distributed_run_with_rccl(tmp_path):
    """The multi-rank code path of bench.py on hardware, as far as one GPU allows: one rank started by
    torch.distributed.run, process group over RCCL (backend "nccl"), barrier and all-reduces on device memory, torch's HIP
    runtime and libpgw_hip.so in one process; small grid.  (Two ranks cannot share one GPU under RCCL; the N > 1 launch
    path is covered with gloo in tests/test_host_logic.py and rehearsed in profiles/bench_r02d_2rank_gloo_1gpu.json.)"""
    import json
    import subprocess
    import sys
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    env = dict(os.environ, PGW_BENCH_FORCE_DIST='1', HSA_ENABLE_IPC_MODE_LEGACY='0')
    for k in ('WORLD_SIZE', 'RANK', 'LOCAL_RANK'):
        env.pop(k, None)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '1', '--master-addr', '127.0.0.1',
                        '--master-port', '29531', os.path.join(root, 'bench.py'), '--gpus', '1', '--steps', '2', '--warmup', '1',
                        '--nlat', '45', '--nlon', '64', '--nlev', '40', '--no-cpu-baseline', '--no-extras', '--overlap-streams', '0'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-3000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 1 and line['collective']['backend'] == 'rccl (torch nccl)'
    assert line['collective']['ranks_counted_by_all_reduce'] == 1 and line['value'] > 0
    # the latency-mode leg ran with its reduce hook on RCCL (one band; the all-reduce goes through a device tensor)
    lm = line['latency_mode']
    assert 'error' not in lm, lm
    assert lm['bands'] == 1 and lm['ms_per_file'] > 0 and lm['exchange'].endswith('rccl')


def test_step03_24_hourly_files_two_ranks(tmp_path):
    """BASELINE.json configs[2] in rehearsal: 24 hourly ERA5 files (small grid) through `step_03 -p 2` - files dealt
    round-robin to two self-spawned worker ranks (both on GPU 0 of a 1-GPU box), every output file against the
    reference-dtype oracle with its own pass count (reference parallel.py:18-32, step_03_apply_to_era.py:590-638)."""
    from pgw4era5_amd import synthetic, step_03_apply_to_era as s3, ncio
    base = synthetic.make_case(8, 12, 24, seed=100, dtype=np.float32, target_dt=dt.datetime(2006, 1, 15, 0))
    cases = []
    for h in range(24):
        c = dict(base)
        rng = np.random.default_rng(100 + h)             # seed 100 + file index
        era = dict(base['era'])
        era['PS'] = (base['era']['PS'] * (1 + 0.004 * rng.standard_normal(base['era']['PS'].shape))).astype(np.float32)
        era['T'] = (base['era']['T'] + rng.standard_normal(base['era']['T'].shape).astype(np.float32)).astype(np.float32)
        c['era'] = era
        c['target_dt'] = dt.datetime(2006, 1, 15, 0) + dt.timedelta(hours=h)      # crosses a delta record at 12:00
        synthetic.write_case_files(c, str(tmp_path / 'era'), str(tmp_path / 'deltas'))
        cases.append(c)
    out_dir = str(tmp_path / 'out')
    n_iters = s3._cli(['-i', str(tmp_path / 'era'), '-o', out_dir, '-d', str(tmp_path / 'deltas'),
                       '-f', '2006011500', '-l', '2006011523', '-H', '1', '-p', '2', '-t'])
    assert len(n_iters) == 24
    for c, n in zip(cases, n_iters):
        want = _oracle(c)
        assert n == want['n_iter'], c['target_dt']
        ds = ncio.open_dataset(os.path.join(out_dir, 'cas{:%Y%m%d%H}0000.nc'.format(c['target_dt'])), decode_times=False)
        np.testing.assert_allclose(ds['PS'].values, want['PS'], rtol=1.3e-7, err_msg=str(c['target_dt']))
        np.testing.assert_allclose(ds['T'].values, want['T'], rtol=1e-9, atol=1e-9)
        scale = np.nanmax(np.abs(want['QV']), axis=(2, 3), keepdims=True)
        assert np.nanmax(np.abs(ds['QV'].values - want['QV']) / scale) < 6e-7


def test_config5_rehearsal_step02_then_step03_two_ranks(tmp_path):
    """BASELINE.json configs[4] in rehearsal (reduced size): end-to-end step_02 + step_03 through the command lines -
    regridding of ALL default variables from a coarse GCM grid (bilinear; tos / siconc from an ocean model's curvilinear
    grid through the NaN-ignoring interpolation) onto the ERA5 grid, then 48 hourly float32 ERA5 files through
    `step_03 -p 2`; the metric the config names: max |dPS| against the CPU oracle (fed the regridded delta files), here for
    every sixth file, with its own pass count."""
    from pgw4era5_amd import synthetic, step_02_preproc_deltas as s2, step_03_apply_to_era as s3, ncio, settings as S
    gcm, deltas, era_dir, out_dir = (str(tmp_path / d) for d in ('gcm', 'deltas', 'era', 'out'))
    synthetic.write_gcm_files(gcm, nlat=24, nlon=48, seed=7)
    base = synthetic.make_case(16, 24, 30, seed=200, dtype=np.float32, target_dt=dt.datetime(2006, 7, 15, 0))
    cases = []
    for h in range(48):
        c = dict(base)
        rng = np.random.default_rng(200 + h)
        era = dict(base['era'])
        era['PS'] = (base['era']['PS'] * (1 + 0.003 * rng.standard_normal(base['era']['PS'].shape))).astype(np.float32)
        era['T'] = (base['era']['T'] + 0.5 * rng.standard_normal(base['era']['T'].shape).astype(np.float32)).astype(np.float32)
        c['era'] = era
        c['target_dt'] = dt.datetime(2006, 7, 15, 0) + dt.timedelta(hours=h)
        synthetic.write_case_files(c, era_dir, str(tmp_path / 'unused_deltas'))
        cases.append(c)
    example = os.path.join(era_dir, 'cas20060715000000.nc')
    done = s2.main(['regridding', '-i', gcm, '-o', deltas, '-e', example])           # all 11 default variables x 2 periods
    assert len(done) == 22
    s3._DELTASETS.clear()
    n_iters = s3._cli(['-i', era_dir, '-o', out_dir, '-d', deltas, '-f', '2006071500', '-l', '2006071623', '-H', '1', '-p', '2', '-t'])
    assert len(n_iters) == 48
    # the oracle's deltas: what step_02 wrote (decoded like the reference's xr.open_dataset)
    darr, times, plev = {}, None, None
    for var in ('ta', 'hur', 'ua', 'va', 'zg', 'tas', 'hurs', 'ts', 'tos', 'siconc'):
        ds = ncio.open_dataset(os.path.join(deltas, '%s_delta.nc' % var))
        darr[var] = ds[var].values.astype(np.float32)                               # the device holds the deltas in the ERA dtype
        times = ds['time'].values
        if var == 'ta':
            plev = ds['plev'].values
        assert ds[var].shape[-2:] == (16, 24)
    darr['ps_hist'] = ncio.open_dataset(os.path.join(deltas, 'ps_historical.nc'))['ps'].values.astype(np.float32)
    assert np.isnan(darr['tos']).any() and not np.isnan(darr['ta']).any()
    worst = 0.0
    for c, n in list(zip(cases, n_iters))[::6]:
        want = R.pgw_for_era5_arrays(c['era'], darr, times, plev, c['target_dt'], True)
        assert n == want['n_iter'], c['target_dt']
        ds = ncio.open_dataset(os.path.join(out_dir, 'cas{:%Y%m%d%H}0000.nc'.format(c['target_dt'])), decode_times=False)
        worst = max(worst, float(np.max(np.abs(ds['PS'].values.astype(np.float64) - want['PS']) / want['PS'])))
        np.testing.assert_allclose(ds['T'].values, want['T'], rtol=1e-9, atol=1e-9)
        np.testing.assert_allclose(ds['T_SKIN'].values, want['T_SKIN'], rtol=1.3e-7, equal_nan=True)
    print('config 5 rehearsal: 48 files, 2 ranks, max relative |dPS| vs the CPU oracle: %.2e' % worst)
    assert worst <= 2.5e-7


def test_step03_cli_raw_io_path_is_byte_identical(files, monkeypatch):
    """The driver's default I/O path (file bytes pread into pinned buffers, uploaded big-endian, byte order
    converted on the GPU both ways) against PGW_IO_RAW=0 (byte order converted on the host): identical output
    files, and the pinned buffers all return to the pool."""
    from pgw4era5_amd import step_03_apply_to_era as s3, ncio
    from pgw4era5_amd.device import default_context
    root, cases = files
    monkeypatch.setattr(ncio, 'BIG_VARIABLE', 1024)                      # the 6 x 8 x 12 test fields count as large
    args = ['-i', str(root / 'era'), '-d', str(root / 'deltas'), '-f', '2006080200', '-l', '2006080203', '-H', '3', '-p', '1', '-t']
    n_raw = s3._cli(args + ['-o', str(root / 'out_raw')])
    pool = s3._pinned_pool(default_context())
    assert pool.allocated > 0 and sum(len(v) for v in pool._free.values()) == len(pool._owned)
    monkeypatch.setenv('PGW_IO_RAW', '0')
    n_host = s3._cli(args + ['-o', str(root / 'out_host')])
    assert n_raw == n_host
    for c in cases:
        name = 'cas{:%Y%m%d%H}0000.nc'.format(c['target_dt'])
        assert open(root / 'out_raw' / name, 'rb').read() == open(root / 'out_host' / name, 'rb').read()


@pytest.mark.parametrize('raw', ['1', '0'])
def test_step03_float32_outputs_are_the_reference_fields_rounded_once(files, monkeypatch, raw):
    """settings.f32_out_dtype = 'float32': same arithmetic as the reference-dtype mode (pass count, PS, the float64 T, QV, U, V
    on the device), but the four 4-D fields are narrowed on the GPU on the way out: the file holds float32 variables that
    equal the default mode's float64 variables cast to float32, everything else is byte for byte the default file's -
    through the pinned raw path (device-side byte order) and the host path."""
    from pgw4era5_amd import step_03_apply_to_era as s3, ncio, settings as S
    root, cases = files
    monkeypatch.setattr(ncio, 'BIG_VARIABLE', 1024)
    monkeypatch.setenv('PGW_IO_RAW', raw)
    args = ['-i', str(root / 'era'), '-d', str(root / 'deltas'), '-f', '2006080200', '-l', '2006080203', '-H', '3', '-p', '1', '-t']
    n64 = s3._cli(args + ['-o', str(root / ('out64_' + raw))])
    monkeypatch.setattr(S, 'f32_out_dtype', 'float32')
    n32 = s3._cli(args + ['-o', str(root / ('out32_' + raw))])
    assert n64 == n32
    for c in cases:
        name = 'cas{:%Y%m%d%H}0000.nc'.format(c['target_dt'])
        a = ncio.open_dataset(str(root / ('out64_' + raw) / name), decode_times=False)
        b = ncio.open_dataset(str(root / ('out32_' + raw) / name), decode_times=False)
        for v in ('T', 'QV', 'U', 'V'):
            assert a[v].dtype == np.float64 and b[v].dtype == np.float32
            np.testing.assert_array_equal(b[v].values, a[v].values.astype(np.float32))
        for v in ('PS', 'T_SKIN', 'T_SO', 'FR_SEA_ICE', 'FIS', 'FR_LAND', 'ak', 'bk'):
            assert a[v].dtype == b[v].dtype
            np.testing.assert_array_equal(a[v].values, b[v].values)
        assert os.path.getsize(str(root / ('out32_' + raw) / name)) < 0.62 * os.path.getsize(str(root / ('out64_' + raw) / name))
    monkeypatch.setattr(S, 'f32_out_dtype', 'float16')
    with pytest.raises(ValueError):
        s3._cli(args + ['-o', str(root / 'out_bad')])


def test_step02_cli_smoothing(tmp_path):
    """`step_02 smoothing`: daily delta files in, smoothed files out (size-1 dimensions squeezed like the
    reference's `.squeeze()`, coordinates kept), against the oracle."""
    from pgw4era5_amd import step_02_preproc_deltas as s2, ncio
    rng = np.random.default_rng(8)
    inp, out = tmp_path / 'daily', tmp_path / 'smooth'
    inp.mkdir()
    t = np.arange(365.0)
    lat, lon, plev = np.linspace(-60, 60, 5), np.arange(6) * 60.0, np.array([85000., 50000., 20000.])
    data = {}
    for name, fname in (('ta', 'ta_delta.nc'), ('ta', 'ta_historical.nc'), ('tas', 'tas_delta.nc'), ('tas', 'tas_historical.nc')):
        ds = ncio.Dataset(attrs=dict(source='synthetic daily annual cycle'))
        ds['time'] = ncio.Field(t, ('time',), attrs=dict(units='days since 1995-01-01 00:00:00', calendar='noleap'))
        ds['lat'] = ncio.Field(lat, ('lat',)); ds['lon'] = ncio.Field(lon, ('lon',))
        if name == 'ta':
            ds['plev'] = ncio.Field(plev, ('plev',))
            v = (2 + np.sin(2 * np.pi * t / 365)[:, None, None, None] + rng.normal(0, 0.5, (365, 3, 5, 6))).astype(np.float32)
            ds[name] = ncio.Field(v, ('time', 'plev', 'lat', 'lon'), attrs=dict(units='K'))
        else:
            v = (1 + rng.normal(0, 0.5, (365, 1, 5, 6))).astype(np.float32)       # a size-1 height dimension
            ds['height'] = ncio.Field(np.array([2.0]), ('height',))
            ds[name] = ncio.Field(v, ('time', 'height', 'lat', 'lon'), attrs=dict(units='K'))
        data[fname] = v
        ncio.to_netcdf(ds, str(inp / fname))
    done = s2.main(['smoothing', '-i', str(inp), '-o', str(out), '-v', 'ta,tas'])
    assert len(done) == 4
    for fname, v in data.items():
        name = fname.split('_')[0]
        res = ncio.open_dataset(str(out / fname))
        sq = v.reshape([n for n in v.shape if n != 1])
        assert res[name].shape == sq.shape and res[name].dtype == np.float32
        assert res[name].dims == (('time', 'plev', 'lat', 'lon') if name == 'ta' else ('time', 'lat', 'lon'))
        np.testing.assert_allclose(res[name].values, O.filter_data_array(sq), rtol=0, atol=2e-6 * np.abs(v).max())
        np.testing.assert_array_equal(res['lat'].values, lat)
        assert res[name].attrs['units'] == 'K'


BAND_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import torch                                   # before libpgw_hip.so (pgw4era5_amd/_lib.py)
import torch.distributed as dist
from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
from pgw4era5_amd.parallel import band_max_hook
os.environ['LOCAL_RANK'] = '0'                 # both ranks on the one GPU of the box
dist.init_process_group('gloo')
rank, world = dist.get_rank(), dist.get_world_size()
out_dir, mode = sys.argv[1], sys.argv[2]
plev = None
if mode == 'poison_pshist':                    # deltas reaching above the ERA5 model top: the model-top check passes
    plev = np.concatenate([synthetic.PLEV19, [0.1]])
case = synthetic.make_case(nlat=21, nlon=32, nlev=40, seed=11, dtype=np.float64, plev=plev)
if mode == 'poison_pref':
    case['era']['PS'][0, 17, 5] = 20000.0       # band 1 only: p_ref = 300 hPa lies below this "surface"
if mode == 'poison_pshist':
    case['deltas']['ps_hist'][:, 17, 5] = 0.05  # band 1 only: historical surface pressure above the top delta level
from pgw4era5_amd.device import default_context
if mode == 'fail_ws' and rank == 1:
    default_context().set_option('test_fail', 1)          # this band's loop workspace "cannot be allocated"
if mode == 'fail_cont':
    default_context().set_option('loop_guess', 2)         # a first launch of 2 passes, then continuation launches
    if rank == 1:
        default_context().set_option('test_fail', 3)      # ... and band 1 fails before its first continuation launch
if mode == 'fail_setup':
    os.environ['PGW_TEST_FAIL_SETUP'] = '1'               # band 1 raises in its host-side set-up, before the C call
res = None
try:
    for rep in range(2):                       # twice: the second file starts from the first one's pass count (loop_guess)
        res = s3.pgw_for_era5_arrays(case['era'], case['deltas'], case['delta_times'], case['plev'], case['target_dt'],
                                     ignore_top_pressure_error=(sys.argv[3] == 't'), band=(rank, world), reduce_max=band_max_hook(),
                                     p_ref=(None if len(sys.argv) < 5 or sys.argv[4] == 'fixed' else 'local'))
    np.savez(os.path.join(out_dir, 'band%%d.npz' %% rank), n_iter=res['n_iter'], max_err=np.asarray(res['max_err']),
             **{k: res[k] for k in ('PS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_SEA_ICE')})
    msg = 'ok'
except ValueError as e:
    msg = 'ValueError: ' + str(e)
except Exception as e:
    msg = type(e).__name__ + ': ' + str(e)
with open(os.path.join(out_dir, 'msg%%d.txt' %% rank), 'w') as f:
    f.write(msg)
dist.barrier()
dist.destroy_process_group()
'''


def _run_bands(tmp_path, mode, top='t', port='29541', p_ref='fixed'):
    script = tmp_path / 'band.py'
    script.write_text(BAND_SCRIPT % ROOT)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', port, str(script), str(tmp_path), mode, top, p_ref],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR='127.0.0.1'))
    assert r.returncode == 0, r.stderr[-3000:]
    return {k: open(str(tmp_path / ('msg%d.txt' % k))).read() for k in (0, 1) if os.path.exists(str(tmp_path / ('msg%d.txt' % k)))}


@pytest.mark.parametrize('p_ref', ['fixed', 'local'])
def test_one_file_in_two_latitude_bands_is_bit_identical(tmp_path, p_ref):
    """SURVEY.md section 8e, row 2 (latency mode): ONE file split into two latitude bands over two ranks (gloo here, both on
    the one GPU; RCCL on a node), the loop's stopping test made global by an all-reduce MAX of the per-pass maxima
    (pgw_set_reduce_hook).  The bands put together must be the single-process result bit for bit - same pass count, same
    max|err| history (step_03_apply_to_era.py:189, 308: the maximum is over all columns)."""
    from pgw4era5_amd import synthetic, step_03_apply_to_era as s3
    from pgw4era5_amd.parallel import band_rows
    msgs = _run_bands(tmp_path, 'clean', port='29541' if p_ref == 'fixed' else '29547', p_ref=p_ref)
    assert msgs == {0: 'ok', 1: 'ok'}
    case = synthetic.make_case(nlat=21, nlon=32, nlev=40, seed=11, dtype=np.float64)
    whole = s3.pgw_for_era5_arrays(case['era'], case['deltas'], case['delta_times'], case['plev'], case['target_dt'],
                                   ignore_top_pressure_error=True, p_ref=(None if p_ref == 'fixed' else 'local'))
    bands = [np.load(str(tmp_path / ('band%d.npz' % r))) for r in range(2)]
    assert band_rows(21, 0, 2) == (0, 11) and band_rows(21, 1, 2) == (11, 21)
    for b in bands:
        assert int(b['n_iter']) == whole['n_iter']
        np.testing.assert_array_equal(b['max_err'], np.asarray(whole['max_err']))      # the GLOBAL maxima, on both ranks
    # the two bands' own maxima differ, so a band alone would in general stop elsewhere: the exchange matters
    for k in ('PS', 'T', 'QV', 'U', 'V', 'T_SKIN', 'T_SO', 'FR_SEA_ICE'):
        got = np.concatenate([bands[0][k], bands[1][k]], axis=-2)
        np.testing.assert_array_equal(got, whole[k], err_msg=k)


@pytest.mark.parametrize('mode', ['poison_pref', 'poison_pshist'])
def test_a_band_that_fails_fails_all_bands(tmp_path, mode):
    """An error status raised by one band reaches every rank through the same reduce, so all ranks raise the reference's
    ValueError and none waits for a peer that has left.  poison_pref: p_ref below the surface in a column of band 1, found
    by the loop's first launch.  poison_pshist: model-top check ON (a host read-back before the loop) and a column of band 1
    whose historical surface pressure lies above the top delta level (functions.py:360-361, a bare ValueError): band 1
    fails BEFORE the loop and meets band 0 in its first reduce."""
    msgs = _run_bands(tmp_path, mode, 't' if mode == 'poison_pref' else 'check', port='29543' if mode == 'poison_pref' else '29545')
    assert set(msgs) == {0, 1}
    for r in (0, 1):
        if mode == 'poison_pref':
            assert msgs[r].startswith('ValueError: p_ref locally lies below the surface'), msgs
        else:
            assert msgs[r] == 'ValueError: ', msgs


def test_step03_cli_bands_writes_the_one_rank_file_byte_for_byte(files, tmp_path):
    """`step_03 --bands` under torch.distributed.run with two ranks (gloo here, both on the one GPU): every file is read,
    computed and written in two latitude bands - each rank `pread`s its rows of the classic NetCDF layout, the loop's
    stopping test is a MAX all-reduce, each rank `pwrite`s its rows into the output file rank 0 laid out.  The files must
    be byte for byte what one rank writes (the array-level form: test_one_file_in_two_latitude_bands_is_bit_identical)."""
    from pgw4era5_amd import step_03_apply_to_era as s3
    root, cases = files
    one = str(tmp_path / 'one')
    n_one = s3._cli(['-i', str(root / 'era'), '-o', one, '-d', str(root / 'deltas'),
                     '-f', '2006080200', '-l', '2006080203', '-H', '3', '-p', '1', '-t'])
    two = str(tmp_path / 'two')
    log = tmp_path / 'n_iter.txt'
    script = tmp_path / 'run_bands.py'
    script.write_text('import sys, os\nsys.path.insert(0, %r)\nimport torch\n'
                      'from pgw4era5_amd import step_03_apply_to_era as s3\n'
                      'n = s3._cli(sys.argv[1:])\n'
                      'open(%r + os.environ["RANK"], "w").write(repr(n))\n' % (ROOT, str(log)))
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29561', str(script),
                        '-i', str(root / 'era'), '-o', two, '-d', str(root / 'deltas'),
                        '-f', '2006080200', '-l', '2006080203', '-H', '3', '-t', '--bands'],
                       capture_output=True, text=True, timeout=600,
                       env=dict(os.environ, MASTER_ADDR='127.0.0.1', PGW_BANDS_BACKEND='gloo'))
    assert r.returncode == 0, r.stderr[-3000:]
    for rank in ('0', '1'):
        assert eval(open(str(log) + rank).read()) == n_one                  # every rank knows the (global) pass counts
    names = sorted(os.listdir(one))
    assert names == sorted(os.listdir(two)) and len(names) == 2
    for n in names:
        assert open(os.path.join(one, n), 'rb').read() == open(os.path.join(two, n), 'rb').read(), n


@pytest.mark.parametrize('mode', ['fail_ws', 'fail_cont', 'fail_setup'])
def test_a_band_that_stops_on_its_own_does_not_leave_the_others_waiting(tmp_path, mode):
    """Latitude-band mode: a band whose pgw_step03_file stops for a reason of its own - the loop's workspace cannot be had
    (fail_ws, before any reduce), a failure between two loop launches (fail_cont, after the first reduce), its host-side
    set-up raising before the C call (fail_setup) - meets the other band in that band's next reduce with an error status
    (pgw_step03_file's wrapper / pgw_band_abort), so both ranks raise within seconds; without it the healthy band blocks in
    dist.all_reduce until the backend's timeout (gloo: 30 minutes), which the 600 s limit of this test would catch."""
    import time
    t0 = time.time()
    msgs = _run_bands(tmp_path, mode, port={'fail_ws': '29551', 'fail_cont': '29553', 'fail_setup': '29555'}[mode])
    assert time.time() - t0 < 300
    assert set(msgs) == {0, 1}
    assert 'ok' not in msgs.values(), msgs
    if mode == 'fail_setup':
        assert msgs[1].startswith('MemoryError: PGW_TEST_FAIL_SETUP'), msgs
    else:
        assert 'PGW_OPT_TEST_FAIL' in msgs[1], msgs


def test_randomised_file_layout_sweep():
    """tools/fuzz_files.py: the step_03 command line over 80 random ERA5 file layouts (float32 / float64, unlimited or fixed
    `time`, shuffled variables, pass-through extras incl. integer and char variables, akm / bkm in the file, a transposed
    4-D field, raw or converted host I/O): written fields against the oracles, everything else byte for byte."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('fuzz_files', os.path.join(ROOT, 'tools', 'fuzz_files.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    argv = sys.argv
    sys.argv = ['fuzz_files.py', '--cases', '80', '--seed', '4']
    try:
        assert mod.main() == 0
    finally:
        sys.argv = argv
