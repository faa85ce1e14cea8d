"""CPU tests of the host-side logic around the hot path: NetCDF-3 labelled I/O, CF time
decoding, delta time bracketing (reference functions.py:224-283), regridding tables
(functions.py:774-893) and the rank-per-GPU launcher (replacement of parallel.py) incl. a
world_size-2 gloo run."""
import datetime as dt
import os
import subprocess
import sys

import numpy as np
import pytest

from oracle import pgw_oracle as O

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def test_ncio_roundtrip_and_labels(tmp_path):
    from pgw4era5_amd import ncio
    ds = ncio.Dataset(attrs=dict(title='t'))
    lat = np.linspace(-10, 10, 5); lon = np.arange(4) * 2.0
    ds['lat'] = ncio.Field(lat, ('lat',)); ds['lon'] = ncio.Field(lon, ('lon',))
    t = np.array(['1995-01-15T12:00:00', '1995-02-14T00:00:00'], dtype='datetime64[s]')
    ds['time'] = ncio.Field(t, ('time',))
    v = np.random.default_rng(0).normal(size=(2, 5, 4)).astype(np.float32); v[0, 1, 2] = np.nan
    ds['tas'] = ncio.Field(v, ('time', 'lat', 'lon'), attrs=dict(units='K'))
    p = str(tmp_path / 'a.nc')
    ncio.to_netcdf(ds, p)
    back = ncio.open_dataset(p)
    np.testing.assert_array_equal(back['tas'].values, v)
    assert back['tas'].dims == ('time', 'lat', 'lon') and back['tas'].attrs['units'] == 'K'
    np.testing.assert_array_equal(back['time'].values, t)
    np.testing.assert_array_equal(back['tas'].coords['lat'], lat)
    raw = ncio.open_dataset(p, decode_times=False)
    assert raw['time'].values.dtype.kind == 'f'
    tr = back['tas'].transpose('lon', 'time', 'lat')
    assert tr.shape == (4, 2, 5)
    with pytest.raises(IOError):
        open(tmp_path / 'b.nc', 'wb').write(b'\x89HDF\r\n\x1a\n' + b'0' * 100)
        ncio.open_dataset(str(tmp_path / 'b.nc'))


def test_native_reader_matches_scipy_reader(tmp_path, monkeypatch):
    """The header parser / pread reader against scipy.io.netcdf_file on files scipy wrote: classic and
    64-bit-offset, fixed and record variables (one and several, odd sizes -> padding), attributes of every
    type; and the raw mode that leaves large variables in the file's byte order."""
    from scipy.io import netcdf_file
    from pgw4era5_amd import ncio
    rng = np.random.default_rng(3)
    for version in (1, 2):
        for nrecvars in (1, 2):
            p = str(tmp_path / ('v%d_%d.nc' % (version, nrecvars)))
            nc = netcdf_file(p, 'w', version=version)
            nc.title = 'demo'; nc.scale = np.float32(2.5); nc.ids = np.array([1, 2, 3], dtype=np.int32)
            nc.createDimension('time', None); nc.createDimension('x', 5); nc.createDimension('y', 3)
            t = nc.createVariable('time', 'd', ('time',)); t.units = 'hours since 2000-01-01 00:00:00'
            a = nc.createVariable('a', 'h', ('time', 'x'))                     # 10 bytes per record: padded unless alone
            a.missing_value = np.int16(-9)
            fx = nc.createVariable('fixed', 'f', ('y', 'x')); fx.units = 'K'
            if nrecvars == 2:
                b = nc.createVariable('b', 'f', ('time', 'y', 'x'))
                b[:] = rng.normal(size=(4, 3, 5)).astype('f4')
            t[:] = np.arange(4.0); a[:] = rng.integers(-100, 100, size=(4, 5)); fx[:] = rng.normal(size=(3, 5))
            nc.close()
            mine = ncio.open_dataset(p, decode_times=False)
            monkeypatch.setenv('PGW_NC_READER', 'scipy')
            ref = ncio.open_dataset(p, decode_times=False)
            monkeypatch.delenv('PGW_NC_READER')
            assert list(mine.variables) == list(ref.variables)
            assert mine.attrs['title'] == 'demo' and mine.attrs['scale'] == np.float32(2.5)
            np.testing.assert_array_equal(mine.attrs['ids'], [1, 2, 3])
            for k in ref.variables:
                assert mine[k].dims == ref[k].dims and mine[k].values.dtype == ref[k].values.dtype, k
                np.testing.assert_array_equal(mine[k].values, ref[k].values, err_msg=k)
                assert set(mine[k].attrs) == set(ref[k].attrs)
            assert mine['a'].attrs['missing_value'] == -9
            dec = ncio.open_dataset(p)
            assert dec['time'].values[1] == np.datetime64('2000-01-01T01:00:00')
    # raw mode: variables above the threshold stay big-endian in caller-supplied buffers, the writer takes them as is
    monkeypatch.setattr(ncio, 'BIG_VARIABLE', 40)
    handed = []

    def alloc(n):
        handed.append(np.zeros(n + 7, dtype=np.uint8))
        return handed[-1]
    raw = ncio.open_dataset(p, decode_times=False, raw_big=True, alloc=alloc)
    assert raw['b'].values.dtype == np.dtype('>f4') and raw['fixed'].values.dtype == np.dtype('>f4')
    assert raw['a'].values.dtype == np.dtype('>i2') and raw['time'].values.dtype.isnative     # 40 bytes: big; time (32) converted
    assert len(handed) == 3 and raw['b'].values.ctypes.data == handed[[i for i, h in enumerate(handed) if h.nbytes == 247][0]].ctypes.data
    np.testing.assert_array_equal(raw['b'].values, mine['b'].values)
    ncio.to_netcdf(raw, str(tmp_path / 'raw_out.nc'))
    ncio.to_netcdf(mine, str(tmp_path / 'native_out.nc'))
    assert open(tmp_path / 'raw_out.nc', 'rb').read() == open(tmp_path / 'native_out.nc', 'rb').read()
    with pytest.raises(IOError):
        open(tmp_path / 'trunc.nc', 'wb').write(open(p, 'rb').read()[:40])
        ncio.open_dataset(str(tmp_path / 'trunc.nc'))


def test_record_reader_reads_one_record_at_a_time(tmp_path):
    """ncio.RecordReader (delta files whose records do not all fit on the device): a variable's records one by one -
    record variables (unlimited time, several of them: padded records) and fixed-size variables - equal to the whole-file
    reader, `_FillValue` and packing decoded, the coordinates decoded once."""
    from scipy.io import netcdf_file
    from pgw4era5_amd import ncio
    rng = np.random.default_rng(8)
    for unlimited in (True, False):
        p = str(tmp_path / ('r%d.nc' % unlimited))
        nc = netcdf_file(p, 'w', version=2)
        nc.createDimension('time', None if unlimited else 5); nc.createDimension('plev', 3)
        nc.createDimension('lat', 4); nc.createDimension('lon', 7)
        t = nc.createVariable('time', 'd', ('time',)); t.units = 'days since 1850-01-01 00:00:00'
        pl = nc.createVariable('plev', 'd', ('plev',)); pl[:] = [1e5, 5e4, 1e4]
        ta = nc.createVariable('ta', 'f', ('time', 'plev', 'lat', 'lon')); ta._FillValue = np.float32(1e20)
        pk = nc.createVariable('tas', 'h', ('time', 'lat', 'lon')); pk.scale_factor = 0.01; pk.add_offset = 2.0
        pk.missing_value = np.int16(-32767)
        t[:] = 52960.5 + np.arange(5) * 30.0
        a = rng.normal(size=(5, 3, 4, 7)).astype('f4'); a[2, 1, 0, 3] = 1e20
        b = rng.integers(-3000, 3000, size=(5, 4, 7)).astype('i2'); b[4, 1, 1] = -32767
        ta[:] = a; pk[:] = b
        nc.close()
        whole = ncio.open_dataset(p)
        for name in ('ta', 'tas'):
            r = ncio.RecordReader(p, name)
            assert r.nrec == 5 and r.dims == whole[name].dims and r.rec_shape == whole[name].shape[1:]
            assert r.dtype == whole[name].values.dtype
            np.testing.assert_array_equal(r.coords['time'], whole['time'].values)
            assert r.coords['time'].dtype.kind == 'M'
            for i in (3, 0, 4, 2, 1):
                np.testing.assert_array_equal(r.read_record(i), whole[name].values[i])
            assert np.isnan(ncio.RecordReader(p, 'ta').read_record(2)[1, 0, 3])
            with pytest.raises(IndexError):
                r.read_record(5)
            r.close()
    with pytest.raises(KeyError):
        ncio.RecordReader(p, 'nope')


@pytest.mark.parametrize('unlimited', [True, False])
def test_band_wise_netcdf_io_is_byte_identical_to_the_whole_file_writer(tmp_path, unlimited):
    """ncio.read_band / BandedWriter (one ERA5 file over several ranks in latitude bands): a rank's rows of a (time, level,
    lat, lon) or (time, lat, lon) variable are one byte range per plane.  Three 'ranks' read their bands of every field
    from a file and write them into a shared output file whose header and remaining variables rank 0 wrote: byte for byte
    the file the whole-array writer produces - with an unlimited time dimension (record variables, padded records) and
    with a fixed one; a field widened to float64 on the way (reference-dtype mode writes float64 T of a float32 file)."""
    from pgw4era5_amd import ncio
    from pgw4era5_amd.parallel import band_rows
    rng = np.random.default_rng(4)
    nlat, nlon, nlev = 7, 5, 3
    ds = ncio.Dataset(attrs=dict(title='bands'), record_dim='time' if unlimited else None)
    F = ncio.Field
    ds['time'] = F(np.array([12.5, 13.5]), ('time',), attrs=dict(units='hours since 2000-01-01 00:00:00'))
    ds['lat'] = F(np.linspace(-3, 3, nlat), ('lat',)); ds['lon'] = F(np.arange(nlon) * 1.0, ('lon',))
    ds['ak'] = F(np.arange(nlev + 1) * 1.0, ('level1',))
    ds['T'] = F(rng.normal(size=(2, nlev, nlat, nlon)).astype('f4'), ('time', 'level', 'lat', 'lon'), attrs=dict(units='K'))
    ds['PS'] = F(rng.normal(size=(2, nlat, nlon)).astype('f4'), ('time', 'lat', 'lon'))
    ds['odd'] = F(rng.integers(-9, 9, size=(2, 3)).astype('i2'), ('time', 'three'))          # 6 bytes per record: padding
    ds['FIS'] = F(rng.normal(size=(2, nlat, nlon)).astype('f4'), ('time', 'lat', 'lon'))
    src = str(tmp_path / 'in.nc')
    ncio.to_netcdf(ds, src)
    # what the run produces: T widened to float64 and changed, PS changed, FIS passed through
    newT = ds['T'].values.astype('f8') * 1.5 + 0.25
    newPS = ds['PS'].values + np.float32(1.0)
    whole = ncio.open_dataset(src, decode_times=False)
    whole['T'] = F(newT, whole['T'].dims, whole['T'].coords, whole['T'].attrs)
    whole['PS'] = F(newPS, whole['PS'].dims, whole['PS'].coords, whole['PS'].attrs)
    want = str(tmp_path / 'whole.nc')
    ncio.to_netcdf(whole, want)
    # band-wise
    banded = ('T', 'PS')
    tmpl = ncio.open_dataset(src, decode_times=False, skip=banded)
    assert ncio.is_placeholder(tmpl['T'].values) and tmpl['T'].shape == (2, nlev, nlat, nlon)
    tmpl['T'] = F(ncio.placeholder(tmpl['T'].shape, 'f8'), tmpl['T'].dims, tmpl['T'].coords, tmpl['T'].attrs)
    out = str(tmp_path / 'banded.nc')
    w = ncio.BandedWriter(tmpl, out, banded)
    w.create()
    for rank in (2, 0, 1):                                        # any order once the file exists
        j0, j1 = band_rows(nlat, rank, 3)
        t = ncio.read_band(src, 'T', j0, j1)
        ps = ncio.read_band(src, 'PS', j0, j1)
        np.testing.assert_array_equal(t, ds['T'].values[:, :, j0:j1]); np.testing.assert_array_equal(ps, ds['PS'].values[:, j0:j1])
        ncio.BandedWriter(tmpl, out, banded).write_band('T', j0, j1, t.astype('f8') * 1.5 + 0.25)
        w.write_band('PS', j0, j1, ps + np.float32(1.0))
    assert open(out, 'rb').read() == open(want, 'rb').read()
    with pytest.raises(ValueError):
        w.write_band('PS', 0, 2, np.zeros((2, 3, nlon), 'f4'))


def test_native_reader_cdf5(tmp_path):
    """CDF-5 (64-bit data) header layout: counts, dimension ids and sizes are 64-bit, extra integer types.  No
    writer for this format exists in the build environment, so the file is assembled by hand from the format
    description (name = count + padded bytes; dimension list 0x0A; attribute list 0x0C; variable list 0x0B)."""
    import struct
    from pgw4era5_amd import ncio

    def name(b):
        return struct.pack('>q', len(b)) + b + b'\x00' * (-len(b) % 4)

    x = (np.arange(5) * 1.5).astype('>f8')                                     # arithmetic returns native order: convert last
    big = (np.arange(12, dtype=np.uint64) + 2 ** 40).reshape(3, 4).astype('>u8')
    recs = np.arange(2 * 4, dtype='>i2').reshape(2, 4)                         # record variable, 8 bytes per record
    dims = [(b't', 0), (b'x', 5), (b'y', 4), (b'z', 3)]
    variables = [(b'x', [1], 6, x.nbytes), (b'big', [3, 2], 11, big.nbytes), (b'r', [0, 2], 3, 8)]   # name, dim ids, nc_type, vsize

    def header(begins):
        h = b'CDF\x05' + struct.pack('>q', 2)                                  # numrecs = 2
        h += struct.pack('>iq', 0x0A, len(dims)) + b''.join(name(n) + struct.pack('>q', ln) for n, ln in dims)
        h += struct.pack('>iq', 0x0C, 1) + name(b'note') + struct.pack('>iq', 2, 3) + b'abc\x00'
        h += struct.pack('>iq', 0x0B, len(variables))
        for (nm, ids, typ, vs), bg in zip(variables, begins):
            h += name(nm) + struct.pack('>q', len(ids)) + b''.join(struct.pack('>q', d) for d in ids)
            h += struct.pack('>iq', 0, 0)                                      # no attributes
            h += struct.pack('>iqq', typ, vs, bg)
        return h

    n0 = len(header([0, 0, 0]))
    begins = [n0, n0 + x.nbytes, n0 + x.nbytes + big.nbytes]
    p = tmp_path / 'c5.nc'
    p.write_bytes(header(begins) + x.tobytes() + big.tobytes() + recs.tobytes())
    ds = ncio.open_dataset(str(p), decode_times=False)
    assert ds.attrs['note'] == 'abc'
    np.testing.assert_array_equal(ds['x'].values, x.astype('f8'))
    assert ds['big'].values.dtype == np.uint64 and ds['big'].dims == ('z', 'y')
    np.testing.assert_array_equal(ds['big'].values, big.astype('u8'))
    assert ds['r'].dims == ('t', 'y') and ds['r'].shape == (2, 4)
    np.testing.assert_array_equal(ds['r'].values, recs.astype('i2'))


def test_writer_variants_produce_the_same_data(tmp_path, monkeypatch):
    """Native writer with 1 / 8 pwrite threads and the scipy writer of the first version (PGW_NC_WRITER=scipy):
    files read back to identical datasets; the native variants are byte-identical."""
    from pgw4era5_amd import ncio
    rng = np.random.default_rng(4)
    monkeypatch.setattr(ncio, 'BIG_VARIABLE', 4096)                      # several 64 MiB-style pieces are not needed: any split works
    ds = ncio.Dataset(attrs=dict(title='w'))
    ds['lat'] = ncio.Field(np.linspace(-5, 5, 11), ('lat',)); ds['lon'] = ncio.Field(np.arange(13.0), ('lon',))
    ds['lev'] = ncio.Field(np.arange(7, dtype=np.int32), ('lev',))
    ds['a'] = ncio.Field(rng.normal(size=(7, 11, 13)).astype(np.float32), ('lev', 'lat', 'lon'), attrs=dict(units='K'))
    ds['b'] = ncio.Field(rng.normal(size=(7, 11, 13)), ('lev', 'lat', 'lon'))
    ds['c'] = ncio.Field(rng.integers(0, 9, size=(11, 13)).astype(np.int16), ('lat', 'lon'))
    ncio.to_netcdf(ds, str(tmp_path / 't8.nc'))
    monkeypatch.setenv('PGW_NC_WRITE_THREADS', '1')
    ncio.to_netcdf(ds, str(tmp_path / 't1.nc'))
    monkeypatch.setenv('PGW_NC_WRITER', 'scipy')
    ncio.to_netcdf(ds, str(tmp_path / 'sp.nc'))
    assert open(tmp_path / 't8.nc', 'rb').read() == open(tmp_path / 't1.nc', 'rb').read()
    a, b = ncio.open_dataset(str(tmp_path / 't8.nc')), ncio.open_dataset(str(tmp_path / 'sp.nc'))
    assert set(a.variables) == set(b.variables)
    for k in a.variables:
        assert a[k].dims == b[k].dims and a[k].values.dtype == b[k].values.dtype
        np.testing.assert_array_equal(a[k].values, b[k].values, err_msg=k)
    assert a['a'].attrs['units'] == b['a'].attrs['units'] == 'K'


def test_ncio_random_files_against_scipy():
    """tools/fuzz_ncio.py: random classic / 64-bit-offset files (record and fixed variables of every classic type incl.
    NC_CHAR, scalars, attributes) written by scipy -> native reader -> native writer -> scipy: nothing is lost."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('fuzz_ncio', os.path.join(ROOT, 'tools', 'fuzz_ncio.py'))
    mod = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(mod)
    assert mod.main(seed=4, cases=120) == 0


def test_decode_cf_time_calendars():
    from pgw4era5_amd.ncio import decode_cf_time
    got = decode_cf_time([0, 31, 59.5], 'days since 1850-01-01 00:00:00', 'proleptic_gregorian')
    assert str(got[1]) == '1850-02-01T00:00:00' and str(got[2]) == '1850-03-01T12:00:00'
    # noleap: day 59 of year is always March 1st, also in leap years
    got = decode_cf_time([365 * 146 + 59], 'days since 1850-01-01', 'noleap')
    assert str(got[0]) == '1996-03-01T00:00:00'
    got = decode_cf_time([15.5], 'days since 1995-01-01 00:00:00', '360_day')
    assert str(got[0]) == '1995-01-16T12:00:00'
    got = decode_cf_time([3600 * 5], 'seconds since 1970-01-01 00:00:00')
    assert str(got[0]) == '1970-01-01T05:00:00'
    with pytest.raises(ValueError):
        decode_cf_time([0], 'fortnights since 1850-01-01')


def test_time_bracket_matches_oracle():
    from pgw4era5_amd.step_03_apply_to_era import delta_time_bracket
    times = np.array(['1995-%02d-15T12:00:00' % m for m in range(1, 13)], dtype='datetime64[s]')
    for tgt in ['2006-08-02T03:00:00', '2006-01-03T00:00:00', '2006-12-31T21:00:00', '2006-03-15T12:00:00',
                '2008-02-29T06:00:00']:
        ib, ia, x_hi, x_new, keep = delta_time_bracket(times, np.datetime64(tgt))
        ob, oa, tb, ta, okeep = O.delta_time_bracket(times, np.datetime64(tgt))
        assert (ib, ia) == (ob, oa) and list(keep) == list(okeep)
        if ib != ia:
            ns = 'datetime64[ns]'
            assert x_hi == float((ta.astype(ns) - tb.astype(ns)).astype(np.int64))
            assert x_new == float((np.datetime64(tgt).astype(ns) - tb.astype(ns)).astype(np.int64))
            assert 0 <= x_new <= x_hi
        else:
            assert x_hi == 0.0
    # python datetime target (what the CLI passes)
    ib, ia, *_ = delta_time_bracket(times, dt.datetime(2006, 8, 2, 3))
    assert (ib, ia) == (6, 7)
    daily = np.arange(np.datetime64('1996-01-01'), np.datetime64('1997-01-01')).astype('datetime64[s]')
    ib, ia, x_hi, x_new, keep = delta_time_bracket(daily, np.datetime64('2006-03-01T06:00:00'))
    assert len(keep) == 365 and 59 not in keep
    assert str(daily[keep[ib]])[5:10] == '03-01' and str(daily[keep[ia]])[5:10] == '03-02'


def test_regrid_tables_reproduce_oracle_on_cpu():
    """The index/weight tables + the kernel's formula, evaluated with numpy, equal the oracle."""
    from pgw4era5_amd import synthetic
    from pgw4era5_amd.functions import regrid_tables
    g = synthetic.make_gcm_grid_case(seed=3)
    for tl in (g['targ_lon'], g['targ_lon'] - 180.0):
        tb = regrid_tables(g['src_lat'], g['src_lon'], g['targ_lat'], tl)
        f = g['field']
        pole_s = np.nanmean(f[..., tb['south_row'], :], axis=-1) if tb['south_row'] >= 0 else None
        pole_n = np.nanmean(f[..., tb['north_row'], :], axis=-1) if tb['north_row'] >= 0 else None

        def row(r):
            if r < 0:
                return np.repeat(pole_s[..., None], f.shape[-1], -1)
            if r >= f.shape[-2]:
                return np.repeat(pole_n[..., None], f.shape[-1], -1)
            return f[..., r, :]
        out = np.empty(f.shape[:-2] + (len(g['targ_lat']), len(tl)))
        for j in range(len(g['targ_lat'])):
            lo, hi = row(tb['lat_lo'][j]), row(tb['lat_hi'][j])
            a = (hi - lo) / tb['lat_Dx'][j] * tb['lat_dx'][j] + lo
            ya, yb = a[..., tb['lon_lo']], a[..., tb['lon_hi']]
            out[..., j, :] = (yb - ya) / tb['lon_Dx'] * tb['lon_dx'] + ya
        want = O.regrid_lat_lon(f, g['src_lat'], g['src_lon'], g['targ_lat'], tl)
        np.testing.assert_allclose(out, want, rtol=1e-13, atol=1e-15)
    with pytest.raises(ValueError) as e:
        regrid_tables(g['src_lat'][::-1], g['src_lon'], g['targ_lat'], g['targ_lon'])
    assert 'North or South' in str(e.value)
    with pytest.raises(ValueError) as e:
        regrid_tables(g['src_lat'], g['src_lon'][:40], g['targ_lat'], g['targ_lon'])
    assert 'East or West' in str(e.value)


def test_synthetic_files_roundtrip(tmp_path):
    from pgw4era5_amd import synthetic, ncio
    case = synthetic.make_case(4, 6, 8, seed=2, dtype=np.float32)
    path = synthetic.write_case_files(case, str(tmp_path / 'era'), str(tmp_path / 'deltas'))
    assert os.path.basename(path) == 'cas20060802030000.nc'
    ds = ncio.open_dataset(path, decode_times=False)
    np.testing.assert_array_equal(ds['T'].values, case['era']['T'])
    assert ds['T'].dims == ('time', 'level', 'lat', 'lon') and ds['T'].dtype == np.float32
    dd = ncio.open_dataset(str(tmp_path / 'deltas' / 'ta_delta.nc'))
    np.testing.assert_array_equal(dd['time'].values, case['delta_times'])
    np.testing.assert_array_equal(dd['plev'].values, case['plev'])
    assert os.path.exists(tmp_path / 'deltas' / 'ps_historical.nc')


def test_delta_files_are_mask_and_scale_decoded(tmp_path):
    """the reference reads the delta files with a plain xr.open_dataset (functions.py:203): _FillValue /
    missing_value -> NaN, scale_factor / add_offset applied; the ERA5 file with decode_cf=False (step_03:60): raw."""
    from pgw4era5_amd import ncio
    F = ncio.Field
    tos = np.array([[1.5, 1e20], [2.5, 3.5]], dtype=np.float32)[None]
    ta = np.array([[[250.0, -999.0], [1e20, 260.0]]], dtype=np.float64)[None]
    packed = np.array([[100, -32768], [200, 300]], dtype=np.int16)[None]
    ds = ncio.Dataset()
    ds['time'] = F(np.array([0.0]), ('time',), attrs=dict(units='days since 2000-01-01'))
    ds['tos'] = F(tos, ('time', 'lat', 'lon'), attrs={'_FillValue': np.float32(1e20), 'units': 'K'})
    ds['ta'] = F(ta, ('time', 'plev', 'lat', 'lon'), attrs={'_FillValue': 1e20, 'missing_value': -999.0})
    ds['pk'] = F(packed, ('time', 'lat', 'lon'), attrs={'_FillValue': np.int16(-32768), 'scale_factor': 0.01, 'add_offset': 273.15})
    path = str(tmp_path / 'd.nc')
    ncio.to_netcdf(ds, path)
    for reader in ('native', 'scipy'):
        os.environ.pop('PGW_NC_READER', None)
        if reader == 'scipy':
            os.environ['PGW_NC_READER'] = 'scipy'
        try:
            got = ncio.open_dataset(path)                                  # like xr.open_dataset
            raw = ncio.open_dataset(path, decode_times=False)              # like decode_cf=False
        finally:
            os.environ.pop('PGW_NC_READER', None)
        assert got['tos'].dtype == np.float32 and np.isnan(got['tos'].values[0, 0, 1]) and got['tos'].values[0, 1, 1] == 3.5
        assert '_FillValue' not in got['tos'].attrs and got['tos'].attrs['units'] == 'K'
        assert np.isnan(got['ta'].values[0, 0, 0, 1]) and np.isnan(got['ta'].values[0, 0, 1, 0]) and got['ta'].values[0, 0, 0, 0] == 250.0
        pk = got['pk'].values
        assert pk.dtype == np.float64 and np.isnan(pk[0, 0, 1])            # int16 with add_offset decodes to float64
        np.testing.assert_allclose(pk[0, 1], [200 * 0.01 + 273.15, 300 * 0.01 + 273.15], rtol=1e-15)
        assert raw['tos'].values[0, 0, 1] == np.float32(1e20) and raw['pk'].dtype == np.int16 and raw['time'].values[0] == 0.0
        assert raw['tos'].attrs['_FillValue'] == np.float32(1e20)


def test_writer_keeps_the_record_dimension_and_packs_vsize_unsigned(tmp_path):
    """(1) the unlimited `time` of the input file survives a read -> modify -> write cycle (the reference's
    to_netcdf keeps it); records of several variables interleave as the classic format prescribes (scipy reads them
    back); (2) vsize is an unsigned field: a 2 ... 4 GiB variable must not overflow the header packer."""
    import struct
    from scipy.io import netcdf_file
    from pgw4era5_amd import ncio
    F = ncio.Field
    rng = np.random.default_rng(0)
    a = rng.normal(size=(3, 2, 5)).astype(np.float32)
    b = rng.normal(size=(3, 5))
    ds = ncio.Dataset(record_dim='time')
    ds['time'] = F(np.arange(3.0), ('time',))
    ds['lon'] = F(np.arange(5.0), ('lon',))
    ds['a'] = F(a, ('time', 'lev', 'lon'))
    ds['fixed'] = F(np.arange(5, dtype=np.int16), ('lon',))            # odd byte count: exercises the padding
    ds['b'] = F(b, ('time', 'lon'))
    path = str(tmp_path / 'rec.nc')
    ncio.to_netcdf(ds, path)
    nc = netcdf_file(path, 'r', mmap=False)                            # independent reader
    assert nc.dimensions['time'] is None and nc.variables['a'].isrec and nc.variables['b'].isrec and not nc.variables['fixed'].isrec
    np.testing.assert_array_equal(nc.variables['a'][:], a)
    np.testing.assert_array_equal(nc.variables['b'][:], b)
    np.testing.assert_array_equal(nc.variables['fixed'][:], np.arange(5))
    nc.close()
    back = ncio.open_dataset(path, decode_times=False)
    assert back.record_dim == 'time'
    np.testing.assert_array_equal(back['a'].values, a)
    np.testing.assert_array_equal(back['b'].values, b)
    back['a'] = F(a * 2, ('time', 'lev', 'lon'))                       # what the step_03 driver does with T, QV, ...
    ncio.to_netcdf(back, str(tmp_path / 'rec2.nc'))
    again = ncio.open_dataset(str(tmp_path / 'rec2.nc'), decode_times=False)
    assert again.record_dim == 'time'
    np.testing.assert_array_equal(again['a'].values, a * 2)
    # one record variable alone is not padded
    one = ncio.Dataset(record_dim='t')
    one['v'] = F(np.arange(6, dtype=np.int8).reshape(3, 2), ('t', 'x'))
    ncio.to_netcdf(one, str(tmp_path / 'one.nc'))
    nc = netcdf_file(str(tmp_path / 'one.nc'), 'r', mmap=False)
    np.testing.assert_array_equal(nc.variables['v'][:], np.arange(6).reshape(3, 2))
    nc.close()
    # header packing of large variables (no data written): 3 GiB fits the unsigned field, 5 GiB is written as 2^32 - 1
    for vsize, want in ((3 << 30, 3 << 30), (5 << 30, 0xFFFFFFFF), ((1 << 32) - 4, (1 << 32) - 4)):
        spec = dict(name='big', dims=('x',), attrs={}, key='f8', vsize=vsize)
        h = ncio._nc_header({}, {'x': vsize // 8}, None, 0, [spec], [1024], {'x': 0})
        assert struct.unpack('>I', h[-12:-8])[0] == want and struct.unpack('>q', h[-8:])[0] == 1024


def _task(x, k):
    return (x * k, int(os.environ.get('RANK', os.environ.get('PGW_RANK', '0'))))


def test_itermp_serial_and_sharding():
    from pgw4era5_amd.parallel import IterMP, shard_indices
    assert shard_indices(7, 1, 3) == [1, 4]
    assert sorted(sum((shard_indices(10, r, 4) for r in range(4)), [])) == list(range(10))
    imp = IterMP(njobs=1)
    imp.run(_task, dict(k=3), [dict(x=i) for i in range(5)])
    assert [o[0] for o in imp.output] == [0, 3, 6, 9, 12]


def test_itermp_spawned_workers():
    from pgw4era5_amd.parallel import IterMP
    imp = IterMP(njobs=2)
    imp.run(_task, dict(k=2), [dict(x=i) for i in range(5)])
    assert [o[0] for o in imp.output] == [0, 2, 4, 6, 8]
    assert [o[1] for o in imp.output] == [0, 1, 0, 1, 0]          # round-robin deal


GLOO_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, %r)
from pgw4era5_amd.parallel import IterMP
def task(x, k):
    return [x * k, int(os.environ['RANK']), os.environ.get('PGW_CARD_SHARE')]
imp = IterMP(njobs=2, backend='gloo')
imp.run(task, dict(k=5), [dict(x=i) for i in range(7)])
if int(os.environ['RANK']) == 0:
    print('RESULT ' + json.dumps(imp.output))
'''


def test_itermp_world_size_2_gloo(tmp_path):
    script = tmp_path / 'w.py'
    script.write_text(GLOO_SCRIPT % ROOT)
    env = dict(os.environ, MASTER_ADDR='127.0.0.1')
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29517', str(script)],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = [l for l in r.stdout.splitlines() if l.startswith('RESULT ')][0]
    import json
    out = json.loads(line[7:])
    assert [o[0] for o in out] == [0, 5, 10, 15, 20, 25, 30]
    assert [o[1] for o in out] == [0, 1, 0, 1, 0, 1, 0]
    # parallel.card_share ran on every rank before its shard (no device here: nobody shares a card)
    assert [o[2] for o in out] == ['1'] * 7


def _dying_task(x):
    if x == 1:
        os._exit(7)                      # dies like a GPU fault / abort inside the native library: no exception, no message
    return x


def test_spawned_worker_that_dies_silently_fails_the_run(monkeypatch):
    """a worker that exits without posting must make `-p N` fail, not hang."""
    from pgw4era5_amd.parallel import IterMP
    monkeypatch.setenv('PGW_WORKER_GRACE_S', '0.5')
    imp = IterMP(njobs=2)
    with pytest.raises(RuntimeError) as e:
        imp.run(_dying_task, {}, [dict(x=i) for i in range(4)])
    assert 'exited with code 7' in str(e.value)


def test_bench_gpus_flag_starts_that_many_ranks():
    """`python bench.py --gpus 2` with no WORLD_SIZE starts two ranks itself (child torch.distributed.run, gloo here,
    no GPU work: --dry-run) and relays their line; a WORLD_SIZE that contradicts --gpus is an error."""
    import json
    env = dict(os.environ, PGW_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None); env.pop('RANK', None); env.pop('LOCAL_RANK', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'],
                       capture_output=True, text=True, timeout=600, env=env)
    assert r.returncode == 0, r.stderr[-2000:]
    line = json.loads([l for l in r.stdout.splitlines() if l.startswith('{')][-1])
    assert line['n_gpus'] == 2 and line['collective']['ranks_counted_by_all_reduce'] == 2
    assert line['collective']['backend'] == 'gloo' and line['dry_run'] is True
    # what an N-rank line reports beside the HBM-resident value: every rank's host placement (disjoint CPU sets here: two
    # ranks, no GPU, the allowed CPUs dealt evenly), the per-rank PCIe-inclusive float32 leg, the all-ranks command line
    pr = line['per_rank']
    assert set(pr) >= {'affinity', 'affinity_disjoint', 'pcie_inclusive_f32_reference', 'end_to_end_cli_all_ranks'}
    assert len(pr['affinity']) == 2 and [x['local_rank'] for x in pr['affinity']] == [0, 1]
    if len(os.sched_getaffinity(0)) >= 2:
        assert all(x['bound'] for x in pr['affinity']) and pr['affinity_disjoint'] is True
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--dry-run'],
                       capture_output=True, text=True, timeout=120, env=dict(env, WORLD_SIZE='1', RANK='0'))
    assert r.returncode == 2 and 'WORLD_SIZE is 1' in r.stderr


def test_rank_cpu_sets_follow_the_gpus_numa_nodes_and_are_disjoint():
    """parallel.rank_cpu_set: a rank's CPUs are the allowed CPUs of its GPU's NUMA node, dealt disjointly to the ranks
    that share the node; unknown nodes or a container whose CPUs miss a node fall back to an even deal of the allowed
    CPUs; the kernel's cpulist format round-trips."""
    from pgw4era5_amd import parallel as P
    nodes = {0: set(range(0, 64)) | set(range(128, 192)), 1: set(range(64, 128)) | set(range(192, 256))}
    cpus_of = lambda n: nodes.get(n, set())
    # an 8-GPU two-socket host: GPUs 0-3 on node 0, 4-7 on node 1
    sets = [P.rank_cpu_set(r, 8, set(range(256)), [0, 0, 0, 0, 1, 1, 1, 1], cpus_of) for r in range(8)]
    assert all(len(s) == 32 for s in sets)
    assert all(not (sets[i] & sets[j]) for i in range(8) for j in range(i))
    assert all(sets[r] <= nodes[0] for r in range(4)) and all(sets[r] <= nodes[1] for r in range(4, 8))
    assert P._cpulist(sets[2]) == '128-159' and P.parse_cpulist('128-159') == sets[2]
    assert P.parse_cpulist('0-3,8,10-11\n') == {0, 1, 2, 3, 8, 10, 11}
    # no NUMA information (this container): even, disjoint deal of the allowed CPUs
    a, b = (P.rank_cpu_set(r, 2, set(range(16)), [-1, -1], cpus_of) for r in range(2))
    assert a == set(range(8)) and b == set(range(8, 16))
    # a container that owns CPUs of one node only while the GPUs sit on both: still disjoint
    a, b = (P.rank_cpu_set(r, 2, set(range(100, 116)), [0, 1], cpus_of) for r in range(2))
    assert not (a & b) and (a | b) == set(range(100, 116))
    # more ranks than CPUs: shared, never empty
    assert all(P.rank_cpu_set(r, 4, {0, 1, 2}, [0, 0, 0, 0], cpus_of) == {0, 1, 2} for r in range(4))
    # bind: switched off by the environment, otherwise within the allowed set
    before = os.sched_getaffinity(0)
    try:
        os.environ['PGW_NUMA_BIND'] = '0'
        assert P.bind_rank_to_numa(0, 2)['bound'] is False and os.sched_getaffinity(0) == before
        del os.environ['PGW_NUMA_BIND']
        info = P.bind_rank_to_numa(1, 2)
        assert info['bound'] and os.sched_getaffinity(0) <= before and P.parse_cpulist(info['cpus']) == os.sched_getaffinity(0)
    finally:
        os.environ.pop('PGW_NUMA_BIND', None)
        os.sched_setaffinity(0, before)


def test_bench_cpu_baseline_legs():
    """The two legs of bench.py's cpu_baseline (one process, file-parallel processes) on a toy file."""
    import argparse
    sys.path.insert(0, ROOT)
    import bench
    from pgw4era5_amd import synthetic
    case = synthetic.make_case(nlat=12, nlon=16, nlev=20, seed=3)
    a = argparse.Namespace(cpu_rows=6, nlat=12, nlon=16, cpu_procs=2)
    res = bench.cpu_baseline(case, a, np)
    assert res['cores'] == 1 and res['kind'] == 'port' and res['value'] > 0
    legs = res['one_process']
    assert set(legs) == {'c_column_loops', 'numpy_vectorised'}
    assert legs['c_column_loops']['iterations'] == legs['numpy_vectorised']['iterations']
    assert res['value'] == max(l['files_per_hour'] for l in legs.values())
    assert res['parallel']['cores'] == 2 and res['parallel']['value'] > 0


def test_cli_argument_surface():
    from pgw4era5_amd import step_03_apply_to_era as s3, step_02_preproc_deltas as s2
    with pytest.raises(ValueError) as e:
        s3._cli(['-o', 'x', '-d', 'y'])
    assert str(e.value) == 'Input directory (-i) is required.'
    with pytest.raises(ValueError) as e:
        s3._cli(['-i', 'x', '-o', 'y', '-d', 'z', '-D', 'bogus'])
    assert 'Invalid input for argument --debug_mode' in str(e.value)
    with pytest.raises(ValueError) as e:
        s2.main(['regridding', '-i', 'a', '-o', 'b'])
    assert str(e.value) == 'era5_file_path is required for regridding step.'
    with pytest.raises(SystemExit):
        s2.main(['frobnicate'])


def test_pipelined_shard_overlaps_stages_and_keeps_order():
    """IterMP runs func.stages = (load, compute, store) as a 3-stage pipeline per rank."""
    import threading
    import time
    from pgw4era5_amd.parallel import run_shard, IterMP
    log = []
    lock = threading.Lock()

    def load(x):
        time.sleep(0.05)
        with lock:
            log.append(('load', x))
        return dict(x=x)

    def compute(item):
        with lock:
            log.append(('compute', item['x']))
        time.sleep(0.02)
        item['y'] = item['x'] * 10
        return item

    def store(item):
        time.sleep(0.05)
        with lock:
            log.append(('store', item['x']))
        return item['y']

    def func(x):
        return store(compute(load(x)))
    func.stages = (load, compute, store)
    tasks = [dict(x=i) for i in range(6)]
    t0 = time.time()
    out = run_shard(func, tasks, list(range(6)))
    el = time.time() - t0
    assert out == [(i, i * 10) for i in range(6)]
    assert el < 6 * 0.12 * 0.85                      # faster than the serial 6 x (0.05 + 0.02 + 0.05)
    assert any(log.index(('load', i + 1)) < log.index(('store', i)) for i in range(5))   # reader runs ahead of the writer
    imp = IterMP(njobs=1)
    imp.run(func, {}, tasks)
    assert imp.output == [i * 10 for i in range(6)]

    def bad_compute(item):
        if item['x'] == 3:
            raise ValueError('boom')
        return compute(item)
    func.stages = (load, bad_compute, store)
    with pytest.raises(ValueError):
        run_shard(func, tasks, list(range(6)))


BAND_HOOK_SCRIPT = r'''
import os, sys, json
sys.path.insert(0, %r)
import numpy as np
import torch.distributed as dist
from pgw4era5_amd.parallel import band_max_hook, band_rows
dist.init_process_group('gloo')
r = dist.get_rank()
hook = band_max_hook()
v = np.array([0.0, 0.0, 1.0, 0.3 + r, 0.0, float(r), -np.inf if r == 0 else 0.125, 13.0 * r])
hook(v)                                        # in place
with open(os.path.join(sys.argv[1], 'hook%%d.json' %% r), 'w') as f:
    json.dump(v.tolist(), f)
dist.barrier()
dist.destroy_process_group()
'''


def test_band_rows_and_band_max_hook_gloo(tmp_path):
    """Host side of the latency mode (one file in latitude bands): the partition, and the exchange step - an element-wise
    MAX all-reduce over two gloo ranks, in place on the array the C library hands to the hook."""
    from pgw4era5_amd.parallel import band_rows
    for nlat, world in ((721, 8), (21, 2), (5, 5), (7, 3)):
        rows = [band_rows(nlat, r, world) for r in range(world)]
        assert rows[0][0] == 0 and rows[-1][1] == nlat
        assert all(rows[i][1] == rows[i + 1][0] for i in range(world - 1))
        sizes = [b - a for a, b in rows]
        assert max(sizes) - min(sizes) <= 1 and min(sizes) >= 1
    script = tmp_path / 'h.py'
    script.write_text(BAND_HOOK_SCRIPT % ROOT)
    r = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2',
                        '--master-addr', '127.0.0.1', '--master-port', '29523', str(script), str(tmp_path)],
                       capture_output=True, text=True, timeout=600, env=dict(os.environ, MASTER_ADDR='127.0.0.1'))
    assert r.returncode == 0, r.stderr[-2000:]
    import json
    lines = {k: json.load(open(str(tmp_path / ('hook%d.json' % k)))) for k in (0, 1)}
    assert lines[0] == lines[1] == [0.0, 0.0, 1.0, 1.3, 0.0, 1.0, 0.125, 13.0]


def test_five_stage_pipeline_with_bounded_buffer_sets_and_abort():
    """The file driver's shape (step_03_apply_to_era.py stages: load, upload, compute, download, store): middle stages have one
    thread each and hand a bounded number of buffer sets round; results keep task order; a failing middle stage sets
    `func.abort`, stages waiting for a buffer set give up and the first failure is re-raised - nothing hangs."""
    import queue
    import threading
    import time
    from pgw4era5_amd.parallel import run_shard
    abort = threading.Event()
    sets_in, sets_out = queue.Queue(), queue.Queue()
    for i in range(2):
        sets_in.put(i); sets_out.put(i)
    in_use = {'in': 0, 'out': 0, 'max_in': 0, 'max_out': 0}
    lock = threading.Lock()
    fail_at = {'x': None}

    def take(q):
        while True:
            try:
                return q.get(timeout=0.05)
            except queue.Empty:
                if abort.is_set():
                    raise RuntimeError('aborted')

    def load(x):
        time.sleep(0.01)
        return dict(x=x)

    def upload(item):
        item['in'] = take(sets_in)
        with lock:
            in_use['in'] += 1; in_use['max_in'] = max(in_use['max_in'], in_use['in'])
        time.sleep(0.02)
        return item

    def compute(item):
        try:
            if item['x'] == fail_at['x']:
                raise ValueError('boom %d' % item['x'])
            item['out'] = take(sets_out)
            with lock:
                in_use['out'] += 1; in_use['max_out'] = max(in_use['max_out'], in_use['out'])
            time.sleep(0.01)
            item['y'] = item['x'] ** 2
        finally:
            with lock:
                in_use['in'] -= 1
            sets_in.put(item.pop('in'))
        return item

    def download(item):
        time.sleep(0.02)
        with lock:
            in_use['out'] -= 1
        sets_out.put(item.pop('out'))
        return item

    def store(item):
        time.sleep(0.01)
        return item['y']

    def func(x):
        return store(download(compute(upload(load(x)))))
    func.stages = (load, upload, compute, download, store)
    func.abort = abort
    tasks = [dict(x=i) for i in range(12)]
    t0 = time.time()
    out = run_shard(func, tasks, list(range(12)))
    el = time.time() - t0
    assert out == [(i, i * i) for i in range(12)]
    assert in_use['max_in'] <= 2 and in_use['max_out'] <= 2 and in_use['max_in'] == 2       # both sets were in flight
    assert el < 12 * 0.07 * 0.8                                                             # stages overlapped
    fail_at['x'] = 5
    t0 = time.time()
    with pytest.raises(ValueError) as e:
        run_shard(func, tasks, list(range(12)))
    assert 'boom 5' in str(e.value) and abort.is_set()
    assert time.time() - t0 < 10
    # A chain cancelled between two stages keeps what its finished stages took (file 6 or 7 may hold an input set whose
    # compute stage never ran).  With `func.reset` (step_03_apply_to_era.reset_after_abort for the real driver) run_shard
    # makes everything available again once all stage threads have stopped, and a second run in the same process works.
    lost = 2 - sets_in.qsize()

    def reset():
        for q in (sets_in, sets_out):
            while not q.empty():
                q.get_nowait()
            for i in range(2):
                q.put(i)
        in_use.update({'in': 0, 'out': 0})
        abort.clear()
    func.reset = reset
    fail_at['x'] = 5
    with pytest.raises(ValueError):
        run_shard(func, tasks, list(range(12)))
    assert sets_in.qsize() == 2 and sets_out.qsize() == 2 and not abort.is_set()
    fail_at['x'] = None
    assert run_shard(func, tasks, list(range(12))) == [(i, i * i) for i in range(12)]
    assert lost >= 0


def test_driver_buffer_sets_and_pinned_pool_are_whole_again_after_an_abort():
    """step_03_apply_to_era.reset_after_abort: every device buffer set is back in its queue and every pinned buffer the
    pool handed out is free again; releasing a buffer to a pool that does not own it is an error, not a silent leak."""
    from pgw4era5_amd import step_03_apply_to_era as s3
    from pgw4era5_amd.device import PinnedPool

    class FakeCtx:                                     # PinnedPool only calls pgw_host_alloc / pgw_host_free through these
        handle = None

        class lib:
            @staticmethod
            def pgw_host_alloc(h, size, pp):
                import ctypes as C
                buf = (C.c_ubyte * size)()
                FakeCtx.keep.append(buf)
                C.cast(pp, C.POINTER(C.c_void_p))[0] = C.addressof(buf)
                return 0

            @staticmethod
            def pgw_host_free(h, addr):
                return 0
        keep = []

        @staticmethod
        def _check(rc):
            assert rc == 0
    pool = PinnedPool(FakeCtx)
    a, b = pool.acquire(100), pool.acquire(3 << 20)
    assert pool.release(a) and pool.acquire(100).ctypes.data == a.ctypes.data        # recycled by size
    other = PinnedPool(FakeCtx)
    with pytest.raises(RuntimeError):
        s3._release_pinned(other, b)
    sets = s3._BufferSets(2)
    x = sets.inp.get(); sets.out.get(); sets.out.get()
    x['T'] = 'device array'
    key = ('test-shape', '<f4')
    s3._DEVICE_BUFFERS[key] = sets
    s3._POOLS[-1] = pool
    s3._ABORT.set()
    try:
        s3.reset_after_abort()
        assert sets.inp.qsize() == 2 and sets.out.qsize() == 2 and not s3._ABORT.is_set()
        assert any(s.get('T') == 'device array' for s in (sets.inp.get(), sets.inp.get()))   # the same sets, arrays kept
        assert {pool.acquire(100).ctypes.data, pool.acquire(3 << 20).ctypes.data} == {a.ctypes.data, b.ctypes.data}
    finally:
        del s3._DEVICE_BUFFERS[key]
        del s3._POOLS[-1]


def test_run_starmap_keeps_the_reference_call_forms():
    """parallel.run_starmap / starmap_helper / test_IMP under the reference's names (parallel.py:12-32, 72-77)."""
    from pgw4era5_amd import parallel as P
    serial = P.run_starmap(P.test_IMP, [dict(iter_arg=i, fixed_arg='x') for i in range(4)], njobs=1)
    assert serial == [0, 1, 2, 3]
    wrapped = [(dict(iter_arg=i, fixed_arg='x', func=P.test_IMP),) for i in range(3)]      # the njobs > 1 form of IterMP.run
    assert P.run_starmap(P.test_IMP, wrapped, njobs=1) == [0, 1, 2]
    assert P.starmap_helper(dict(func=P.test_IMP, iter_arg=7, fixed_arg=None)) == 7


def test_smi_sampler_window_summary_without_a_card():
    """tools/smi/sampler.py (bench.py's `device_state`): no helper library or no card -> `available` is False and every call
    is a no-op; the window arithmetic (means, extremes, limiter residency from the accumulated counters) on fed samples."""
    import importlib.util
    spec = importlib.util.spec_from_file_location('smi_sampler', os.path.join(ROOT, 'tools', 'smi', 'sampler.py'))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    s = m.Sampler()
    if not s.available:                              # this container: no GPU
        assert s.summary() is None and s.start() is s and s.stop() is s and s.samples == []
    base = {k: -1.0 for k in m.FIELDS}
    def row(t, clk, lo, hi, w, acc, ppt):
        r = dict(base, gfx_mhz=clk, gfx_mhz_min_xcd=lo, gfx_mhz_max_xcd=hi, power_w=w, acc_counter=acc, ppt_acc=ppt,
                 socket_thm_acc=0.0, t_hotspot_c=50.0 + t)
        return (float(t), r)
    s.samples = [row(0, 2300, 2250, 2350, 400, 1000, 100), row(1, 2200, 2100, 2300, 1300, 2000, 700),
                 row(2, 2100, 2000, 2200, 1390, 3000, 1500), row(3, 2400, 2400, 2400, 300, 4000, 1500)]
    w = s.window(1.0, 2.0)
    assert w['samples'] == 2 and w['gfx_mhz_mean'] == 2150.0 and w['gfx_mhz_min'] == 2000 and w['gfx_mhz_max'] == 2300
    assert w['power_w_mean'] == 1345.0 and w['power_w_max'] == 1390 and w['ppt_residency'] == 0.8
    assert w['socket_thm_residency'] == 0.0 and 'hbm_thm_residency' not in w and w['uclk_mhz'] is None
    assert s.window(None, None)['samples'] == 4 and s.window(10, 11) is None


def test_spread_pool_draw_logic_with_a_simulated_card():
    """device.SpreadPool without a GPU: a stand-in context whose memory is handed out in allocation order and whose two-write
    probe answers 5000 GB/s for two arrays of one 90 GB stretch and 6600 across stretches (what the MI355X does, DESIGN.md
    section 4).  The draw strides through the first stretch with spacers, keeps half of the stock inside and half outside,
    frees everything else, stays within its budget (one class when the budget ends inside the first stretch), and dropped
    arrays come back into the stock."""
    import gc
    from pgw4era5_amd import device as D

    class Lib:
        def __init__(self):
            self.freed = []

        def pgw_free(self, h, p):
            self.freed.append(p)
            return 0

        def pgw_device_count(self, ref):
            ref._obj.value = 1
            return 0

    class Ctx:
        STRETCH = 90e9

        def __init__(self):
            self.handle, self._live, self.lib, self.cursor, self.alive = 1, 0, Lib(), 4096, {}

        def mem_info(self):
            return int(300e9), int(300e9)

        def empty(self, shape, dtype):
            n = int(np.prod(shape)) * np.dtype(dtype).itemsize
            a = D.DeviceArray(self, shape, dtype, ptr=self.cursor)
            a._owner = a
            self.alive[self.cursor] = n
            self.cursor += n
            self._live += n
            return a

        def region(self, ptr):
            return int(ptr // self.STRETCH)

        def placement_probe(self, src, dst, reps=3, rows=None):
            a, b = dst
            return 5000.0 if self.region(a.ptr) == self.region(b.ptr) else 6600.0

    field = 137 * 721 * 1440 * 8
    ctx = Ctx()
    pool = D.SpreadPool(ctx, field, 9)
    assert pool.info['classes'] == 2 and pool.info['kept_per_class'] == [5, 4] and pool.info['drawn_GB'] > 90
    assert pool.info['probe_GBps_inside'] == 5000 and pool.info['probe_GBps_outside'] == 6600
    assert all(ctx.region(a.ptr) == 0 for a in pool.stock[0]) and all(ctx.region(a.ptr) >= 1 for a in pool.stock[1])
    kept = {a.ptr for a in pool.stock[0] + pool.stock[1]}
    assert set(ctx.alive) - set(ctx.lib.freed) == kept              # spacers and surplus candidates all went back
    spacers = [p for p in ctx.lib.freed if ctx.alive[p] == D.SpreadPool.SPACER]
    assert 8 <= len(spacers) <= 12                                    # ~ (90 GB - 5 arrays) / (8 GiB + one array)
    views = [pool.take((1, 137, 721, 1440), np.float64, cls=c) for c in (0, 1, 0, 1)]
    assert [v.placement_class for v in views] == [0, 1, 0, 1] and len(pool.stock[0]) == 3 and len(pool.stock[1]) == 2
    p1 = views[1].ptr
    del views[1]
    gc.collect()
    assert [a.ptr for a in pool.stock[1]][-1] == p1 and len(pool.stock[1]) == 3      # dropped -> back, same class
    half = pool.take((1, 137, 721, 1440), np.float32)                # a float32 field of a float64 pool: a view of a stock array
    assert half.nbytes == field // 2 and half.placement_class in (0, 1)
    pool.close()
    assert pool.stock == [[], []]
    # a budget that ends inside the first stretch: plain stock, one class
    ctx2 = Ctx()
    pool2 = D.SpreadPool(ctx2, field, 9, budget_bytes=int(40e9))
    assert pool2.info['classes'] == 1 and pool2.info['kept'] == 9 and pool2.info['drawn_GB'] <= 40.0 + 9 * field / 1e9
    assert len(pool2.stock[0]) == 9 and pool2.stock[1] == []
    # small arrays are not worth a draw
    pool3 = D.SpreadPool(Ctx(), 4096, 3)
    assert pool3.info['classes'] == 1 and pool3.info['kept'] == 3 and pool3.info['drawn_GB'] == 0.0
