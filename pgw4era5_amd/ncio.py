"""
Minimal labelled arrays + NetCDF-3 file I/O for the step_02 / step_03 drivers.

The reference does its I/O through xarray (`xr.open_dataset`, `.to_netcdf`; reference
step_03_apply_to_era.py:60,378, functions.py:203, step_02_preproc_deltas.py:127-150).  Neither
xarray nor netCDF4/HDF5 is available to this build, so the drivers use this small layer on
`scipy.io.netcdf_file` (NetCDF-3 classic / 64-bit offset).  It holds only what the hot path's
callers need: named dimensions, coordinate variables, attributes, CF time decoding.
NetCDF-4/HDF5 files must be converted (`nccopy -k cdf5`/`-k 64-bit-offset`) - file formats are
outside the compute path this package replaces (SURVEY.md section 8 f, rank 1).
"""
import datetime as _dt
import re

import numpy as np
from scipy.io import netcdf_file


class Field:
    """A named-dimension array: `.values`, `.dims`, `.coords` (dim -> 1-D array), `.attrs`."""

    def __init__(self, values, dims, coords=None, attrs=None, name=None):
        self.values = np.asarray(values)
        self.dims = tuple(dims)
        if self.values.ndim != len(self.dims):
            raise ValueError('dims %s do not match array of shape %s' % (self.dims, self.values.shape))
        self.coords = dict(coords or {})
        self.attrs = dict(attrs or {})
        self.name = name

    @property
    def shape(self):
        return self.values.shape

    @property
    def dtype(self):
        return self.values.dtype

    def like(self, values, dims=None):
        """Same labels, new data (used by the functions.py mirror to re-wrap results)."""
        values = np.asarray(values)
        dims = self.dims if dims is None else tuple(dims)
        return Field(values, dims, {k: v for k, v in self.coords.items() if k in dims}, self.attrs, self.name)

    def transpose(self, *dims):
        order = [self.dims.index(d) for d in dims]
        return Field(np.transpose(self.values, order), dims, self.coords, self.attrs, self.name)

    def isel(self, **idx):
        v = self.values
        dims = list(self.dims)
        coords = dict(self.coords)
        for d, i in idx.items():
            ax = dims.index(d)
            v = np.take(v, i, axis=ax)
            if np.ndim(i) == 0:
                dims.pop(ax)
                coords.pop(d, None)
            elif d in coords:
                coords[d] = np.asarray(coords[d])[i]
        return Field(v, dims, coords, self.attrs, self.name)

    def __getitem__(self, key):
        return self.coords[key]

    def __repr__(self):
        return 'Field(%s, dims=%s, dtype=%s)' % (self.name, dict(zip(self.dims, self.shape)), self.dtype)


class Dataset:
    """Variables (name -> Field), dimension coordinates and global attributes of one file."""

    def __init__(self, variables=None, attrs=None):
        self.variables = dict(variables or {})
        self.attrs = dict(attrs or {})

    def __contains__(self, name):
        return name in self.variables

    def __getitem__(self, name):
        return self.variables[name]

    def __setitem__(self, name, field):
        if not isinstance(field, Field):
            raise TypeError('Dataset values must be Field objects')
        field.name = name
        self.variables[name] = field

    def __delitem__(self, name):
        del self.variables[name]

    def __getattr__(self, name):
        try:
            return self.__dict__['variables'][name]
        except KeyError:
            raise AttributeError(name)

    def dims(self):
        out = {}
        for f in self.variables.values():
            for d, n in zip(f.dims, f.shape):
                out.setdefault(d, n)
        return out

    def close(self):
        pass


# ------------------------------------------------------------------------------ CF time
_UNITS = re.compile(r'^\s*(\w+)\s+since\s+(\d{1,4})-(\d{1,2})-(\d{1,2})(?:[ T](\d{1,2}):(\d{1,2})(?::(\d{1,2}(?:\.\d*)?))?)?')
_SECONDS = {'seconds': 1, 'second': 1, 'secs': 1, 's': 1, 'minutes': 60, 'minute': 60, 'hours': 3600, 'hour': 3600,
            'hrs': 3600, 'h': 3600, 'days': 86400, 'day': 86400, 'd': 86400}
_CUM365 = np.array([0, 31, 59, 90, 120, 151, 181, 212, 243, 273, 304, 334, 365])


def decode_cf_time(values, units, calendar='standard'):
    """CF 'X since Y' numbers -> datetime64[s].  Calendars: standard/gregorian/
    proleptic_gregorian, noleap/365_day (dates map to the same month/day in the standard
    calendar, which is what xarray's `to_datetimeindex()` does, functions.py:219-221) and 360_day
    (day 30 of a 28/29-day February is clipped)."""
    m = _UNITS.match(units)
    if not m:
        raise ValueError('cannot parse time units %r' % units)
    unit = m.group(1).lower()
    if unit not in _SECONDS:
        raise ValueError('unsupported time unit %r' % unit)
    y, mo, d = int(m.group(2)), int(m.group(3)), int(m.group(4))
    hh, mi = int(m.group(5) or 0), int(m.group(6) or 0)
    ss = float(m.group(7) or 0)
    secs = np.asarray(values, dtype=np.float64) * _SECONDS[unit]
    cal = (calendar or 'standard').lower()
    if cal in ('standard', 'gregorian', 'proleptic_gregorian'):
        base = np.datetime64('%04d-%02d-%02dT%02d:%02d:%02d' % (y, mo, d, hh, mi, int(ss)), 's')
        return base + np.round(secs).astype('timedelta64[s]')
    out = []
    for s in np.atleast_1d(secs):
        if cal in ('noleap', '365_day'):
            base_days = y * 365 + _CUM365[mo - 1] + (d - 1)
            tot = base_days * 86400.0 + hh * 3600 + mi * 60 + ss + s
            day, rem = divmod(tot, 86400.0)
            yy, doy = divmod(int(day), 365)
            mm = int(np.searchsorted(_CUM365, doy, side='right'))
            dd = doy - _CUM365[mm - 1] + 1
        elif cal == '360_day':
            base_days = y * 360 + (mo - 1) * 30 + (d - 1)
            tot = base_days * 86400.0 + hh * 3600 + mi * 60 + ss + s
            day, rem = divmod(tot, 86400.0)
            yy, doy = divmod(int(day), 360)
            mm, dd = doy // 30 + 1, doy % 30 + 1
            dd = min(dd, 28 if mm == 2 else 30)
        else:
            raise ValueError('unsupported calendar %r' % calendar)
        rem = int(round(rem))
        out.append(np.datetime64('%04d-%02d-%02d' % (yy, mm, dd), 's') + np.timedelta64(rem, 's'))
    return np.array(out, dtype='datetime64[s]').reshape(np.shape(values))


# ------------------------------------------------------------------------------ file I/O
def _attrs(obj):
    out = {}
    for k, v in obj._attributes.items():
        if isinstance(v, bytes):
            v = v.decode('utf-8', 'replace')
        out[k] = v
    return out


def open_dataset(path, decode_times=True, threads=4):
    """Read a NetCDF-3 file completely into memory (`xr.open_dataset(...).load()`).
    decode_times=False corresponds to the reference's `decode_cf=False` (step_03:60).
    The file is memory-mapped and every variable is converted to a native-endian array in ONE
    pass (NetCDF-3 data are big-endian); large variables are converted concurrently (numpy releases
    the GIL while it copies / byte-swaps)."""
    from concurrent.futures import ThreadPoolExecutor
    try:
        nc = netcdf_file(path, 'r', mmap=True)
    except (TypeError, ValueError) as e:
        raise IOError('%s is not a NetCDF-3 file (%s). NetCDF-4/HDF5 files must be converted, e.g. '
                      '`nccopy -k 64-bit-offset in.nc out.nc`.' % (path, e))
    try:
        ds = Dataset(attrs=_attrs(nc))
        names = list(nc.variables)

        def convert(name):
            var = nc.variables[name]
            src = var.data
            native = src.dtype.newbyteorder('=') if src.dtype.byteorder in ('>', '<') else src.dtype
            return np.array(src, dtype=native, copy=True, order='C'), tuple(var.dimensions), _attrs(var)

        big = [n for n in names if nc.variables[n].data.nbytes >= (16 << 20)]
        raw = {}
        if len(big) > 1 and threads > 1:
            with ThreadPoolExecutor(max_workers=min(threads, len(big))) as pool:
                for n, r in zip(big, pool.map(convert, big)):
                    raw[n] = r
        for n in names:
            if n not in raw:
                raw[n] = convert(n)
    finally:
        try:
            nc.close()
        except Exception:           # scipy complains if views of the map are still alive; all data were copied
            pass
    coords = {}
    for name, (data, dims, attrs) in raw.items():
        if dims == (name,):
            if decode_times and 'since' in str(attrs.get('units', '')):
                data = decode_cf_time(data, attrs['units'], attrs.get('calendar', 'standard'))
            coords[name] = data
    for name in names:
        data, dims, attrs = raw[name]
        if name in coords:
            data = coords[name]
        ds.variables[name] = Field(data, dims, {d: coords[d] for d in dims if d in coords}, attrs, name)
    return ds


_NC_TYPE = {'i1': 1, 'S1': 2, 'i2': 3, 'i4': 4, 'f4': 5, 'f8': 6}


def _pad4(b):
    return b + b'\x00' * (-len(b) % 4)


def _nc_name(name):
    import struct
    raw = name.encode('utf-8')
    return struct.pack('>i', len(raw)) + _pad4(raw)


def _nc_atts(attrs):
    import struct
    if not attrs:
        return b'\x00' * 8                                   # ABSENT
    out = struct.pack('>ii', 0x0C, len(attrs))
    for k, v in attrs.items():
        out += _nc_name(k)
        if isinstance(v, bytes):
            v = v.decode('utf-8', 'replace')
        if isinstance(v, str):
            raw = v.encode('utf-8')
            out += struct.pack('>ii', 2, len(raw)) + _pad4(raw)
        else:
            arr = np.atleast_1d(np.asarray(v))
            if arr.dtype == np.int64:
                arr = arr.astype(np.int32)
            if arr.dtype == np.bool_:
                arr = arr.astype(np.int8)
            key = arr.dtype.str[1:]
            if key not in _NC_TYPE or key == 'S1':
                raw = str(v).encode('utf-8')
                out += struct.pack('>ii', 2, len(raw)) + _pad4(raw)
            else:
                out += struct.pack('>ii', _NC_TYPE[key], arr.size) + _pad4(arr.astype(arr.dtype.newbyteorder('>')).tobytes())
    return out


def to_netcdf(ds, path, threads=4):
    """Write a Dataset as NetCDF-3 64-bit-offset (`.to_netcdf(path, mode='w')`, step_03:378).

    Native writer (the classic format is a header + fixed-size big-endian arrays): every variable
    is byte-swapped in one pass and written with `os.pwrite` at its offset, large variables
    concurrently - scipy's writer makes three copies of every array under the GIL and was the
    bottleneck of the whole command line (1.06 s per 2.3 GB file; PGW_NC_WRITER=scipy selects it)."""
    import os
    import struct
    from concurrent.futures import ThreadPoolExecutor
    if os.environ.get('PGW_NC_WRITER') == 'scipy':
        return _to_netcdf_scipy(ds, path)
    dims = {}
    for f in ds.variables.values():
        for d, n in zip(f.dims, f.shape):
            if d in dims and dims[d] != int(n):
                raise ValueError('dimension %s has inconsistent lengths %d and %d' % (d, dims[d], n))
            dims.setdefault(d, int(n))
    dim_ids = {d: i for i, d in enumerate(dims)}
    # variables: data converted lazily (dtype decided here)
    specs = []
    for name, f in ds.variables.items():
        data = f.values
        attrs = dict(f.attrs)
        if data.dtype.kind == 'M':
            data = (data.astype('datetime64[s]') - np.datetime64('1970-01-01T00:00:00', 's')).astype(np.float64)
            attrs['units'] = 'seconds since 1970-01-01 00:00:00'
            attrs['calendar'] = 'proleptic_gregorian'
        if data.dtype == np.int64:
            data = data.astype(np.int32)
        if data.dtype == np.bool_:
            data = data.astype(np.int8)
        key = data.dtype.str[1:]
        if key not in _NC_TYPE or key == 'S1':
            raise TypeError('variable %s: dtype %s cannot be stored in NetCDF-3' % (name, data.dtype))
        nbytes = int(data.size) * data.dtype.itemsize
        specs.append(dict(name=name, f=f, data=data, attrs=attrs, key=key, nbytes=nbytes, vsize=nbytes + (-nbytes % 4)))

    def header(begins):
        h = b'CDF\x02' + struct.pack('>i', 0)
        if dims:
            h += struct.pack('>ii', 0x0A, len(dims))
            for d, n in dims.items():
                h += _nc_name(d) + struct.pack('>i', n)
        else:
            h += b'\x00' * 8
        h += _nc_atts(ds.attrs)
        if specs:
            h += struct.pack('>ii', 0x0B, len(specs))
            for sp, b in zip(specs, begins):
                h += _nc_name(sp['name']) + struct.pack('>i', len(sp['f'].dims))
                for d in sp['f'].dims:
                    h += struct.pack('>i', dim_ids[d])
                h += _nc_atts(sp['attrs'])
                h += struct.pack('>ii', _NC_TYPE[sp['key']], min(sp['vsize'], 0xFFFFFFFF - 3) if sp['vsize'] < 2 ** 32 else -1)
                h += struct.pack('>q', b)
        else:
            h += b'\x00' * 8
        return h

    hlen = len(header([0] * len(specs)))
    begins, off = [], hlen
    for sp in specs:
        begins.append(off)
        off += sp['vsize']
    hdr = header(begins)
    assert len(hdr) == hlen
    fd = os.open(path, os.O_WRONLY | os.O_CREAT | os.O_TRUNC, 0o644)
    try:
        os.ftruncate(fd, off)
        os.pwrite(fd, hdr, 0)
        CH = 64 << 20                                            # swap + write in 64 MiB pieces

        def write_var(i):
            sp, b = specs[i], begins[i]
            flat = np.ascontiguousarray(sp['data']).reshape(-1)
            be = flat.dtype.newbyteorder('>')
            step = max(CH // max(flat.dtype.itemsize, 1), 1)
            pos = b
            for s0 in range(0, flat.size, step):
                chunk = flat[s0:s0 + step].astype(be)              # one pass: copy + byte swap
                mv = memoryview(chunk).cast('B')
                while len(mv):
                    n = os.pwrite(fd, mv, pos)
                    pos += n
                    mv = mv[n:]
            # padding bytes are already zero (ftruncate)

        big = [i for i, sp in enumerate(specs) if sp['nbytes'] >= (16 << 20)]
        if len(big) > 1 and threads > 1:
            with ThreadPoolExecutor(max_workers=min(threads, len(big))) as pool:
                list(pool.map(write_var, big))
        else:
            for i in big:
                write_var(i)
        for i in range(len(specs)):
            if i not in big:
                write_var(i)
    finally:
        os.close(fd)


def _to_netcdf_scipy(ds, path):
    """Write a Dataset as NetCDF-3 64-bit-offset (`.to_netcdf(path, mode='w')`, step_03:378)."""
    nc = netcdf_file(path, 'w', version=2)
    for k, v in ds.attrs.items():
        setattr(nc, k, v)
    made = {}
    for f in ds.variables.values():
        for d, n in zip(f.dims, f.shape):
            if d not in made:
                nc.createDimension(d, int(n))
                made[d] = n
    for name, f in ds.variables.items():
        data = f.values
        attrs = dict(f.attrs)
        if data.dtype.kind == 'M':
            data = (data.astype('datetime64[s]') - np.datetime64('1970-01-01T00:00:00', 's')).astype(np.float64)
            attrs['units'] = 'seconds since 1970-01-01 00:00:00'
            attrs['calendar'] = 'proleptic_gregorian'
        if data.dtype == np.int64:
            data = data.astype(np.int32)
        if data.dtype == np.bool_:
            data = data.astype(np.int8)
        var = nc.createVariable(name, data.dtype.newbyteorder('='), f.dims)
        var[...] = data
        for k, v in attrs.items():
            if k in ('_FillValue',) and np.ndim(v) == 0:
                v = np.asarray(v, dtype=data.dtype)
            setattr(var, k, v)
    nc.close()
