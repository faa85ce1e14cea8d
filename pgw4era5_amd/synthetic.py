"""
Deterministic synthetic ERA5 file + climate-delta set of the shape step_03 touches
(SURVEY.md section 8d / appendix B).  Used by bench.py, the tests and smoke(); there is no
network and the reference ships no sample data (.gitignore:11-12 of the reference excludes
*.nc), so every run is on these arrays.

The hybrid coefficients are a generated monotone set, NOT ECMWF's L137 table (which is in
neither the reference nor the container): ak[0] = bk[0] = 0 so that the top half level hits
the `p > 0` guard of integ_geopot (functions.py:135), bk[N] = 1, ak[N] = 0, pure-pressure
levels above ~70 hPa and a smooth blend below.
"""
import datetime as _dt

import numpy as np

CON_G = 9.80665

# CMIP6 Amon plev19 [Pa], descending as in the files (step_01 output)
PLEV19 = np.array([100000., 92500., 85000., 70000., 60000., 50000., 40000., 30000., 25000.,
                   20000., 15000., 10000., 7000., 5000., 3000., 2000., 1000., 500., 100.])


def hybrid_coefficients(nlev):
    """Monotone synthetic hybrid half-level coefficients (ak [Pa], bk [1]), length nlev+1.

    Reference half-level pressures (at ps = 101325 Pa) follow p_k = 101325 (k/N)^a with a chosen
    so that the first non-zero half level sits at 1 Pa (for N = 137: a = 2.34, 55 full levels
    below 300 hPa - close to ECMWF's L137 distribution)."""
    n = nlev
    k = np.arange(n + 1, dtype=np.float64)
    a = np.log(101325.0) / np.log(float(n)) if n > 1 else 1.0
    p_ref = 101325.0 * (k / n) ** a
    p_ref[0] = 0.0
    # b: zero above 7000 Pa, smooth monotone blend to 1 at the surface
    s = np.clip((p_ref - 7000.0) / (101325.0 - 7000.0), 0.0, 1.0)
    bk = s ** 1.6
    bk[-1] = 1.0
    ak = p_ref - bk * 101325.0
    ak[-1] = 0.0
    ak[0] = 0.0
    bk[0] = 0.0
    return ak, bk


def _smooth2d(rng, nlat, nlon, nmodes=6):
    """Smooth random field in [0,1] from a few low-order harmonics (periodic in lon)."""
    lat = np.linspace(-0.5 * np.pi, 0.5 * np.pi, nlat)[:, None]
    lon = np.linspace(0, 2 * np.pi, nlon, endpoint=False)[None, :]
    f = np.zeros((nlat, nlon))
    for _ in range(nmodes):
        kx = rng.integers(1, 5); ky = rng.integers(1, 4)
        ph1, ph2 = rng.uniform(0, 2 * np.pi, 2)
        f += rng.uniform(0.3, 1.0) * np.sin(kx * lon + ph1) * np.cos(ky * lat + ph2)
    f -= f.min()
    f /= max(f.max(), 1e-30)
    return f


def make_case(nlat=10, nlon=10, nlev=20, seed=0, dtype=np.float64, plev=None, nsoil=4,
              target_dt=None, noise=True):
    """Build one synthetic ERA5 file and the monthly deltas on the same grid.

    Returns dict(era=..., deltas=..., delta_times=..., plev=..., target_dt=..., lat, lon).
    4-D arrays are C-order (time=1, lev, lat, lon) in `dtype`; coefficients are float64.
    """
    rng = np.random.default_rng(seed)
    plev = PLEV19 if plev is None else np.asarray(plev, dtype=np.float64)
    S = len(plev)
    dt = np.dtype(dtype)
    ak, bk = hybrid_coefficients(nlev)
    akm = 0.5 * (ak[1:] - ak[:-1]) + ak[:-1]
    bkm = 0.5 * (bk[1:] - bk[:-1]) + bk[:-1]

    orog = 5000.0 * _smooth2d(rng, nlat, nlon) ** 3
    land = (_smooth2d(rng, nlat, nlon) > 0.5).astype(np.float64)
    orog *= land                                         # sea level over ocean
    fis = CON_G * orog
    ps = 101325.0 * np.exp(-orog / 8000.0) * (1 + 0.01 * np.clip(rng.standard_normal((nlat, nlon)), -3, 3))

    # 3-D fields level by level (8 MB slabs stay in cache; no full-size temporaries)
    shp4 = (1, nlev, nlat, nlon)
    T = np.empty(shp4, dtype=dt); QV = np.empty(shp4, dtype=dt)
    U = np.empty(shp4, dtype=dt); V = np.empty(shp4, dtype=dt)
    rh2d = rng.uniform(10.0, 95.0, (nlat, nlon))
    T0, Ti = 273.16, 250.16
    f32 = np.float32
    for l in range(nlev):
        pa = akm[l] + ps * bkm[l]
        t = np.maximum(288.0 + 0.0065 * 8000.0 * np.log(pa / 101325.0), 215.0)
        if noise:
            t = t + rng.standard_normal((nlat, nlon), dtype=f32)
        t = t.astype(dt).astype(np.float64)           # humidity consistent with the stored T
        # humidity from RH ~ U(10,95) % through the IFS formulas (functions.py:74-125)
        alpha = np.where(t >= T0, 1.0, np.where(t <= Ti, 0.0, ((t - Ti) / (T0 - Ti)) ** 2))
        es = (alpha * 611.21 * np.exp(17.502 * (t - T0) / (t - 32.19)) +
              (1 - alpha) * 611.21 * np.exp(22.587 * (t - T0) / (t + 0.7)))
        # RH decays above 200 hPa so that the stratosphere is dry (q of a few 1e-6, e << p)
        e = rh2d * np.clip(pa / 20000.0, 0.0, 1.0) ** 3 / 100.0 * es
        T[0, l] = t
        QV[0, l] = 0.622 * e / (pa - 0.378 * e)
        if noise:
            U[0, l] = 10.0 * rng.standard_normal((nlat, nlon), dtype=f32)
            V[0, l] = 10.0 * rng.standard_normal((nlat, nlon), dtype=f32)
        else:
            U[0, l] = 5.0
            V[0, l] = -3.0

    sic = np.clip(_smooth2d(rng, nlat, nlon) * 1.5 - 0.7, 0, 1)
    sic = np.where(land > 0.5, np.nan, sic)
    soil1 = np.array([0.035, 0.175, 0.64, 1.945])[:nsoil]
    t_skin = 288.0 - 0.0065 * orog + rng.standard_normal((nlat, nlon))
    era = dict(
        ak=ak, bk=bk,
        PS=ps[None].astype(dt), FIS=fis[None].astype(dt), T=T, QV=QV, U=U, V=V,
        T_SKIN=t_skin[None].astype(dt),
        T_SO=(t_skin[None, None] + np.zeros((1, nsoil, 1, 1))).astype(dt),
        FR_LAND=land[None].astype(dt), FR_SEA_ICE=sic[None].astype(dt),
        soil1=soil1, level=np.arange(1, nlev + 1), level1=np.arange(1, nlev + 2),
    )

    # ---- monthly deltas on the ERA5 grid ------------------------------------------------
    months = np.arange(12)
    season = np.cos(2 * np.pi * (months - 0.5) / 12.0)          # (12,)
    pat = _smooth2d(rng, nlat, nlon)                            # (lat,lon) in [0,1]
    prof = np.clip(1.0 + 4.0 * np.exp(-((np.log(plev) - np.log(30000.0)) / 1.2) ** 2), 1.0, 5.0)
    prof = np.where(plev < 10000.0, 1.0 - 3.0 * (1 - plev / 10000.0), prof)   # stratospheric cooling
    def outer(prof_ts, pat2d):
        """(12, S) profile x (lat, lon) pattern -> (12, S, lat, lon) in the storage dtype."""
        o = np.empty((12, S, nlat, nlon), dtype=dt)
        np.multiply(prof_ts.astype(dt)[:, :, None, None], pat2d.astype(dt)[None, None], out=o)
        return o

    sea = season[:, None]
    d_ta = outer(prof[None, :] * (0.8 + 0.2 * sea), 0.8 + 0.4 * pat)
    d_hur = outer(5.0 * np.cos(np.linspace(0, np.pi, S))[None, :] * (1 + 0.2 * sea),
                  2 * _smooth2d(rng, nlat, nlon) - 1)
    d_ua = outer(2.0 * np.ones((12, S)), 2 * _smooth2d(rng, nlat, nlon) - 1)
    d_va = outer(2.0 * np.ones((12, S)), 2 * _smooth2d(rng, nlat, nlon) - 1)
    # geopotential-height delta consistent with a warmer column: grows with height
    h = np.clip(np.log(100000.0 / plev) / np.log(100000.0 / 100.0), 0, 1)
    d_zg = outer((20.0 + 100.0 * h[None, :] ** 0.7) * (1 + 0.1 * sea), 0.9 + 0.2 * pat)
    d_tas = 2.0 * (0.8 + 0.2 * season[:, None, None]) * (0.8 + 0.4 * pat[None])
    d_hurs = -2.0 * (2 * pat[None] - 1) * np.ones((12, 1, 1))
    d_ts = d_tas * 1.05
    d_tos = np.where(land[None] > 0.5, np.nan, 0.8 * d_tas)
    d_sic = -20.0 * np.clip(_smooth2d(rng, nlat, nlon), 0, 1)[None] * np.ones((12, 1, 1))
    ps_hist = ps[None] * (1 + 0.002 * (2 * _smooth2d(rng, nlat, nlon)[None] - 1)) * np.ones((12, 1, 1))
    deltas = dict(ta=d_ta, hur=d_hur, ua=d_ua, va=d_va, zg=d_zg)
    for k, v in dict(tas=d_tas, hurs=d_hurs, ts=d_ts, tos=d_tos, siconc=d_sic, ps_hist=ps_hist).items():
        deltas[k] = np.ascontiguousarray(v).astype(dt)
    delta_times = np.array(['1995-%02d-15T12:00:00' % (m + 1) for m in months], dtype='datetime64[s]')
    if target_dt is None:
        target_dt = _dt.datetime(2006, 8, 2, 3)
    lat = np.linspace(-90, 90, nlat) if nlat > 1 else np.array([0.0])
    lon = np.arange(nlon) * (360.0 / nlon)
    return dict(era=era, deltas=deltas, delta_times=delta_times, plev=plev,
                target_dt=target_dt, lat=lat, lon=lon)


def make_gcm_grid_case(nlat_src=48, nlon_src=96, nlat=37, nlon=72, nplev=5, ntime=3, seed=0,
                       dtype=np.float64):
    """step_02 regridding case: Gaussian-like source lats that do not reach the poles, periodic
    source lons starting at 0; target regular grid including +-90 (cfg 4)."""
    rng = np.random.default_rng(seed)
    # Gaussian-like: ascending, nearly equally spaced, first/last row half a cell off the pole
    x = (np.arange(nlat_src) + 0.5) / nlat_src
    src_lat = (-90.0 + 180.0 * x)
    src_lat = src_lat * (1 - 0.3 / nlat_src)
    src_lon = np.arange(nlon_src) * (360.0 / nlon_src)
    targ_lat = np.linspace(-90.0, 90.0, nlat)
    targ_lon = np.arange(nlon) * (360.0 / nlon)
    f = rng.standard_normal((ntime, nplev, nlat_src, nlon_src)).astype(dtype)
    return dict(field=f, src_lat=src_lat, src_lon=src_lon, targ_lat=targ_lat, targ_lon=targ_lon)


def make_ocean_grid_case(nj=40, ni=60, ntime=12, seed=0, land_patches=3):
    """An ocean-model-like delta for step_02's NaN-ignoring interpolation (tos / siconc): curvilinear grid with 2-D
    `latitude` / `longitude` coordinates (a regular grid sheared and stretched so that rows are not parallels, longitudes
    0 ... 360 like CMIP ocean output), values smooth in space, NaN over a few "continents".
    Returns dict(latitude (nj, ni), longitude (nj, ni), values (ntime, nj, ni), times)."""
    rng = np.random.default_rng(seed)
    j = (np.arange(nj) + 0.5) / nj
    i = (np.arange(ni) + 0.5) / ni
    lat0 = -78.0 + 166.0 * j                                   # -78 ... 88: no ocean points at the south pole
    lon0 = 360.0 * i
    lat2 = lat0[:, None] + 3.0 * np.sin(2 * np.pi * i)[None, :] * np.cos(np.deg2rad(lat0))[:, None]
    lon2 = (lon0[None, :] + 8.0 * (j[:, None] - 0.5) ** 2 * 4.0) % 360.0
    lat2 = np.clip(lat2, -89.5, 89.5)
    base = 1.5 + np.cos(np.deg2rad(lat2)) * (1.0 + 0.3 * np.sin(np.deg2rad(2 * lon2)))
    season = 1.0 + 0.2 * np.cos(2 * np.pi * (np.arange(ntime) - 0.5) / max(ntime, 1))
    vals = season[:, None, None] * base[None]
    land = np.zeros((nj, ni), dtype=bool)
    for _ in range(land_patches):
        cj, ci = rng.integers(nj // 6, 5 * nj // 6), rng.integers(0, ni)
        rj, ri = rng.integers(2, max(nj // 6, 3)), rng.integers(2, max(ni // 6, 3))
        jj, ii = np.ogrid[:nj, :ni]
        di = np.minimum(np.abs(ii - ci), ni - np.abs(ii - ci))
        land |= ((jj - cj) / rj) ** 2 + (di / ri) ** 2 <= 1.0
    vals = np.where(land[None], np.nan, vals)
    times = np.array(['1995-%02d-15T12:00:00' % (m % 12 + 1) for m in range(ntime)], dtype='datetime64[s]')
    return dict(latitude=lat2, longitude=lon2, values=vals, times=times, land=land)


def write_gcm_files(gcm_dir, nlat=24, nlon=48, plev=None, seed=0, ocean=(30, 44)):
    """The input of step_02 `regridding` for ALL default variables (step_02_preproc_deltas.py:77-80): for every variable
    `{var}_delta.nc` and `{var}_historical.nc` on a coarse Gaussian-like GCM grid (lats short of the poles, lon 0 ... 360 - d)
    with 12 monthly records - ta, hur, ua, va, zg on `plev` (descending), tas, hurs, ts, ps 2-D - and tos, siconc on an
    ocean model's curvilinear grid (2-D latitude / longitude, NaN over land).  Values are smooth in space so that the
    regridded deltas are physically plausible inputs of step_03 (historical ps between 950 and 1030 hPa)."""
    import os
    from . import ncio
    rng = np.random.default_rng(seed)
    os.makedirs(gcm_dir, exist_ok=True)
    plev = PLEV19 if plev is None else np.asarray(plev, dtype=np.float64)
    S = len(plev)
    x = (np.arange(nlat) + 0.5) / nlat
    lat = (-90.0 + 180.0 * x) * (1 - 0.3 / nlat)
    lon = np.arange(nlon) * (360.0 / nlon)
    la, lo = np.deg2rad(lat)[:, None], np.deg2rad(lon)[None, :]
    pat = 0.5 + 0.5 * np.cos(la) * np.sin(2 * lo + 0.3) * np.cos(1.5 * la)                 # (lat, lon) in [0, 1]
    pat2 = 0.5 + 0.5 * np.sin(la * 2) * np.cos(lo - 0.7)
    months = np.arange(12)
    sea = np.cos(2 * np.pi * (months - 0.5) / 12.0)
    times = np.array(['1995-%02d-15T12:00:00' % (m + 1) for m in months], dtype='datetime64[s]')
    prof = np.clip(1.0 + 4.0 * np.exp(-((np.log(plev) - np.log(30000.0)) / 1.2) ** 2), 1.0, 5.0)
    prof = np.where(plev < 10000.0, 1.0 - 3.0 * (1 - plev / 10000.0), prof)
    h = np.clip(np.log(100000.0 / plev) / np.log(100000.0 / 100.0), 0, 1)

    def f4(p, field2d, amp=1.0):
        return amp * (1 + 0.2 * sea)[:, None, None, None] * p[None, :, None, None] * field2d[None, None]

    def f3(field2d, amp=1.0):
        return amp * (1 + 0.2 * sea)[:, None, None] * field2d[None]

    v4 = dict(ta=f4(prof, 0.8 + 0.4 * pat), hur=f4(5.0 * np.cos(np.linspace(0, np.pi, S)), 2 * pat2 - 1),
              ua=f4(np.ones(S), 2 * pat - 1, 2.0), va=f4(np.ones(S), 2 * pat2 - 1, 2.0),
              zg=f4(20.0 + 100.0 * h ** 0.7, 0.9 + 0.2 * pat))
    v3 = dict(tas=f3(0.8 + 0.4 * pat, 2.0), hurs=f3(1 - 2 * pat2, 2.0), ts=f3(0.8 + 0.4 * pat, 2.1),
              ps=np.ones((12, 1, 1)) * (95000.0 + 8000.0 * pat)[None])
    F = ncio.Field
    for base in ('{}_delta.nc', '{}_historical.nc'):
        scale = 1.0 if 'delta' in base else 1.0
        for var, arr in v4.items():
            ds = ncio.Dataset()
            ds['time'] = F(times, ('time',)); ds['plev'] = F(plev, ('plev',), attrs=dict(units='Pa'))
            ds['lat'] = F(lat, ('lat',)); ds['lon'] = F(lon, ('lon',))
            ds[var] = F(arr * scale, ('time', 'plev', 'lat', 'lon'))
            ncio.to_netcdf(ds, os.path.join(gcm_dir, base.format(var)))
        for var, arr in v3.items():
            ds = ncio.Dataset()
            ds['time'] = F(times, ('time',)); ds['lat'] = F(lat, ('lat',)); ds['lon'] = F(lon, ('lon',))
            ds[var] = F(arr * scale, ('time', 'lat', 'lon'))
            ncio.to_netcdf(ds, os.path.join(gcm_dir, base.format(var)))
        oc = make_ocean_grid_case(nj=ocean[0], ni=ocean[1], ntime=12, seed=seed + 1)
        for var, vals in (('tos', 0.8 * oc['values']), ('siconc', -8.0 * oc['values'])):
            ds = ncio.Dataset()
            ds['time'] = F(oc['times'], ('time',))
            ds['latitude'] = F(oc['latitude'], ('j', 'i')); ds['longitude'] = F(oc['longitude'], ('j', 'i'))
            ds[var] = F(vals, ('time', 'j', 'i'))
            ncio.to_netcdf(ds, os.path.join(gcm_dir, base.format(var)))
    return dict(lat=lat, lon=lon, plev=plev, times=times)


def write_case_files(case, era_dir, delta_dir, era_name=None):
    """Write a make_case() result as the NetCDF-3 files the step_03 driver reads: one ERA5 file
    (reference file schema) and the delta directory ({var}_delta.nc,
    ps_historical.nc).  Returns the ERA5 file path."""
    import os
    from . import ncio, settings as S
    os.makedirs(era_dir, exist_ok=True)
    os.makedirs(delta_dir, exist_ok=True)
    era, lat, lon = case['era'], case['lat'], case['lon']
    nlev = era['T'].shape[1]
    F = ncio.Field
    ds = ncio.Dataset(attrs=dict(title='synthetic ERA5 file (pgw4era5_amd.synthetic)'))
    tsec = (np.datetime64(case['target_dt']).astype('datetime64[s]') - np.datetime64('1970-01-01T00:00:00')).astype(np.float64)
    cv = dict(time=np.array([tsec]), level=np.arange(1, nlev + 1, dtype=np.float64),
              level1=np.arange(1, nlev + 2, dtype=np.float64), soil1=np.asarray(era['soil1'], dtype=np.float64),
              lat=np.asarray(lat, dtype=np.float64), lon=np.asarray(lon, dtype=np.float64))
    for k, v in cv.items():
        attrs = dict(units='seconds since 1970-01-01 00:00:00') if k == 'time' else {}
        ds[k] = F(v, (k,), {k: v}, attrs)
    ds['ak'] = F(era['ak'], ('level1',), {'level1': cv['level1']})
    ds['bk'] = F(era['bk'], ('level1',), {'level1': cv['level1']})
    d4 = ('time', 'level', 'lat', 'lon')
    d3 = ('time', 'lat', 'lon')
    # a scalar NC_CHAR variable that only carries attributes, as in COSMO / int2lm boundary files (`char rotated_pole`):
    # the driver has to pass it through untouched
    ds['rotated_pole'] = F(np.array(b'', dtype='S1'), (), attrs=dict(grid_mapping_name='rotated_latitude_longitude',
                                                                     grid_north_pole_latitude=np.float32(43.0),
                                                                     grid_north_pole_longitude=np.float32(-170.0)))
    for name in ('T', 'QV', 'U', 'V'):
        ds[name] = F(era[name], d4, {d: cv[d] for d in d4}, attrs=dict(grid_mapping='rotated_pole'))
    for name in ('PS', 'FIS', 'T_SKIN', 'FR_LAND', 'FR_SEA_ICE'):
        ds[name] = F(era[name], d3, {d: cv[d] for d in d3})
    ds['T_SO'] = F(era['T_SO'], ('time', 'soil1', 'lat', 'lon'), {d: cv[d] for d in ('time', 'soil1', 'lat', 'lon')})
    name = era_name or S.era5_file_name_base.format(case['target_dt'])
    path = os.path.join(era_dir, name)
    ncio.to_netcdf(ds, path)
    # deltas (case['delta_times']: one time axis for all files, or a dict var -> axis: every delta file has its own)
    for var, arr in case['deltas'].items():
        times = case['delta_times'][var] if isinstance(case['delta_times'], dict) else case['delta_times']
        tdays = (np.asarray(times).astype('datetime64[s]') - np.datetime64('1850-01-01T00:00:00')).astype('timedelta64[s]').astype(np.float64) / 86400.0
        dd = ncio.Dataset()
        dd['time'] = F(tdays, ('time',), attrs=dict(units='days since 1850-01-01 00:00:00', calendar='proleptic_gregorian'))
        dd['lat'] = F(cv['lat'], ('lat',))
        dd['lon'] = F(cv['lon'], ('lon',))
        if arr.ndim == 4:
            dd['plev'] = F(np.asarray(case['plev'], dtype=np.float64), ('plev',), attrs=dict(units='Pa'))
            dims = ('time', 'plev', 'lat', 'lon')
        else:
            dims = ('time', 'lat', 'lon')
        vname, fname = (('ps', S.file_name_bases['HIST'].format('ps')) if var == 'ps_hist'
                        else (var, S.file_name_bases['SCEN-HIST'].format(var)))
        dd[vname] = F(arr, dims)
        ncio.to_netcdf(dd, os.path.join(delta_dir, fname))
    return path


def resample_deltas(case, stamps_by_var, seed=0, noise=0.02):
    """Give the delta files of a make_case() result their own time axes (the reference loads every file on its own,
    functions.py:195-303): `stamps_by_var[var]` = new stamps (datetime64) of that variable; its records are the monthly
    records interpolated periodically to those stamps plus a little record-dependent noise (so that neighbouring records
    differ and a wrong bracket shows).  Variables not named keep the 12 monthly records.  Returns (deltas, times_by_var)."""
    rng = np.random.default_rng(seed)
    base_t = np.asarray(case['delta_times']).astype('datetime64[s]')
    doy = lambda t: (t - t.astype('datetime64[Y]')).astype('timedelta64[s]').astype(np.float64) / 86400.0
    x0 = doy(base_t)
    deltas, times = {}, {}
    for var, arr in case['deltas'].items():
        if var not in stamps_by_var:
            deltas[var], times[var] = arr, base_t
            continue
        st = np.asarray(stamps_by_var[var]).astype('datetime64[s]')
        x = doy(st)
        xp = np.concatenate([[x0[-1] - 365.0], x0, [x0[0] + 365.0]])
        a64 = np.asarray(arr, dtype=np.float64)
        ext = np.concatenate([a64[-1:], a64, a64[:1]])
        j = np.clip(np.searchsorted(xp, x, side='right') - 1, 0, len(xp) - 2)
        w = ((x - xp[j]) / (xp[j + 1] - xp[j])).reshape((-1,) + (1,) * (a64.ndim - 1))
        new = ext[j] * (1 - w) + ext[j + 1] * w
        scale = np.nanstd(a64) or 1.0
        new = new + noise * scale * rng.normal(size=(len(st),) + (1,) * (a64.ndim - 1))
        new[np.isnan(ext[j]) | np.isnan(ext[j + 1])] = np.nan
        if var == 'siconc':
            new = np.minimum(new, 0.0)
        deltas[var], times[var] = new.astype(arr.dtype), st
    return deltas, times
