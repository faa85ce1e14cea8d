"""step_02 for tos / siconc: the NaN-ignoring Gaussian-kernel interpolation from an ocean model's curvilinear grid
(reference functions.py:900-1060, interp_wrapper :1095-1135).  PARITY UNPINNED: pyproj / pyvista / VTK are not
installable, the reference ships no fixture for it; the checker is oracle/pgw_oracle.py's restatement of the published
algorithms (Vincenty inverse; vtkGaussianKernel) plus properties the scheme must have."""
import os

import numpy as np
import pytest

from oracle import pgw_oracle as O


# ------------------------------------------------------------------ geodesy (host side, no GPU)
def test_geodesy_known_answers_and_oracle():
    from pgw4era5_amd import geodesy as G
    assert abs(G.QUARTER_MERIDIAN - 10001965.7293) < 1e-3                      # WGS84 quarter meridian
    assert abs(float(G.same_latitude_geodesic(0.0, 1.0)) - 111319.4908) < 1e-3  # one degree of the equator: a * pi / 180
    assert float(G.same_latitude_geodesic(0.0, 180.0)) == 2 * G.QUARTER_MERIDIAN  # antipodal: over a pole
    assert float(G.same_latitude_geodesic(37.0, 0.0)) == 0.0 and float(G.same_latitude_geodesic(90.0, 77.0)) == 0.0
    # on the equator the geodesic is the equator up to (1 - f) * 180 deg, then it leaves it (and gets shorter than a * L)
    assert float(G.same_latitude_geodesic(0.0, 179.0)) == pytest.approx(G.WGS84_A * np.deg2rad(179.0), rel=1e-15)
    assert G.WGS84_A * np.deg2rad(179.7) - float(G.same_latitude_geodesic(0.0, 179.7)) > 1e3
    # against the oracle's Vincenty inverse iteration (a different arrangement of the same series)
    rng = np.random.default_rng(0)
    lat, lon = rng.uniform(-89, 89, 300), rng.uniform(-179, 179, 300)
    a, b = O.planar_metres(lat, lon), G.planar_metres(lat, lon)
    for x, y in zip(a, b):
        np.testing.assert_allclose(y, x, rtol=0, atol=1e-4)                     # metres (the truncation level of the series)
    # monotone in longitude along every parallel (the planar map keeps the order of the points)
    la, lo = np.meshgrid(np.linspace(-89, 89, 60), np.linspace(-180, 180, 241), indexing='ij')
    _, y, off = G.planar_metres(la.ravel(), lo.ravel())
    assert (np.diff(y.reshape(la.shape), axis=1) > 0).all()
    np.testing.assert_allclose(y.reshape(la.shape)[:, -1], off.reshape(la.shape)[:, -1], rtol=1e-15)   # lon = 180: the offset itself


# ------------------------------------------------------------------ GPU
def _era_grid(nlat, nlon):
    lat = np.linspace(-90.0, 90.0, nlat)
    lon = np.arange(nlon) * (360.0 / nlon)                                     # 0 ... 360 like the ERA5 files
    return lat, lon


@pytest.mark.gpu
def test_planar_metres_on_the_gpu_equals_the_host_geodesy():
    """pgw_planar_metres (one thread per point: meridian arc, same-parallel geodesic by bisection, over-the-pole length)
    against pgw4era5_amd/geodesy.py - the same formulas in numpy - incl. the poles, the equator up to and beyond
    (1 - f) 180 deg, lon = 0 and +-180, and against the oracle's Vincenty inverse where that converges."""
    from pgw4era5_amd import functions as F, geodesy as G
    rng = np.random.default_rng(4)
    lat = np.concatenate([rng.uniform(-90, 90, 4000), [0.0, 0.0, 0.0, 0.0, 90.0, -90.0, 45.0, 45.0, 1e-9, -37.0, 0.0]])
    lon = np.concatenate([rng.uniform(-180, 180, 4000), [179.0, 179.7, -179.9, 180.0, 33.0, -120.0, 0.0, 180.0, 170.0, -180.0, 0.0]])
    got = F.planar_metres(lat, lon)
    want = G.planar_metres(lat, lon)
    for g, w, name in zip(got, want, ('lat_m', 'lon_m', 'lon_off')):
        np.testing.assert_allclose(g, w, rtol=0, atol=2e-6, err_msg=name)           # metres
    la, lo = rng.uniform(-89, 89, 300), rng.uniform(-179, 179, 300)
    for g, w in zip(F.planar_metres(la, lo), O.planar_metres(la, lo)):
        np.testing.assert_allclose(g, w, rtol=0, atol=1e-4)
    assert all(len(x) == 0 for x in F.planar_metres(np.zeros(0), np.zeros(0)))


@pytest.mark.gpu
def test_gauss_interp_vs_oracle_small():
    from pgw4era5_amd import functions as F, synthetic
    oc = synthetic.make_ocean_grid_case(nj=18, ni=26, ntime=3, seed=2)
    lat, lon = _era_grid(13, 20)                                               # 15 deg / 18 deg: no nearly antipodal equator points
    rng = np.random.default_rng(1)
    land = (rng.uniform(size=(13, 20)) > 0.8).astype(np.float64)
    R, s = 2.5e6, 4.0
    got = F.gauss_interp_fields(land, lat, lon, oc['latitude'], oc['longitude'], list(oc['values']), R, s)
    assert got.shape == (3, 13, 20)
    for m in range(3):
        want = O.nan_ignoring_interp(land, lat, lon, oc['latitude'], oc['longitude'], oc['values'][m], R, s)
        np.testing.assert_allclose(got[m], want, rtol=1e-10, atol=0, equal_nan=True)
        assert np.isnan(want).sum() >= (land > 0.7).sum() > 0
    # a month with its own NaN pattern: that month's cloud loses the points, the others do not change
    v2 = oc['values'].copy()
    v2[1, 3:9, 5:15] = np.nan
    got2 = F.gauss_interp_fields(land, lat, lon, oc['latitude'], oc['longitude'], list(v2), R, s)
    np.testing.assert_array_equal(got2[0], got[0]); np.testing.assert_array_equal(got2[2], got[2])
    want = O.nan_ignoring_interp(land, lat, lon, oc['latitude'], oc['longitude'], v2[1], R, s)
    np.testing.assert_allclose(got2[1], want, rtol=1e-10, equal_nan=True)
    assert np.nanmax(np.abs(got2[1] - got[1])) > 1e-6


@pytest.mark.gpu
def test_gauss_interp_properties():
    from pgw4era5_amd import functions as F, synthetic, geodesy as G
    oc = synthetic.make_ocean_grid_case(nj=120, ni=200, ntime=2, seed=3)
    lat, lon = _era_grid(181, 360)
    land = np.zeros((181, 360))
    land[100:110, 40:60] = 1.0
    R, s = 1.0e6, 4.0                                                          # settings.py:127-129 defaults
    const = np.where(np.isnan(oc['values'][0]), np.nan, 2.5)
    got = F.gauss_interp_fields(land, lat, lon, oc['latitude'], oc['longitude'], [oc['values'][0], const], R, s)
    # a constant field stays constant wherever a value exists; a weighted mean stays inside the data range
    ok = ~np.isnan(got[1])
    np.testing.assert_allclose(got[1][ok], 2.5, rtol=1e-14)
    assert np.nanmin(got[0]) >= np.nanmin(oc['values'][0]) - 1e-12 and np.nanmax(got[0]) <= np.nanmax(oc['values'][0]) + 1e-12
    # land points of the ERA5 grid are NaN, and so is everything farther than R from every ocean point (the south pole cap:
    # the ocean grid ends at 78 S, 1000 km reaches to about 87 S)
    assert np.isnan(got[0][100:110, 40:60]).all()
    assert np.isnan(got[0][0]).all() and not np.isnan(got[0][10]).any()
    assert np.isnan(got[0]).sum() == np.isnan(got[1]).sum()
    # the +-2 * lon_offset copies of the cloud (functions.py:977-991) serve the targets next to lon = +-180: a strip of
    # targets on both sides of the date line against the oracle (which builds the three copies literally)
    rows = np.array([20, 45, 75, 120, 150])                                   # not the equator row: nearly antipodal pairs
    cols = np.array([178, 179, 180, 181, 182])                                # lon 178 ... 182 -> 178, 179, 180, -179, -178
    want = O.nan_ignoring_interp(np.zeros((len(rows), len(cols))), lat[rows], lon[cols], oc['latitude'], oc['longitude'],
                                 oc['values'][0], R, s)
    np.testing.assert_allclose(got[0][np.ix_(rows, cols)], want, rtol=1e-10)
    # a target that coincides with a source point takes that point's value (vtkGaussianKernel's exact-hit rule)
    j, i = 60, 77
    tl, tn = np.array([oc['latitude'][j, i]]), np.array([oc['longitude'][j, i]])
    hit = F.gauss_interp_fields(np.zeros((1, 1)), tl, tn, oc['latitude'], oc['longitude'], [oc['values'][0]], R, s)
    assert hit[0, 0, 0] == oc['values'][0][j, i]


@pytest.mark.gpu
def test_gauss_interp_full_size_config4_ocean_leg():
    """BASELINE.json configs[3], the tos / siconc leg at production size: 12 months of an 802 x 404 curvilinear ocean grid
    (NaN over land) onto the 0.25 deg ERA5 grid (721 x 1440) with the defaults of settings.py:127-129 (R = 1000 km,
    sharpness 4, land above FR_LAND 0.7) - the properties of test_gauss_interp_properties at that size, plus strips of
    targets along the date line, around the north pole and over a land patch against the oracle (reference
    functions.py:958-1057: three copies of the planar cloud, Gaussian weights inside the radius)."""
    from pgw4era5_amd import functions as F, synthetic
    oc = synthetic.make_ocean_grid_case(nj=404, ni=802, ntime=12, seed=6, land_patches=6)
    nlat, nlon = 721, 1440
    lat, lon = _era_grid(nlat, nlon)
    land = np.zeros((nlat, nlon))
    land[400:430, 200:260] = 1.0                                               # FR_LAND > 0.7 -> NaN (functions.py:1054-1057)
    land[500:510, 900:905] = 0.69                                              # below the threshold: stays ocean
    R, s = 1.0e6, 4.0
    fields = list(oc['values'])
    fields[1] = np.where(np.isnan(oc['values'][1]), np.nan, -1.75)            # one constant month
    got = F.gauss_interp_fields(land, lat, lon, oc['latitude'], oc['longitude'], fields, R, s)
    assert got.shape == (12, nlat, nlon)
    ok = ~np.isnan(got[1])
    np.testing.assert_allclose(got[1][ok], -1.75, rtol=1e-13)                  # constant in -> constant out
    for m in (0, 5, 11):                                                       # a weighted mean stays inside the data range
        assert np.nanmin(got[m]) >= np.nanmin(oc['values'][m]) - 1e-12 and np.nanmax(got[m]) <= np.nanmax(oc['values'][m]) + 1e-12
    assert np.isnan(got[:, 400:430, 200:260]).all() and not np.isnan(got[0][500:510, 900:905]).any()
    assert np.isnan(got[0][0]).all()                                           # the south pole: farther than R from the ocean grid
    nan0 = np.isnan(got[0])
    assert all((np.isnan(got[m]) == nan0).all() for m in range(12))            # one NaN pattern (all months share the land mask)
    assert 0.85 < (~nan0).mean() < 0.999
    # strips against the oracle: both sides of the date line, the polar cap, the edge of the land patch
    checks = [(np.array([60, 200, 333, 520, 650]), np.array([717, 718, 719, 720, 721, 722, 723])),        # lon 179.25 ... 180.75
              (np.array([708, 712, 716, 719, 720]), np.array([0, 360, 719, 720, 1080, 1439])),            # 87 N ... the pole
              (np.array([398, 399, 400, 429, 430, 431]), np.array([198, 199, 200, 259, 260, 261]))]
    for n, (rows, cols) in enumerate(checks):
        sub = land[np.ix_(rows, cols)]
        for m in ((0, 7) if n == 0 else (0,)):                                 # (the oracle takes ~10 s per strip and month)
            want = O.nan_ignoring_interp(sub, lat[rows], lon[cols], oc['latitude'], oc['longitude'], fields[m], R, s)
            np.testing.assert_allclose(got[m][np.ix_(rows, cols)], want, rtol=1e-10, atol=0, equal_nan=True)
    # a target that coincides with a source point takes that point's value (vtkGaussianKernel's exact-hit rule)
    j, i = 150, 333
    hit = F.gauss_interp_fields(np.zeros((1, 1)), np.array([oc['latitude'][j, i]]), np.array([oc['longitude'][j, i]]),
                                oc['latitude'], oc['longitude'], [oc['values'][3]], R, s)
    assert hit[0, 0, 0] == oc['values'][3][j, i]


@pytest.mark.gpu
def test_step02_cli_ocean_variables(tmp_path):
    """`step_02 regridding -v tos`: ocean-grid file with 2-D latitude / longitude in, (time, lat, lon) on the ERA5 grid
    out (reference interp_wrapper, functions.py:1095-1135)."""
    from pgw4era5_amd import synthetic, ncio, step_02_preproc_deltas as s2, settings as S
    oc = synthetic.make_ocean_grid_case(nj=30, ni=44, ntime=12, seed=5)
    inp, out = tmp_path / 'gcm', tmp_path / 'regridded'
    os.makedirs(inp)
    Fd = ncio.Field
    for base in ('tos_delta.nc', 'tos_historical.nc'):
        ds = ncio.Dataset()
        ds['time'] = Fd(oc['times'], ('time',))
        ds['latitude'] = Fd(oc['latitude'], ('j', 'i'))
        ds['longitude'] = Fd(oc['longitude'], ('j', 'i'))
        ds['tos'] = Fd(oc['values'], ('time', 'j', 'i'), attrs={'units': 'K'})
        ncio.to_netcdf(ds, str(inp / base))
    lat, lon = _era_grid(19, 24)
    land = np.zeros((1, 19, 24)); land[0, 8:11, 3:6] = 1.0
    era = ncio.Dataset()
    era['lat'] = Fd(lat, ('lat',)); era['lon'] = Fd(lon, ('lon',))
    era['FR_LAND'] = Fd(land, ('time', 'lat', 'lon'))
    era['time'] = Fd(np.array([0.0]), ('time',))
    ncio.to_netcdf(era, str(tmp_path / 'era.nc'))
    done = s2.main(['regridding', '-i', str(inp), '-o', str(out), '-e', str(tmp_path / 'era.nc'), '-v', 'tos'])
    assert len(done) == 2
    res = ncio.open_dataset(str(out / 'tos_delta.nc'))
    assert res['tos'].dims == ('time', 'lat', 'lon') and res['tos'].shape == (12, 19, 24)
    np.testing.assert_array_equal(res['lat'].values, lat)
    np.testing.assert_array_equal(res['time'].values, oc['times'])
    want = O.nan_ignoring_interp(land[0], lat, lon, oc['latitude'], oc['longitude'], oc['values'][4],
                                 S.nan_interp_kernel_radius, S.nan_interp_sharpness)
    np.testing.assert_allclose(res['tos'].values[4], want, rtol=1e-10, equal_nan=True)
    assert np.isnan(res['tos'].values[:, 8:11, 3:6]).all()
