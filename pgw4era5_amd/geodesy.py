"""
WGS84 geodesic lengths for the planar "metre" coordinates of the NaN-ignoring SST / sea-ice interpolation
(reference functions.py:958-975, 1010-1023: three `pyproj.Geod(ellps="WGS84").inv` calls per point).

The reference maps every (lon, lat) to
    lat_m = sign(lat) * |geodesic (lon, 0) -> (lon, lat)|                      the meridian arc
    lon_m = sign(lon) * |geodesic (0, lat) -> (lon, lat)|                      two points on ONE parallel
    lon_offset = |geodesic (0, lat) -> (180, lat)|                             over the pole: 2 (Q - meridian arc)
pyproj (Karney's algorithm) is not installable in the build container, so the three lengths are computed here
from Vincenty's series (Survey Review XXII, 176, 1975; truncation error < 0.1 mm on the WGS84 ellipsoid), arranged so that
no iteration can fail:
  * meridian arc and over-the-pole length: the direct series with azimuth 0;
  * two points of equal latitude: the geodesic between them is symmetric about the meridian half way, so it is the
    direct problem "leave (lat) with azimuth a1, travel until the latitude is reached again"; its longitude
    difference L(a1) falls monotonically from 180 deg (a1 = 0, over the pole) to 0 (a1 = 90 deg), and a1 is found by
    bisection (Vincenty's INVERSE iteration does not converge for nearly antipodal points, e.g. the ERA5 grid point
    lon = 180 on the equator row).  On the equator the geodesic is the equator itself up to L = (1 - f) 180 deg.
Host-side numpy: this is grid geometry, computed once per (source grid, target grid) pair and reused for all months.
"""
import numpy as np

WGS84_A = 6378137.0
WGS84_F = 1.0 / 298.257223563
WGS84_B = WGS84_A * (1.0 - WGS84_F)


def _series(cos2_alpha):
    """Vincenty's A, B (distance) and C (longitude) coefficients for a geodesic with equatorial azimuth alpha."""
    u2 = cos2_alpha * (WGS84_A ** 2 - WGS84_B ** 2) / WGS84_B ** 2
    A = 1 + u2 / 16384 * (4096 + u2 * (-768 + u2 * (320 - 175 * u2)))
    B = u2 / 1024 * (256 + u2 * (-128 + u2 * (74 - 47 * u2)))
    C = WGS84_F / 16 * cos2_alpha * (4 + WGS84_F * (4 - 3 * cos2_alpha))
    return A, B, C


def _arc(sigma, cos_2sm, A, B):
    """Length of an arc of angular size sigma on the auxiliary sphere with mid-point argument 2 sigma_m."""
    sin_s, cos_s = np.sin(sigma), np.cos(sigma)
    dsig = B * sin_s * (cos_2sm + B / 4 * (cos_s * (-1 + 2 * cos_2sm ** 2) -
                                           B / 6 * cos_2sm * (-3 + 4 * sin_s ** 2) * (-3 + 4 * cos_2sm ** 2)))
    return WGS84_B * A * (sigma - dsig)


def reduced_latitude(lat_deg):
    return np.arctan((1.0 - WGS84_F) * np.tan(np.deg2rad(np.asarray(lat_deg, dtype=np.float64))))


def meridian_arc(lat_deg):
    """|geodesic| from the equator to latitude `lat_deg` along a meridian [m], >= 0."""
    U = np.abs(reduced_latitude(lat_deg))
    U = np.where(np.abs(np.asarray(lat_deg, dtype=np.float64)) >= 90.0, 0.5 * np.pi, U)
    A, B, _ = _series(1.0)                              # azimuth 0: cos^2(alpha) = 1
    return _arc(U, np.cos(U), A, B)                     # sigma_1 = 0, sigma = U, 2 sigma_m = U


QUARTER_MERIDIAN = float(meridian_arc(90.0))


def over_pole(lat_deg):
    """|geodesic| between (0, lat) and (180, lat): along the two meridians over the nearer pole."""
    return 2.0 * (QUARTER_MERIDIAN - meridian_arc(lat_deg))


def _symmetric(U, a1):
    """Leave reduced latitude U (>= 0) with azimuth a1 in [0, pi/2], travel until latitude U is reached again:
    (longitude difference [rad], length [m])."""
    sinU, cosU = np.sin(U), np.cos(U)
    sin_a1, cos_a1 = np.sin(a1), np.cos(a1)
    s1 = np.arctan2(sinU, cosU * cos_a1)                # tan(sigma_1) = tan(U) / cos(a1)
    sigma = np.pi - 2.0 * s1
    sin_alpha = cosU * sin_a1
    cos2_alpha = 1.0 - sin_alpha ** 2
    A, B, C = _series(cos2_alpha)
    sin_s, cos_s = np.sin(sigma), np.cos(sigma)
    omega = np.arctan2(sin_s * sin_a1, cosU * cos_s - sinU * sin_s * cos_a1)
    omega = np.where(omega < 0, omega + 2 * np.pi, omega)
    cos_2sm = -np.ones_like(sigma)                      # 2 sigma_m = 2 sigma_1 + sigma = pi
    L = omega - (1 - C) * WGS84_F * sin_alpha * (sigma + C * sin_s * (cos_2sm + C * cos_s * (-1 + 2 * cos_2sm ** 2)))
    return L, _arc(sigma, cos_2sm, A, B)


def same_latitude_geodesic(lat_deg, dlon_deg, iterations=70):
    """|geodesic| between (0, lat) and (dlon, lat) [m]; |dlon| <= 180."""
    lat = np.asarray(lat_deg, dtype=np.float64)
    lat, dl = np.broadcast_arrays(lat, np.abs(np.asarray(dlon_deg, dtype=np.float64)))
    U = np.abs(reduced_latitude(lat))
    Ls = np.deg2rad(dl)
    lo = np.zeros_like(Ls)                              # L(lo) = pi  >= L*
    hi = np.full_like(Ls, 0.5 * np.pi)                  # L(hi) = 0 (on the equator: (1 - f) pi)  <= L*
    for _ in range(iterations):
        mid = 0.5 * (lo + hi)
        L, _s = _symmetric(U, mid)
        big = L > Ls
        lo = np.where(big, mid, lo)
        hi = np.where(big, hi, mid)
    _, s = _symmetric(U, 0.5 * (lo + hi))
    # on (or within rounding of) the equator the geodesic is the equator itself up to L = (1 - f) pi
    equator = (U < 1e-15) & (Ls <= (1.0 - WGS84_F) * np.pi)
    s = np.where(equator, WGS84_A * Ls, s)
    s = np.where(dl >= 180.0, over_pole(lat), s)
    s = np.where(np.abs(lat) >= 90.0, 0.0, s)
    return np.where(dl == 0.0, 0.0, s)


def planar_metres(lat_deg, lon_deg):
    """The reference's point-cloud coordinates (functions.py:958-975): lon folded to (-180, 180] by the caller.
    Returns (lat_m, lon_m, lon_offset)."""
    lat = np.asarray(lat_deg, dtype=np.float64)
    lon = np.asarray(lon_deg, dtype=np.float64)
    lat_m = meridian_arc(lat) * np.sign(lat)
    lon_m = same_latitude_geodesic(lat, lon) * np.sign(lon)
    return lat_m, lon_m, over_pole(lat)
