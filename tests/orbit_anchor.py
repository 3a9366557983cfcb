"""Anchors for the zero-Doppler solver that neither the builder's oracle nor its kernel had a hand in (isce3 is absent):

* a CIRCULAR orbit in the equatorial plane, S(t) = r (cos wt, sin wt, 0), V = dS/dt (the geometry of the reference's
  test/fake_raytracing:73-111, with velocities consistent with the positions).  For ANY target T the zero-Doppler condition
  (S - T) . V = 0 reduces to T . V = 0 (S . V = 0 on a circle), i.e. sin(lon_T - w t) = 0: the azimuth time is EXACTLY lon_T / w,
  the sensor is at the target's longitude, slant range^2 = r^2 + |T|^2 - 2 r |T| cos(geocentric latitude), and the look vector
  follows in closed form.  What remains between that and the solver is the 4-point Hermite interpolation error of a circle
  sampled every 10 s: <= (w dt)^4 / 384 r = 2e-4 m.
* the eight Sentinel-1 state vectors the reference's own test suite carries (test/test_losreader.py:20-92, from
  test/orbit_files/S1_orbit_example.EOF): Hermite interpolation through every OTHER vector must land on the skipped ones.
"""
import numpy as np

W = 2 * np.pi / 5900.0            # rad/s: a 98-minute orbit
RS = 6378137.0 + 700000.0         # test/fake_raytracing:82,89


def circular_orbit(n=41, dt=10.0):
    t = dt * np.arange(n)
    pos = np.stack([RS * np.cos(W * t), RS * np.sin(W * t), np.zeros(n)], -1)
    vel = np.stack([-RS * W * np.sin(W * t), RS * W * np.cos(W * t), np.zeros(n)], -1)
    return t, pos, vel


def targets(rng, n=500, t_lo=60.0, t_hi=340.0):
    """Targets on / near the ellipsoid whose closed-form azimuth time lies inside the orbit arc; either look side."""
    lon = W * rng.uniform(t_lo, t_hi, n)
    lat = np.radians(rng.uniform(2.0, 9.0, n)) * rng.choice([-1.0, 1.0], n)
    rad = 6378137.0 * (1 - 0.00335 * np.sin(lat) ** 2) + rng.uniform(-100.0, 9000.0, n)        # geocentric radius
    T = np.stack([rad * np.cos(lat) * np.cos(lon), rad * np.cos(lat) * np.sin(lon), rad * np.sin(lat)], -1)
    t0 = lon / W
    S = np.stack([RS * np.cos(lon), RS * np.sin(lon), np.zeros(n)], -1)
    rg = np.sqrt(RS ** 2 + rad ** 2 - 2 * RS * rad * np.cos(lat))
    los = (S - T) / rg[:, None]
    return T, t0, rg, los


# test/test_losreader.py:20-92 (seconds after 2018-11-12T23:00:02)
S1_T = np.arange(8) * 10.0
S1_POS = np.array([[-2064965.285362, 6434865.494987, 2090670.967443], [-2056228.553736, 6460407.492520, 2019650.417312],
                   [-2047224.526705, 6485212.031660, 1948401.684024], [-2037955.293282, 6509275.946120, 1876932.818066],
                   [-2028422.977002, 6532596.156540, 1805251.894958], [-2018629.735564, 6555169.670917, 1733367.014327],
                   [-2008577.760461, 6576993.585012, 1661286.298987], [-1998269.276601, 6598065.082739, 1589017.893976]])
S1_VEL = np.array([[860.239634, 2590.964968, -7090.378144], [887.072466, 2517.380329, -7113.598127], [913.698134, 2443.474728, -7136.014344],
                   [940.113169, 2369.256838, -7157.624244], [966.314136, 2294.735374, -7178.425371], [992.297636, 2219.919093, -7198.415359],
                   [1018.060311, 2144.816789, -7217.591940], [1043.598837, 2069.437298, -7235.952940]])
