"""Numeric constants of the delay path (values of tools/RAiDER/constants.py:12-23)."""
import numpy as np

# heights (m)
_ZMIN, _ZREF = np.float64(-100), np.float64(26000)     # lowest required model height; default top of the integration
_STEP = np.float64(15.0)                               # fixed step of the legacy makePoints ray sampler
_CUBE_SPACING_IN_M = float(2000)                       # default horizontal posting of output cubes

# Earth
R_EARTH_MAX_WGS84, R_EARTH_MIN_WGS84 = 6378137, 6356752
_RE = np.float64(6371008.7714)
_g0, _g1 = np.float64(9.80665), np.float64(9.80616)   # standard gravity; gravity at 45 deg latitude

_THRESHOLD_SECONDS = 60                                 # temporal-interpolation tolerance
