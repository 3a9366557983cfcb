"""Hot-path constants (tools/RAiDER/constants.py:12-23)."""
import numpy as np

_ZMIN = np.float64(-100)     # minimum required height
_ZREF = np.float64(26000)    # default maximum integration height
_STEP = np.float64(15.0)     # legacy fixed integration step (makePoints)
_g0 = np.float64(9.80665)
_g1 = np.float64(9.80616)
_RE = np.float64(6371008.7714)
R_EARTH_MAX_WGS84 = 6378137
R_EARTH_MIN_WGS84 = 6356752
_CUBE_SPACING_IN_M = float(2000)
_THRESHOLD_SECONDS = 1 * 60
