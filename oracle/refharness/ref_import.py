"""Import the read-only reference (`/root/reference/tools/RAiDER`) IN PLACE, in this container only.

TEST INFRASTRUCTURE.  Used by oracle/refharness/gen_golden.py (fixture generation) and by the
optional `tests/test_oracle_vs_reference.py` (skipped when /root/reference is absent, i.e. on the GPU
box).  Recipe = SURVEY.md App. B:
  1. pre-seed sys.modules['RAiDER'] with a bare package whose __path__ points at the reference dir
     (+ oracle/_ref/RAiDER for the two compiled native extensions) - bypasses
     tools/RAiDER/__init__.py:7-10 (importlib.metadata.version of an uninstalled dist);
  2. put the build-owned stubs of pyproj / xarray / rasterio (absent from this image) at the END of sys.path - real packages win;
  3. point the reference logger at a temp dir (logger.py:59-86 would write debug.log into CWD).
"""
import os
import sys
import tempfile
import types
from pathlib import Path

HERE = Path(__file__).resolve().parent
REF_ROOT = Path(os.environ.get('RAIDER_REFERENCE', '/root/reference'))
REF_PKG = REF_ROOT / 'tools' / 'RAiDER'
REF_SO = HERE.parent / '_ref' / 'RAiDER'


def available() -> bool:
    return REF_PKG.is_dir()


def import_reference():
    """Returns the seeded `RAiDER` package object (reference modules importable as RAiDER.xxx)."""
    if not available():
        raise RuntimeError(f'reference not present at {REF_PKG}')
    if 'RAiDER' in sys.modules and getattr(sys.modules['RAiDER'], '_oracle_seeded', False):
        return sys.modules['RAiDER']
    # The stand-ins go to the END of sys.path: a REAL pyproj / xarray / rasterio / isce3, the day the image has one, is found first and
    # used instead (provenance() then says so, and every fixture and the live-check record carry that string)
    stubs = str(HERE / 'stubs')
    if stubs not in sys.path:
        sys.path.append(stubs)
    pkg = types.ModuleType('RAiDER')
    pkg.__path__ = [str(REF_PKG), str(REF_SO)]
    pkg.__version__ = '0.0-reference-in-place'
    pkg._oracle_seeded = True
    sys.modules['RAiDER'] = pkg
    import RAiDER.cli.conf as conf  # noqa: E402
    conf.setLoggerPath(Path(tempfile.mkdtemp(prefix='raider_ref_log_')))
    # RAiDER.models/__init__.py imports every provider (herbie, cdsapi, ...): seed a bare sub-package so that
    # RAiDER.models.weatherModel alone can be imported (needs only the shapely import stub)
    models = types.ModuleType('RAiDER.models')
    models.__path__ = [str(REF_PKG / 'models')]
    sys.modules['RAiDER.models'] = models
    return pkg


def provenance():
    """What the reference's third-party geometry ran on when a fixture was made: {'geodesy': ..., 'look_vectors': ...}.
    geodesy: 'pyproj <ver> / PROJ <ver>' when the real pyproj is importable, else the builder's stand-in (WGS84 geodetic <-> ECEF
    restated from PROJ's `cart` conversion: parity of THAT arithmetic is then unpinned, everything downstream is the reference's code).
    look_vectors: 'isce3 <ver>' when isce3 is importable (Raytracing.getLookVectors = geo2rdr + orbit.interpolate, losreader.py:219-255),
    else 'absent' (golden g14 cannot be generated; orbit look vectors stay pinned only by closed-form orbits)."""
    stubs = str(HERE / 'stubs')
    out = {}
    try:
        import pyproj
        real = not str(getattr(pyproj, '__file__', '')).startswith(stubs)
        if real:
            out['geodesy'] = f"pyproj {pyproj.__version__} / PROJ {getattr(pyproj, 'proj_version_str', getattr(pyproj, '__proj_version__', '?'))}"
        else:
            out['geodesy'] = 'builder stub oracle/refharness/stubs/pyproj (WGS84 cart conversion restated; PROJ binary parity unpinned)'
    except ImportError:
        out['geodesy'] = 'no pyproj at all'
    try:
        import isce3
        out['look_vectors'] = f"isce3 {getattr(isce3, '__version__', '?')}"
    except ImportError:
        out['look_vectors'] = 'absent (isce3 not importable: golden g14 not generated)'
    return out
