"""LIVE check of the oracle against the unmodified reference, imported in place - only where /root/reference exists (the build
container); skipped on the GPU box, where the committed goldens made the same way stand in.  The check runs in a subprocess
(oracle/refharness/live_check.py) so that the stand-in packages the reference needs never enter this test process."""
import json
import subprocess
import sys
from pathlib import Path

import pytest

ROOT = Path(__file__).resolve().parent.parent
HAVE = Path('/root/reference/tools/RAiDER').is_dir() and any((ROOT / 'oracle' / '_ref' / 'RAiDER').glob('interpolate*.so'))


@pytest.mark.skipif(not HAVE, reason='the reference tree / its compiled extensions are not here')
def test_oracle_against_the_live_reference():
    """Fresh random ray-traced scenes, a zenith cube with outside nodes, the native interpolator and makePoints: the oracle vs what
    the reference itself returns right now (delays to 1e-11 m, zenith to 1e-15 relative, natives bit for bit)."""
    out = subprocess.run([sys.executable, str(ROOT / 'oracle' / 'refharness' / 'live_check.py')], capture_output=True, text=True, timeout=600)
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['ray_max_abs_m'] < 1e-11 and res['zenith_max_rel'] < 1e-15 and res['natives_bit_exact'] and res['makepoints_bit_exact'], res
