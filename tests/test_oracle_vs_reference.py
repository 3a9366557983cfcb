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
    assert res['reversed_segment_max_abs_m'] < 1e-11, res        # (round 6) an origin above zref inside zref's model interval: one reversed segment, positive length


OLD_PY = '/opt/conda/bin/python3.9'


@pytest.mark.skipif(not Path('/root/reference/tools/RAiDER').is_dir() or not Path(OLD_PY).exists(), reason='the reference tree / a second interpreter are not here')
def test_the_reference_in_its_own_generation_of_libraries_gives_the_same_numbers():
    """The reference pins numpy < 2 (environment.yml) and no scipy; the goldens were made under numpy 2.2 / scipy 1.15.  The image's
    Anaconda interpreter (python 3.9, numpy 1.26, scipy 1.7.1 - the generation of libraries a RAiDER installation has) runs the SAME live
    check of the reference's Python path against the oracle: delays to 1e-14 m (observed: one ulp), zenith bit for bit.  So neither the
    goldens nor the parity target depend on which side of the numpy-2 / compiled-RGI changes an installation is."""
    import os
    env = {k: v for k, v in os.environ.items() if not k.startswith('PYTHON')}
    out = subprocess.run([OLD_PY, '-W', 'ignore', str(ROOT / 'oracle' / 'refharness' / 'live_check.py')], capture_output=True, text=True, timeout=900, env=env)
    if out.returncode != 0 and ('ModuleNotFoundError' in out.stderr or 'ImportError' in out.stderr):
        pytest.skip('the second interpreter lacks a module: ' + out.stderr.strip().splitlines()[-1])
    assert out.returncode == 0, out.stderr[-3000:]
    res = json.loads([ln for ln in out.stdout.splitlines() if ln.startswith('{')][-1])
    assert res['numpy'].startswith('1.') and res['ray_max_abs_m'] < 1e-14 and res['zenith_max_rel'] < 1e-15 and res['reversed_segment_max_abs_m'] < 1e-14, res
