"""Generator <-> fixture consistency: re-run the three golden generators (`oracle/gen_golden*.py`) against the reference and require
every committed fixture to be reproduced BIT FOR BIT, key for key.  Needs `/root/reference` (the build container); skipped on the
GPU box, where only the fixtures travel.  Catches a generator edited without re-committing its fixtures (and vice versa)."""
import os
import subprocess
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
GOLDEN = os.path.join(ROOT, 'tests', 'golden')
REF = os.environ.get('VKN_REFERENCE', '/root/reference')
GENERATORS = ('gen_golden.py', 'gen_golden_vis.py', 'gen_golden_tracker.py')


@pytest.fixture(scope='module')
def regenerated(tmp_path_factory):
    if not os.path.isdir(os.path.join(REF, 'knet')):
        pytest.skip('the reference tree is not present (GPU box): fixtures cannot be regenerated here')
    out = tmp_path_factory.mktemp('golden_regen')
    env = dict(os.environ, VKN_GOLDEN_OUT=str(out), VKN_REFERENCE=REF, PYTHONDONTWRITEBYTECODE='1')
    for g in GENERATORS:
        r = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', g)], env=env, capture_output=True, text=True, timeout=900)
        assert r.returncode == 0, f'{g} failed:\n{r.stdout[-2000:]}\n{r.stderr[-2000:]}'
    return str(out)


def _same(a, b):
    if a.dtype != b.dtype or a.shape != b.shape:
        return False
    if a.dtype.kind in 'fc':
        return np.array_equal(a, b, equal_nan=True)
    return np.array_equal(a, b)


def test_every_fixture_is_reproduced_bit_for_bit(regenerated):
    committed = sorted(f for f in os.listdir(GOLDEN) if f.endswith('.npz'))
    fresh = sorted(f for f in os.listdir(regenerated) if f.endswith('.npz'))
    assert committed == fresh, f'fixtures without a generator: {sorted(set(committed) - set(fresh))}; ' \
                               f'generated but not committed: {sorted(set(fresh) - set(committed))}'
    bad = []
    for f in committed:
        a, b = np.load(os.path.join(GOLDEN, f)), np.load(os.path.join(regenerated, f))
        if sorted(a.files) != sorted(b.files):
            bad.append(f'{f}: keys differ (committed - fresh = {sorted(set(a.files) - set(b.files))}, '
                       f'fresh - committed = {sorted(set(b.files) - set(a.files))})')
            continue
        bad += [f'{f}[{k}]' for k in a.files if not _same(a[k], b[k])]
    assert not bad, 'generator and fixtures are out of sync:\n  ' + '\n  '.join(bad)
