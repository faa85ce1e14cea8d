import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: test needs a real MI355X (run with -m gpu)')


@pytest.fixture(scope='session')
def golden():
    import json
    import numpy as np
    z = np.load(os.path.join(GOLDEN, 'ref_leaf_vectors.npz'), allow_pickle=False)
    with open(os.path.join(GOLDEN, 'ref_leaf_vectors.json')) as f:
        meta = json.load(f)
    return dict(z), meta


@pytest.fixture(scope='session')
def golden_harmonic():
    """Outputs of the reference's harmonic_ac_analysis (oracle/make_golden.py harmonic)."""
    import json
    import numpy as np
    z = np.load(os.path.join(GOLDEN, 'ref_harmonic_vectors.npz'), allow_pickle=False)
    with open(os.path.join(GOLDEN, 'ref_harmonic_vectors.json')) as f:
        meta = json.load(f)
    return dict(z), meta
