import json
import os
import sys

import numpy as np
import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)
GOLDEN = os.path.join(ROOT, 'tests', 'golden')


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


@pytest.fixture(scope='session')
def known():
    with open(os.path.join(GOLDEN, 'reference_known_answers.json')) as f:
        return json.load(f)


@pytest.fixture(scope='session')
def golden_dir():
    return GOLDEN


@pytest.fixture(scope='session')
def oracle_built():
    """Build oracle/liboracle.so if it is not there yet (gcc, seconds)."""
    so = os.path.join(ROOT, 'oracle', 'liboracle.so')
    if not os.path.exists(so):
        import subprocess
        subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle')])
    return so


@pytest.fixture(scope='session')
def hip():
    """Initialised HIP library; GPU tests must FAIL (not skip) without it."""
    from cellcomm_amd import _lib
    return _lib.ensure_init(0)


def load_npz(name):
    return np.load(os.path.join(GOLDEN, name), allow_pickle=False)
