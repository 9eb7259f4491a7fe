import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
HOT_CASES = ['cologne1_d200', 'cologne8_d200', 'cologne8_d50', 'ingolstadt21_d200', 'cologne3_d200',
             'ingolstadt7_d200', 'ingolstadt1_d200']


def pytest_configure(config):
    config.addinivalue_line('markers', 'gpu: needs a real MI355X (run with -m gpu on the GPU box)')


def pytest_collection_modifyitems(config, items):
    """A kernel that never returns (e.g. a barrier reached by part of a workgroup) blocks inside
    hipStreamSynchronize, where no Python-level timeout can fire: GPU tests get a hard per-test limit enforced
    from a watchdog thread that ends the process (pytest-timeout, method 'thread')."""
    try:
        import pytest_timeout  # noqa: F401
    except ImportError:
        return
    for item in items:
        if item.get_closest_marker('gpu') and not item.get_closest_marker('timeout'):
            item.add_marker(pytest.mark.timeout(180, method='thread'))


@pytest.fixture(scope='session')
def root():
    return ROOT


def load_scenario(name):
    from resco_amd.scenario import Scenario
    return Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))


def load_golden(tag):
    import json

    import numpy as np
    with open(os.path.join(GOLDEN, tag + '.json')) as f:
        meta = json.load(f)
    arr = dict(np.load(os.path.join(GOLDEN, tag + '.npz')))
    return meta, arr
