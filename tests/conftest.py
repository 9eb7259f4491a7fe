import os
import sys

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)

GOLDEN = os.path.join(ROOT, 'tests', 'golden')
HOT_CASES = ['cologne1_d200', 'cologne8_d200', 'cologne8_d50', 'ingolstadt21_d200', 'cologne3_d200',
             'ingolstadt7_d200', 'ingolstadt1_d200']
# the reference driven on a LOADED network (180 env-steps of pre-roll before its Signal objects are built) and through a
# whole 360-step episode (tests/golden/make_golden.py)
WARM_CASES = ['ingolstadt21_d200_warm180', 'cologne8_d200_warm180']
FULL_CASES = ['cologne1_d50_full']
# MultiSignal(step_ratio=2): two simulation steps per step_sim() (multi_signal.py:102-105), driven by the reference's own loop
RATIO_CASES = ['cologne8_d200_sr2']
# tls_expiry = 1 (rs_params.tls_hold = 0: a phase set through setPhase runs out after its programme duration -- the library's default since
# round 6; the cases WITHOUT the suffix were generated with tls_expiry = 0, round 5's default, and say so in their meta data)
EXPIRY_CASES = ['cologne1_d200_exp', 'ingolstadt21_d200_exp', 'cologne8_d200_sr2_exp']
ALL_CASES = HOT_CASES + WARM_CASES + FULL_CASES + RATIO_CASES + EXPIRY_CASES


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


def preroll_actions(sc, seed, env_index, k):
    """the on-device random policy (rs_act_random): murmur(seed ^ 0xA5A5A5A5; env, signal, step, 7) % n_green"""
    import numpy as np
    from oracle.pyoracle import lib
    L = lib()
    return np.array([L.orc_hash((seed ^ 0xA5A5A5A5) & 0xFFFFFFFF, env_index, s, k, 7) % int(sc.tls_ngreen[s])
                     for s in range(sc.n_signals)], np.int32)


def load_golden(tag):
    import json

    import numpy as np
    with open(os.path.join(GOLDEN, tag + '.json')) as f:
        meta = json.load(f)
    arr = dict(np.load(os.path.join(GOLDEN, tag + '.npz')))
    return meta, arr
