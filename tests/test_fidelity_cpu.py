"""CPU (not gpu): the oracle's microsimulation against the reference-held result figures (tests/golden/ref_bands.json, see
tests/golden/make_ref_bands.py) on the two single-intersection maps -- the same bands tests/test_gpu_parity.py::
test_reference_result_bands asserts for the HIP path on all six maps with 64 environments."""
import json
import os

import pytest

from conftest import ROOT

from oracle import fidelity_eval


@pytest.mark.parametrize('name', ['cologne1', 'ingolstadt1'])
def test_oracle_inside_the_reference_bands(name):
    with open(os.path.join(ROOT, 'tests', 'golden', 'ref_bands.json')) as f:
        ref = json.load(f)[name]
    res = {}
    for policy in ('FIXED', 'MAXPRESSURE'):
        res[policy] = fidelity_eval.run(name, policy, envs=3, seed=0, steps=360)
        ratio = res[policy]['delay'] / ref[policy]['delay']
        assert 0.65 <= ratio <= 1.35, (name, policy, res[policy]['delay'], ref[policy]['delay'])
    # travel time of the routes at the speed limits: duration - (timeLoss + departDelay) under a controller that keeps the map fluid
    resid = res['MAXPRESSURE']['duration'] - res['MAXPRESSURE']['delay']
    assert 0.9 <= resid / ref['free_flow_residual'] <= 1.1, (name, resid, ref['free_flow_residual'])
