"""CPU: tools/sumo_runner.py -- the SUMO side of the comparison, which cannot run where SUMO is absent -- exercised end to
end against a stand-in for `libsumo`: the TraCI subset it calls (program install through getAllProgramLogics /
setProgramLogic, getPhase / setPhase, simulationStep, lane.getLastStepVehicleNumber, vehicletype setters, start / close)
served by the CPU oracle through oracle/ref_harness.FakeSumo, next to the host emulation of the kernel.  Because both sides
then run the same model, every phase vector and every lane count must agree: what is tested is the runner's own restatement of
the reference's step loop (prep -> yellow ticks -> set -> green ticks), its yellow-programme install, its lane bookkeeping and
its comparison -- so that the day it meets a real SUMO, a difference it reports is SUMO's."""
import os
import sys
import types

import numpy as np
import pytest

from conftest import ROOT, load_scenario

sys.path.insert(0, ROOT)


def fake_libsumo(sc, seed):
    from oracle.pyoracle import OracleEnv
    from oracle.ref_harness import FakeSumo, _Phase
    mod = types.ModuleType('libsumo')
    state = {}

    def start(cmd, **kw):
        assert '--time-to-teleport' in cmd and '-1' in cmd and '--seed' in cmd     # multi_signal.py:127-131
        orc = OracleEnv(sc, env_index=0, seed=seed, sigma=0.0, speed_dev=0, max_distance=1.0e9)
        fs = FakeSumo(sc, orc)
        state['fs'] = fs
        mod.trafficlight = fs.trafficlight
        mod.trafficlight.Phase = _Phase
        mod.lane = fs.lane
        mod.lane.getLastStepVehicleNumber = lambda lane: len(fs._lane_vehicles(lane)) if lane in fs.lane_index else (_ for _ in ()).throw(KeyError(lane))
        mod.simulationStep = fs.simulationStep

    mod.start = start
    mod.close = lambda *a, **k: None
    mod.vehicletype = types.SimpleNamespace(getIDList=lambda: ('t',), setImperfection=lambda vt, x: None,
                                            setSpeedDeviation=lambda vt, x: None)
    return mod


@pytest.mark.parametrize('name,steps', [('cologne1', 60), ('cologne8', 40), ('ingolstadt7', 40)])
def test_runner_reports_no_difference_when_both_sides_run_the_same_model(name, steps, monkeypatch):
    from tools import sumo_runner
    from hostemu.emu import EmuSim
    sc = load_scenario(name)
    mod = fake_libsumo(sc, seed=3)
    monkeypatch.setattr(sumo_runner, 'find_sumo', lambda: ('libsumo', mod, 'sumo'))
    monkeypatch.setattr(sumo_runner, 'find_sumocfg', lambda m: os.path.join('/nonexistent', m + '.sumocfg'))
    r = sumo_runner.diff_vs_sumo(name, steps=steps, seed=3, sim_cls=EmuSim)
    assert r['phase_vectors_equal'] == steps
    assert r['lane_counts_compared'] > 0 and r['lane_counts_equal'] == r['lane_counts_compared'] and r['mean_abs_count_diff'] == 0.0
    # traffic did build up: the comparison was not about empty lanes
    assert any(row['lanes_differing'] == 0 for row in r['per_step'])


def test_runner_says_what_is_missing_without_sumo():
    from tools import sumo_runner
    kind, api, binary = sumo_runner.find_sumo()
    if api is None:
        msg = sumo_runner.sumo_baseline('cologne1', 0.1)
        assert isinstance(msg, str) and 'SUMO' in msg
