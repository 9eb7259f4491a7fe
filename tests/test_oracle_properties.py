"""CPU: invariants of the microsimulation model (no SUMO trace exists to diff against)."""
import numpy as np
import pytest

from conftest import load_scenario
from oracle.pyoracle import OracleEnv


@pytest.mark.parametrize('name,fixed,steps', [('cologne1', 0, 120), ('cologne8', 1, 90), ('ingolstadt21', 0, 40),
                                              ('cologne3', 0, 60), ('ingolstadt1', 1, 60), ('ingolstadt7', 0, 50)])
def test_invariants(name, fixed, steps):
    sc = load_scenario(name)
    A = sc.arrays
    env = OracleEnv(sc, env_index=5, seed=11, sigma=-1.0, speed_dev=1, fixed_program=fixed)
    rng = np.random.default_rng(5)
    prev_cursor = {}
    for k in range(steps):
        env.step(np.array([rng.integers(0, g) for g in sc.tls_ngreen], np.int32))
        v = env.vehicles()
        st = env.stats()
        hw = v['hw']
        lane, pos, spd, trip = v['lane'][:hw], v['pos'][:hw], v['speed'][:hw], v['trip'][:hw]
        act = lane < 0xFFFE
        # conservation: every inserted trip is active or arrived; the backlog is what has departed but is not inserted
        assert st['inserted'] == st['arrived'] + st['active']
        assert v['next_trip'] == st['inserted']
        waited, n_back = env.backlog_delay()
        t_now = env.time
        assert n_back == int((A['trip_depart'] < t_now).sum()) - st['inserted'] and waited >= 0
        assert st['pending'] == int((A['trip_depart'] <= t_now - 1).sum()) - st['inserted']     # tried in the last tick or before
        assert int(act.sum()) == st['active']
        # kinematic bounds
        assert (spd[act] >= 0).all() and (pos[act] >= 0).all()
        assert (pos[act] <= A['lane_len'][lane[act]] + 1e-3).all()
        vt = A['vtype_params'][A['trip_vtype'][trip[act]]]
        assert (spd[act] <= np.minimum(vt[:, 6], A['lane_vmax'].max() * 2.0) + 1e-3).all()
        # trips are unique, cursors never move backwards
        assert len(set(trip[act].tolist())) == int(act.sum())
        for t_, c_ in zip(trip[act].tolist(), v['cursor'][:hw][act].tolist()):
            assert c_ >= prev_cursor.get(t_, 0)
            prev_cursor[t_] = c_
        # no two vehicles of one lane overlap by more than a vehicle length (merges may touch)
        for ln in np.unique(lane[act]):
            idx = np.nonzero(act & (lane == ln))[0]
            if len(idx) < 2:
                continue
            order = idx[np.argsort(pos[idx])]
            lens = A['vtype_params'][A['trip_vtype'][trip[order]], 0]
            gaps = pos[order][1:] - lens[1:] - pos[order][:-1]
            assert (gaps > -lens[1:]).all()
        assert env.time == (k + 1) * 10


def test_parity_mode_is_deterministic_and_env_keyed():
    sc = load_scenario('cologne1')
    acts = np.random.default_rng(0).integers(0, 4, (30, 1)).astype(np.int32)

    def run(env_index, sigma):
        e = OracleEnv(sc, env_index=env_index, seed=3, sigma=sigma, speed_dev=1 if sigma != 0 else 0)
        for a in acts:
            e.step(a)
        return e.vehicles()['pos'].copy(), e.outputs()['lane_agg'].copy()

    p0, a0 = run(0, 0.0)
    p1, a1 = run(9, 0.0)
    np.testing.assert_array_equal(p0, p1)          # sigma = 0, speedFactor = 1: the RNG is never consulted
    q0, _ = run(0, 0.5)
    q1, _ = run(9, 0.5)
    q0b, _ = run(0, 0.5)
    np.testing.assert_array_equal(q0, q0b)
    assert not np.array_equal(q0, q1)              # environments decorrelate through the env index


def test_fsm_duration_auto_advance():
    """SURVEY 8(a) A7 [SUMO-K], the default (rs_params.tls_hold = 0): a 6 s green expires inside the 7 post-yellow ticks -> phase advances
    to (a+1) % P."""
    sc = load_scenario('cologne1')
    e = OracleEnv(sc, sigma=0.0)
    e.observe()
    assert e.outputs()['phase'][0] == 0
    e.step(np.array([1], np.int32))                # green 1 has duration 6
    assert e.outputs()['phase'][0] == 2
    e.step(np.array([2], np.int32))                # same phase requested: no yellow, restart at set_phase
    assert e.outputs()['phase'][0] == 2
    e.step(np.array([0], np.int32))                # 29 s green: stays
    assert e.outputs()['phase'][0] == 0
    e.step(np.array([3], np.int32))                # 6 s green, last green: wraps into the first yellow (index 4)
    assert e.outputs()['phase'][0] == 4
    e.step(np.array([0], np.int32))                # current phase is a yellow index: no yellow_dict key, plain set
    assert e.outputs()['phase'][0] == 0


def test_fsm_tls_hold_keeps_the_selected_phase():
    """rs_params.tls_hold = 1 (tls_expiry=0, round 5's default): the phase entered through setPhase is the phase observed, whatever its
    programme duration; the phase installed at reset still runs on its duration until the first setPhase, and so does the net's own
    programme (fixed_program)"""
    sc = load_scenario('cologne1')
    e = OracleEnv(sc, sigma=0.0, tls_expiry=0)
    e.observe()
    for a in (1, 3, 1, 0, 3, 3, 2):
        e.step(np.array([a], np.int32))
        assert e.outputs()['phase'][0] == a
    e = OracleEnv(sc, sigma=0.0)                   # nobody calls setPhase: the installed programme runs (29 s, then the 6 s green ...)
    for _ in range(29):
        e.tick()
    assert e.get_phase(0) == 0
    e.tick()
    assert e.get_phase(0) == 1
    for _ in range(6):
        e.tick()
    assert e.get_phase(0) == 2
    f = OracleEnv(sc, sigma=0.0, fixed_program=1)
    seen = set()
    for _ in range(90):
        f.tick()
        seen.add(f.get_phase(0))
    assert seen == set(range(8))                   # the net's eight phases, one 90 s cycle
