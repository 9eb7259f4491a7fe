"""CPU: the C oracle's observe / state / reward / FSM vs the REFERENCE's own Python.

The golden fixtures were produced by the reference's unmodified MultiSignal + Signal + states + rewards
running over a FakeSumo backed by this same oracle's dynamics (tests/golden/make_golden.py), so every value
compared here went through the reference's Python on one side and through oracle/resco_oracle.c's
restatement on the other."""
import numpy as np
import pytest

from conftest import ALL_CASES, load_golden, load_scenario, preroll_actions
from oracle.pyoracle import OracleEnv


def flat_sig(sc, per_obs):
    return per_obs


@pytest.mark.parametrize('tag', ALL_CASES)
def test_oracle_matches_reference_python(tag):
    meta, g = load_golden(tag)
    sc = load_scenario(meta['map'])
    assert meta['all_ts_ids'] == sc.signal_ids
    env = OracleEnv(sc, env_index=0, seed=meta['seed'], max_distance=meta['max_distance'], sigma=-1.0, speed_dev=1,
                    step_ratio=meta.get('step_ratio', 1), tls_expiry=meta.get('tls_expiry', 0))
    env.observe()
    t0 = 0
    if meta.get('preroll'):         # loaded network: roll forward, then fresh Signal objects (as the fixture's generator did)
        for k in range(meta['preroll']):
            env.step(preroll_actions(sc, meta['seed'], 0, k))
        env.reinit_signals()
        env.observe()
    S, O = sc.n_signals, sc.n_obs
    for k in range(meta['steps'] + 1):
        if k > 0:
            env.step(g['actions'][k - 1])
        o = env.outputs()
        # integer-valued quantities: bit-exact
        np.testing.assert_array_equal(o['phase'], g['phase'][k])
        np.testing.assert_array_equal(o['lane_agg'][:, :4], g['agg'][k][:, :4])
        np.testing.assert_array_equal(o['mplight'].reshape(-1), g['mplight'][k])
        np.testing.assert_array_equal(o['wave'].reshape(-1), g['wave'][k])
        np.testing.assert_array_equal(o['wait'], g['wait'][k])
        np.testing.assert_array_equal(o['pressure'], g['pressure'][k])
        np.testing.assert_array_equal(o['wait_norm'], g['wait_norm'][k].astype(np.float32))
        if k > 0:
            np.testing.assert_array_equal(o['queue_sum'], g['queue_sum'][k - 1])
            np.testing.assert_array_equal(o['queue_max'], g['queue_max'][k - 1])
        # speed sums: the reference adds float64 in vehicle order, the kernels add Q16 fixed point.
        # tolerance: 2^-17 per vehicle (<= 64 vehicles per lane) + fp32 rounding of the sum
        np.testing.assert_allclose(o['lane_agg'][:, 4], g['agg'][k][:, 4], rtol=1e-6, atol=64 * 2.0 ** -17)
        # states.drq_norm / drq rows (signal-major, Signal.lanes order) -- fp32 vs the reference's float64
        dn = g['drq_norm'][k].reshape(O, 5)
        np.testing.assert_allclose(o['drq_norm'], dn, rtol=2e-6, atol=2e-6)
        np.testing.assert_array_equal(o['drq_norm'][:, 0], dn[:, 0])
        dq = g['drq'][k].reshape(O, 5)
        np.testing.assert_array_equal(o['lane_agg'][:, [1, 2, 0]], dq[:, 1:4])
        assert env.time == g['time'][k] - sc.begin
    st = env.stats()
    for key, val in meta['oracle_stats'].items():
        assert st[key] == val, key


def test_done_rule_and_time():
    meta, g = load_golden('cologne1_d200')
    assert not g['done'].any()          # 48 steps of 360
    assert g['time'][0] == 25200.0 and g['time'][1] == 25210.0
    meta, g = load_golden('cologne1_d50_full')      # the whole episode: done exactly at the last step
    assert g['done'][-1] and not g['done'][:-1].any() and g['time'][-1] == 28800.0
    meta, g = load_golden('ingolstadt21_d200_warm180')
    assert g['time'][0] == 57600.0 + 1800.0 and g['agg'][:, :, 3].max() >= 200     # long waiting_times on the loaded network


def _generated_case():
    import json
    import os
    from conftest import GOLDEN
    from resco_amd.scenario import Scenario
    with open(os.path.join(GOLDEN, 'grid4x4_generated.json')) as f:
        meta = json.load(f)
    return meta, dict(np.load(os.path.join(GOLDEN, 'grid4x4_generated.npz'))), Scenario.load(os.path.join(GOLDEN, 'grid4x4_generated_scenario.npz'))


def test_generate_config_matches_the_reference_s_own_method():
    """Signal.generate_config (traffic_signal.py:106-170), the fallback for signals without a signal_configs entry: grid4x4's net with
    NO per-signal entry -- the reference's unmodified Signal derived lanes / lane_sets / downstream for all 16 signals from
    getControlledLinks (tests/golden/make_generated_config_golden.py); the scenario compiler's restatement must give the same."""
    from resco_amd.scenario import generate_signal_config
    meta, g, sc = _generated_case()
    assert meta['all_ts_ids'] == sc.signal_ids and len(sc.signal_ids) == 16
    for sid, want in meta['signals'].items():
        m = sc.signal_meta[sid]
        assert m['generated'] and m['lanes'] == want['lanes'] and m['lane_sets'] == want['lane_sets'] and m['downstream'] == want['downstream']
        assert want['lane_sets_outbound'] == {} and want['outbound_lanes'] == [] and m['outbound_lanes'] == []
        assert {k: int(v) for k, v in m['yellow_dict'].items()} == want['yellow_dict']
        lanes, sets, down = generate_signal_config([[tuple(t) for t in lk] for lk in m['controlled_links']], sid)
        assert (lanes, sets, down) == (want['lanes'], want['lane_sets'], want['downstream'])
    # what generate_config cannot digest is refused, not guessed at
    with pytest.raises(EnvironmentError):
        generate_signal_config([[('x_0', 'y_0', '')]] * 40, 'big')
    with pytest.raises(EnvironmentError):
        generate_signal_config([[('-12345#1_0', 'y_0', '')]] * 36, 'osm')


def test_oracle_on_generated_config_matches_reference_python():
    meta, g, sc = _generated_case()
    env = OracleEnv(sc, env_index=0, seed=meta['seed'], max_distance=200, sigma=-1.0, speed_dev=1)
    env.observe()
    O = sc.n_obs
    for k in range(meta['steps'] + 1):
        if k > 0:
            env.step(g['actions'][k - 1])
        o = env.outputs()
        np.testing.assert_array_equal(o['phase'], g['phase'][k])
        np.testing.assert_array_equal(o['lane_agg'][:, :4], g['agg'][k][:, :4])
        np.testing.assert_array_equal(o['wave'].reshape(-1), g['wave'][k])
        np.testing.assert_array_equal(o['wait'], g['wait'][k])
        np.testing.assert_array_equal(o['pressure'], g['pressure'][k])       # no outbound lanes: pressure = -queue
        np.testing.assert_allclose(o['drq_norm'], g['drq_norm'][k].reshape(O, 5), rtol=2e-6, atol=2e-6)
    st = env.stats()
    for key, val in meta['oracle_stats'].items():
        assert st[key] == val, key
