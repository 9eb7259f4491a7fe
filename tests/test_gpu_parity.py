"""GPU (-m gpu): the HIP library through the C ABI vs the CPU oracle (bit-exact) and vs the golden outputs
of the reference's own Python, plus size-independent properties at the BASELINE sizes."""
import json
import os
import tempfile

import numpy as np
import pytest

import sys

from conftest import ALL_CASES, HOT_CASES, ROOT, load_golden, load_scenario, preroll_actions

sys.path.insert(0, os.path.join(ROOT, 'tests'))

pytestmark = pytest.mark.gpu

INT_BUFS = ['phase', 'mplight', 'wave', 'pressure', 'queue_sum', 'queue_max', 'arrivals', 'departures']
FLT_BUFS = ['lane_agg', 'drq_norm', 'wait', 'wait_norm', 'mplight_full']
VEH = [('veh_lane', 'lane'), ('veh_trip', 'trip'), ('veh_pos', 'pos'), ('veh_speed', 'speed'),
       ('veh_cursor', 'cursor'), ('veh_swait', 'sumo_wait'), ('veh_tloss', 'time_loss'), ('veh_rwait', 'resco_wait'),
       ('veh_owner', 'owner'), ('veh_depart', 'depart'), ('veh_accel', 'accel')]


def assert_env_equal(sim, orcs, step):
    out = sim.outputs(INT_BUFS + FLT_BUFS)
    vg = {g: sim.read(g) for g, _ in VEH}
    env = sim.read('env')
    for e, o in enumerate(orcs):
        ref = o.outputs()
        for b in INT_BUFS + FLT_BUFS:      # floats too: both sides are IEEE fp32 without contraction
            np.testing.assert_array_equal(out[b][e], ref[b], err_msg='%s env %d step %d' % (b, e, step))
        vo = o.vehicles()
        assert env[e, 2] == vo['hw'] and env[e, 1] == vo['next_trip'] and env[e, 0] == o.time and env[e, 3] == o.stats()['active']
        hw = vo['hw']
        free = vo['lane'][:hw] == 0xFFFF
        active = ~free
        for g, r in VEH:
            a, b_ = vg[g][e][:hw], vo[r][:hw]
            if r == 'trip':
                a, b_ = a.astype(np.int64), b_.astype(np.int64) & 0xFFFF
            elif r in ('resco_wait', 'owner', 'depart', 'accel'):
                a, b_ = a[active], b_[active]
            elif r != 'lane':
                a, b_ = a[~free], b_[~free]
            np.testing.assert_array_equal(a, b_, err_msg='%s env %d step %d' % (g, e, step))


@pytest.mark.parametrize('name,n_envs,steps,sigma,speed_dev,fixed', [
    ('cologne1', 5, 40, 0.0, 0, 0),          # parity mode (deterministic)
    ('cologne1', 3, 60, -1.0, 1, 0),         # bench mode (dawdling + speedFactor)
    ('cologne1', 2, 30, -1.0, 1, 1),         # FIXED programme (BASELINE config 1 plumbing)
    ('cologne8', 3, 40, -1.0, 1, 0),
    ('ingolstadt21', 2, 45, -1.0, 1, 0),
    ('cologne3', 2, 40, -1.0, 1, 0),         # <vehicle><route> demand, explicit routes
    ('ingolstadt1', 2, 40, -1.0, 1, 1),
    ('ingolstadt7', 2, 40, -1.0, 1, 0),
    ('ingolstadt7', 2, 260, -1.0, 1, 1),     # long enough for lane-change blocks and cooperation requests
])
def test_gpu_equals_oracle_bit_exact(name, n_envs, steps, sigma, speed_dev, fixed):
    from oracle.pyoracle import OracleEnv
    from resco_amd.sim import BatchedSim
    sc = load_scenario(name)
    base = 17
    sim = BatchedSim(sc, n_envs, seed=3, sigma=sigma, speed_dev=speed_dev, fixed_program=fixed, env_base=base)
    orcs = [OracleEnv(sc, env_index=base + e, seed=3, sigma=sigma, speed_dev=speed_dev, fixed_program=fixed)
            for e in range(n_envs)]
    for o in orcs:
        o.observe()
    assert_env_equal(sim, orcs, -1)
    rng = np.random.default_rng(0)
    for step in range(steps):
        acts = np.stack([rng.integers(0, sc.tls_ngreen) for _ in range(n_envs)]).astype(np.int32)
        if step == 5:
            acts[0, 0] = 99                  # out-of-range action: ignored, phase kept
        sim.step(acts)
        for e, o in enumerate(orcs):
            o.step(acts[e])
        assert_env_equal(sim, orcs, step)
    st = sim.stats()
    for e, o in enumerate(orcs):
        so = o.stats()
        for k in st:
            assert st[k][e] == so[k], (k, e)
    sim.close()


def _random_configs(n, seed):
    """seeded draws over what a handle can be created with (the same generator as tests/test_hostemu.py uses for the host emulation)"""
    rng = np.random.default_rng(seed)
    maps = ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
    out = []
    for i in range(n):
        out.append(dict(name=maps[int(rng.integers(0, len(maps)))], sigma=float(rng.choice([-1.0, 0.0, 0.3, 0.9])),
                        speed_dev=int(rng.integers(0, 2)), max_distance=float(rng.choice([1.0, 50.0, 200.0, 9999.0])),
                        fixed=int(rng.random() < 0.25), step_ratio=int(rng.choice([1, 1, 2, 3])), seed=int(rng.integers(0, 2 ** 31)),
                        env_base=int(rng.integers(0, 5000)), warm=int(rng.choice([0, 40, 90, 150])), n_envs=int(rng.integers(1, 4)),
                        steps=int(rng.integers(12, 28)), case=i,
                        tls_expiry=i % 2))       # both answers to what setPhase leaves behind (rs_params.tls_expiry)
    return out


@pytest.mark.parametrize('cfg', _random_configs(10, 4096), ids=lambda c: '%d-%s' % (c['case'], c['name']))
def test_randomised_parameter_sweep_bit_exact(cfg):
    """HIP path == oracle on every output and every vehicle field for random combinations of the handle's parameters (driver
    imperfection, speed factors, detector range, programme, step_ratio, seed, first environment), after a warm start under the
    on-device random policy, under random, repeated and out-of-range actions"""
    from oracle.pyoracle import OracleEnv
    from resco_amd.sim import BatchedSim
    sc = load_scenario(cfg['name'])
    kw = dict(seed=cfg['seed'], sigma=cfg['sigma'], speed_dev=cfg['speed_dev'], max_distance=cfg['max_distance'],
              fixed_program=cfg['fixed'], step_ratio=cfg['step_ratio'], tls_expiry=cfg['tls_expiry'])
    n = cfg['n_envs']
    sim = BatchedSim(sc, n, env_base=cfg['env_base'], **kw)
    orcs = [OracleEnv(sc, env_index=cfg['env_base'] + e, **kw) for e in range(n)]
    for o in orcs:
        o.observe()
    for k in range(cfg['warm'] // cfg['step_ratio']):
        sim.act_random(k)
        sim.sync()
        a = sim.read('actions')
        sim.step(None)
        for e, o in enumerate(orcs):
            o.step(a[e])
    rng = np.random.default_rng(cfg['case'])
    prev = np.zeros((n, sc.n_signals), np.int32)
    for step in range(cfg['steps']):
        a = np.stack([rng.integers(0, sc.tls_ngreen) for _ in range(n)]).astype(np.int32)
        keep = rng.random((n, sc.n_signals)) < 0.3
        a[keep] = prev[keep]
        if step % 9 == 4:
            a[int(rng.integers(0, n)), int(rng.integers(0, sc.n_signals))] = int(rng.choice([-1, 97]))
        prev = a.copy()
        sim.step(a)
        for e, o in enumerate(orcs):
            o.step(a[e])
        if step % 6 == 5 or step == cfg['steps'] - 1:
            assert_env_equal(sim, orcs, step)
    sim.close()


@pytest.mark.parametrize('name,block,steps', [
    ('ingolstadt7', 64, 120),           # one wave for 256 slots: no wave roles, every thread handles its slots in full
    ('ingolstadt7', 256, 120),          # one thread per slot
    ('ingolstadt7', -128, 120),         # 128-VGPR build
    ('ingolstadt21', 1024, 150),        # 64-VGPR build, 16 waves
    ('ingolstadt21', -10768, 150),      # 80-VGPR build, 12 waves
    ('ingolstadt21', -10256, 150),      # 4 waves for ~300 vehicles: the lists take most of them
])
def test_block_sizes_and_register_budgets_bit_exact(name, block, steps):
    """the wave-role scheduling of the step kernel (lists on their own waves, slots shared by the rest) gives the oracle's
    results for every workgroup shape and register budget, not only the default one"""
    from oracle.pyoracle import OracleEnv
    from resco_amd.sim import BatchedSim
    sc = load_scenario(name)
    n = 2
    sim = BatchedSim(sc, n, seed=21, block_threads=block, env_base=5)
    orcs = [OracleEnv(sc, env_index=5 + e, seed=21, sigma=-1.0, speed_dev=1) for e in range(n)]
    for o in orcs:
        o.observe()
    rng = np.random.default_rng(6)
    for step in range(steps):
        acts = np.stack([rng.integers(0, sc.tls_ngreen) for _ in range(n)]).astype(np.int32)
        sim.step(acts)
        for e, o in enumerate(orcs):
            o.step(acts[e])
        if step % 30 == 29:
            assert_env_equal(sim, orcs, step)
    sim.close()


def test_full_episode_bit_exact_ingolstadt21():
    """a whole 360-step episode (3600 ticks, ~4000 trips, congestion, slot reuse) stays bit-identical to the oracle"""
    from oracle.pyoracle import OracleEnv
    from oracle_batch import hashed_random_actions
    from resco_amd.sim import BatchedSim
    sc = load_scenario('ingolstadt21')
    n, seed = 2, 12
    sim = BatchedSim(sc, n, seed=seed, trip_log=1)
    orcs = [OracleEnv(sc, env_index=e, seed=seed, sigma=-1.0, speed_dev=1, trip_log=1) for e in range(n)]
    for o in orcs:
        o.observe()
    for k in range(360):
        sim.act_random(k)
        sim.step(None)
        acts = hashed_random_actions(sc, seed, 0, n, k)
        for e, o in enumerate(orcs):
            o.step(acts[e])
        if k % 30 == 29 or k == 359:
            assert_env_equal(sim, orcs, k)
    st = sim.stats()
    log = sim.read('trip_log')
    for e, o in enumerate(orcs):
        so = o.stats()
        for key in st:
            assert st[key][e] == so[key], (key, e)
        np.testing.assert_array_equal(log[e], o.trip_log())
    assert st['arrived'].min() > 3000
    sim.close()


@pytest.mark.parametrize('tag', ALL_CASES)
@pytest.mark.parametrize('fast', [True, False])
def test_multisignal_matches_reference_python(tag, fast):
    """The reference's MultiSignal/Signal/states/rewards (golden) vs resco_amd.MultiSignal on the HIP path."""
    from resco_amd import rewards, states
    from resco_amd.config.map_config import map_configs
    from resco_amd.multi_signal import MultiSignal
    meta, g = load_golden(tag)
    mc = map_configs[meta['map']]
    tmp = tempfile.mkdtemp() + os.sep
    env = MultiSignal('golden', meta['map'], mc['net'], states.mplight, rewards.wait, step_length=mc['step_length'],
                      yellow_length=mc['yellow_length'], end_time=mc['end_time'], max_distance=meta['max_distance'],
                      lights=mc['lights'], log_dir=tmp, seed=meta['base_seed'], use_fast_path=fast, step_ratio=meta.get('step_ratio', 1),
                      tls_expiry=meta.get('tls_expiry', 0))
    ids = meta['all_ts_ids']
    assert env.all_ts_ids == ids and env.ts_order == meta['ts_order']
    assert {k: list(v) for k, v in env.obs_shape.items()} == meta['obs_shape']
    assert env.connection_name == meta['connection_name'] and env.n_agents == len(ids)
    assert [len(env.phases[t]) for t in ids] == meta['n_green']
    for t in ids:
        assert env.signals[t].lanes == meta['signals'][t]['lanes']
        assert env.signals[t].yellow_dict == meta['signals'][t]['yellow_dict']
    obs = env.reset()
    if meta.get('preroll'):
        # loaded network: roll the simulation forward under the on-device random policy, then build fresh Signal objects on
        # it -- exactly what the fixture's generator did before the reference's MultiSignal took over
        for k in range(meta['preroll']):
            env.sim.act_random(k)
            env.sim.step(None)
        obs = env.reinit_signals()
    assert list(obs.keys()) == ids

    def check(k):
        for fn in ('drq', 'drq_norm', 'mplight', 'mplight_full', 'wave'):
            f = getattr(states, fn)
            out = env._evaluate(f) if fast else f(env.signals)
            flat = np.concatenate([np.asarray(out[t], dtype=np.float64).reshape(-1) for t in ids])
            if fn in ('mplight', 'wave'):
                np.testing.assert_array_equal(flat, g[fn][k], err_msg=fn)
            else:   # fp32 kernels vs the reference's float64 sums of per-vehicle speeds
                np.testing.assert_allclose(flat, g[fn][k], rtol=2e-6, atol=1e-3 if fn in ('drq', 'mplight_full') else 2e-6, err_msg=fn)
        for fn in ('wait', 'wait_norm', 'pressure'):
            f = getattr(rewards, fn)
            out = env._evaluate(f) if fast else f(env.signals)
            np.testing.assert_array_equal(np.asarray([float(out[t]) for t in ids], np.float32),
                                          g[fn][k].astype(np.float32), err_msg=fn)
        assert [env.signals[t].phase for t in ids] == g['phase'][k].tolist()
        assert env.sim_time() == g['time'][k]

    check(0)
    for k in range(meta['steps']):
        obs, rew, done, info = env.step({t: int(a) for t, a in zip(ids, g['actions'][k])})
        assert done == bool(g['done'][k]) and info == {'eps': 1}
        np.testing.assert_array_equal(np.concatenate([obs[t] for t in ids]), g['mplight'][k + 1])
        assert [float(rew[t]) for t in ids] == g['wait'][k + 1].tolist()
        check(k + 1)
    ts = env.trip_stats()
    for key in ('inserted', 'arrived', 'sum_duration', 'sum_depart_delay', 'sum_waiting', 'sum_time_loss_q10'):
        assert ts[key] == meta['oracle_stats'][key], key
    env.reset()
    with open(os.path.join(tmp, env.connection_name, 'metrics_1.csv')) as f:
        assert f.read() == meta['metrics_csv']          # calc_metrics / save_metrics text, byte for byte
    env.close()




def test_generated_signal_config_on_the_device():
    """A map WITHOUT per-signal signal_configs entries (grid4x4's net, synthetic demand): every signal's lanes come from the
    reference's generate_config fallback (traffic_signal.py:106-170).  HIP path vs the fixture the reference's own Python produced
    (tests/golden/make_generated_config_golden.py): phases, lane aggregates, wave, wait, pressure exact, drq_norm to fp32."""
    import json
    from conftest import GOLDEN
    from resco_amd.scenario import Scenario
    from resco_amd.sim import BatchedSim
    with open(os.path.join(GOLDEN, 'grid4x4_generated.json')) as f:
        meta = json.load(f)
    g = dict(np.load(os.path.join(GOLDEN, 'grid4x4_generated.npz')))
    sc = Scenario.load(os.path.join(GOLDEN, 'grid4x4_generated_scenario.npz'))
    sim = BatchedSim(sc, 1, seed=meta['seed'], max_distance=200, tls_expiry=meta.get('tls_expiry', 0))      # (fixtures without the key: round 5's default)
    for k in range(meta['steps'] + 1):
        if k > 0:
            sim.step(g['actions'][k - 1][None, :])
        np.testing.assert_array_equal(sim.read('phase')[0], g['phase'][k])
        np.testing.assert_array_equal(sim.read('lane_agg')[0][:, :4], g['agg'][k][:, :4])
        np.testing.assert_array_equal(sim.read('wave')[0].reshape(-1), g['wave'][k])
        np.testing.assert_array_equal(sim.read('wait')[0], g['wait'][k])
        np.testing.assert_array_equal(sim.read('pressure')[0], g['pressure'][k])
        np.testing.assert_allclose(sim.read('drq_norm')[0], g['drq_norm'][k].reshape(sc.n_obs, 5), rtol=2e-6, atol=2e-6)
    st = sim.stats()
    for key in ('inserted', 'arrived', 'sum_duration', 'sum_waiting'):
        assert int(st[key][0]) == meta['oracle_stats'][key], key
    sim.close()


@pytest.mark.parametrize('tag', HOT_CASES + ['cologne1_d50_full'])
def test_device_agents_match_reference(tag):
    """rs_act_maxwave (MAXPRESSURE / MAXWAVE on device) vs the reference agents' actions (golden)."""
    from resco_amd.sim import BatchedSim
    meta, g = load_golden(tag)
    sc = load_scenario(meta['map'])
    sim = BatchedSim(sc, 3, seed=meta['seed'], max_distance=meta['max_distance'], tls_expiry=meta.get('tls_expiry', 0))
    # env 0 is the golden environment; envs 1,2 differ (other RNG keys) and only have to stay in range
    for k in range(meta['steps'] + 1):
        sim.act_maxwave(1)
        sim.sync()
        a1 = sim.read('actions')
        sim.act_maxwave(0)
        sim.sync()
        a2 = sim.read('actions')
        np.testing.assert_array_equal(a1[0], g['act_maxpressure'][k])
        np.testing.assert_array_equal(a2[0], g['act_maxwave'][k])
        assert (a1 >= 0).all() and (a1 < sc.tls_ngreen[None, :]).all()
        if k < meta['steps']:
            sim.step(np.repeat(g['actions'][k][None, :], 3, axis=0))
    sim.close()


@pytest.mark.parametrize('tag', ['cologne8_d200', 'ingolstadt21_d200'])
def test_fma2c_through_signal_views(tag):
    """FMA2C state + reward (arrivals / departures sets, manager keys) through MultiSignal's Signal views vs the
    reference's own functions (golden)."""
    from resco_amd import rewards, states
    from resco_amd.config.map_config import map_configs
    from resco_amd.config.mdp_config import activate
    from resco_amd.multi_signal import MultiSignal
    meta, g = load_golden(tag)
    mc = map_configs[meta['map']]
    activate('FMA2C', meta['map'])
    env = MultiSignal('golden', meta['map'], mc['net'], states.fma2c, rewards.fma2c, yellow_length=3,
                      end_time=mc['end_time'], max_distance=meta['max_distance'], lights=mc['lights'],
                      log_dir=tempfile.mkdtemp() + os.sep, seed=meta['base_seed'], tls_expiry=bool(meta.get('tls_expiry', 0)))
    keys = meta['fma2c_keys']['fma2c']
    assert list(env.obs_shape.keys()) == keys and env.ts_order == keys
    assert len(env.action_space) == len(meta['all_ts_ids'])          # managers have no action space
    obs = env.reset()

    def flat(d):
        return np.concatenate([np.asarray(d[k_], dtype=np.float64).reshape(-1) for k_ in keys])

    np.testing.assert_allclose(flat(obs), g['state_fma2c'][0], rtol=1e-6, atol=1e-6)
    for k in range(meta['steps']):
        act = {t: int(a) for t, a in zip(meta['all_ts_ids'], g['actions'][k])}
        obs, rew, done, info = env.step(act)
        np.testing.assert_allclose(flat(obs), g['state_fma2c'][k + 1], rtol=1e-6, atol=1e-6)
        np.testing.assert_allclose([float(rew[k_]) for k_ in keys], g['reward_fma2c'][k + 1], rtol=1e-6, atol=1e-6)
    env.close()


def test_trip_log_and_tripinfo_output():
    """per-trip records (f-4): HIP == oracle, and the tripinfo XML MultiSignal writes is what the reference's
    utils/readXML.py reads (timeLoss + departDelay per trip, unfinished trips included)."""
    import xml.etree.ElementTree as ET
    from oracle.pyoracle import OracleEnv
    from resco_amd import rewards, states
    from resco_amd.multi_signal import MultiSignal
    from resco_amd.sim import BatchedSim
    sc = load_scenario('cologne1')
    sim = BatchedSim(sc, 3, seed=8, trip_log=1)
    orcs = [OracleEnv(sc, env_index=e, seed=8, sigma=-1.0, speed_dev=1, trip_log=1) for e in range(3)]
    rng = np.random.default_rng(1)
    for k in range(60):
        a = rng.integers(0, 4, (3, 1)).astype(np.int32)
        sim.step(a)
        for e, o in enumerate(orcs):
            o.step(a[e])
    log = sim.read('trip_log')
    wt = sim.read('veh_wtot')
    for e, o in enumerate(orcs):
        np.testing.assert_array_equal(log[e], o.trip_log())
        v = o.vehicles()
        act = v['lane'][:v['hw']] != 0xFFFF
        np.testing.assert_array_equal(wt[e][:v['hw']][act], o.wtot()[:v['hw']][act])
    assert (log[:, :, 1] > 0).sum() == sim.stats()['arrived'].sum()
    sim.close()

    tmp = tempfile.mkdtemp() + os.sep
    env = MultiSignal('t', 'cologne1', None, states.mplight, rewards.wait, yellow_length=3, end_time=28800, log_dir=tmp, seed=2)
    env.reset()
    for k in range(80):
        env.step({env.all_ts_ids[0]: k % 4})
    ts = env.trip_stats()
    (n_queued,), (waited,) = env.sim.backlog()
    delay = env.sim.trip_delay()[0]
    delay_all = env.sim.trip_metrics()['delay_all'][0]
    env.reset()                                   # closes episode 1: writes metrics_1.csv and tripinfo_1.xml
    root = ET.parse(os.path.join(tmp, env.connection_name, 'tripinfo_1.xml')).getroot()
    trips = list(root)
    assert len(trips) == ts['inserted'] and sum(1 for t in trips if float(t.get('arrival')) > 0) == ts['arrived']
    fin = [t for t in trips if float(t.get('arrival')) > 0]
    assert abs(sum(float(t.get('timeLoss')) for t in fin) - ts['sum_time_loss_q10'] / 1024.0) < 0.01 * len(fin) + 1
    assert sum(float(t.get('duration')) for t in fin) == ts['sum_duration']
    assert sum(float(t.get('departDelay')) for t in trips) == ts['sum_depart_delay']
    assert all(float(t.get('depart')) >= 25200 for t in trips)
    # BatchedSim.trip_delay() is utils/readXML.py's episode figure: (timeLoss + departDelay) per tripinfo entry; the
    # trips still queued for insertion are charged (delay_all) only for <vehicle> demand files -- cologne1 lists <trip>s
    total = sum(float(t.get('timeLoss')) + float(t.get('departDelay')) for t in trips)
    assert abs(delay - total / len(trips)) < 0.02
    assert abs(delay_all - (total + float(waited)) / (len(trips) + int(n_queued))) < 0.02
    env.close()


@pytest.mark.parametrize('name', ['cologne1', 'cologne3'])
def test_trip_metrics_equal_what_the_reference_reads(name):
    """f-4 closed against its consumer: tests/golden/readxml_<map>.json holds what the reference's unmodified
    utils/readXML.py:16-77 computed (build container, tests/golden/make_readxml_fixture.py) from a tripinfo_1.xml written by
    this package's writer for one episode of the hashed random policy; the same episode through the C ABI must give the
    same per-episode averages from BatchedSim.trip_metrics() -- cologne1 (<trip> demand: 189 trips never get onto the
    network and readXML leaves them out) and cologne3 (<vehicle> demand: readXML charges them)."""
    from resco_amd.sim import BatchedSim
    with open(os.path.join(ROOT, 'tests', 'golden', 'readxml_%s.json' % name)) as f:
        fx = json.load(f)
    sc = load_scenario(name)
    sim = BatchedSim(sc, 2, seed=fx['seed'], max_distance=fx['max_distance'], trip_log=1, tls_expiry=fx.get('tls_expiry', 0))
    for k in range(360):
        sim.act_random(k)
        sim.step(None)
    m = sim.trip_metrics()
    st = sim.stats()
    assert int(st['arrived'][0]) == fx['arrived'] and int(st['arrived'][0] + st['active'][0]) == fx['entries']
    assert int(sim.backlog()[0][0]) == fx['queued_never_departed']
    # (the XML carries two decimals per entry; on <vehicle> files readXML's never-departed rule -- scheduled later than the
    #  last vehicle that did depart -- is an approximation of the insertion backlogs this build keeps per lane)
    tol = 2e-4 if sc.demand_tag == 'trip' else 1e-3
    assert abs(m['delay'][0] / fx['readXML']['timeLoss'] - 1.0) < tol
    assert abs(m['duration'][0] / fx['readXML']['duration'] - 1.0) < 1e-6
    assert abs(m['waiting'][0] / fx['readXML']['waitingTime'] - 1.0) < 1e-6
    sim.close()


def test_full_episode_done_rule_and_metrics_file():
    """360 steps of 10 s: done turns True exactly when simulation.getTime() reaches end_time
    (multi_signal.py:190); metrics_<run>.csv gets one line per step at the next reset (multi_signal.py:107-111)."""
    from resco_amd import rewards, states
    from resco_amd.multi_signal import MultiSignal
    tmp = tempfile.mkdtemp() + os.sep
    env = MultiSignal('t', 'cologne1', None, states.wave, rewards.pressure, yellow_length=3, end_time=28800,
                      max_distance=50, log_dir=tmp, seed=5)
    for episode in (1, 2):
        obs = env.reset()
        steps, done = 0, False
        while not done:
            obs, rew, done, info = env.step({env.all_ts_ids[0]: steps % 4})
            steps += 1
            assert info == {'eps': episode}
        assert steps == 360 and env.sim_time() == 28800.0
    env.close()
    for episode in (1, 2):
        with open(os.path.join(tmp, env.connection_name, 'metrics_%d.csv' % episode)) as f:
            rows = f.read().splitlines()
        assert len(rows) == 360 and rows[0].startswith('25210.0, ') and rows[-1].startswith('28800.0, ')
    ts = env.scenario.n_trips
    assert ts == 2015


def test_gymma_list_api_and_custom_state_fn():
    from resco_amd import rewards, states
    from resco_amd.multi_signal import MultiSignal

    def my_state(signals):          # an unmodified states.py-style plugin the registry has never heard of
        return {sid: np.asarray([s.phase, sum(s.full_observation[l]['queue'] for l in s.lanes),
                                 len(s.full_observation['num_vehicles'])]) for sid, s in signals.items()}

    env = MultiSignal('t', 'cologne8', None, my_state, rewards.pressure, yellow_length=3, end_time=28800,
                      log_dir=tempfile.mkdtemp() + os.sep, seed=1, gymma=True)
    obs = env.reset()
    assert isinstance(obs, list) and len(obs) == 8 and obs[0].shape == (3,)
    for _ in range(12):
        obs, rew, done, info = env.step([0] * 8)
    assert isinstance(rew, list) and done == [False] and info == {'eps': 1}
    ref = states.mplight(env.signals)
    for i, t in enumerate(env.ts_order):
        assert obs[i][0] == ref[t][0]
        fo = env.signals[t].full_observation
        assert obs[i][1] == sum(fo[l]['queue'] for l in env.signals[t].lanes)
        assert fo['arrivals'] <= fo['num_vehicles']
        assert set(env.signals[t].waiting_times) <= fo['num_vehicles']
    env.close()


def test_sampled_envs_bit_exact_at_full_occupancy():
    """BASELINE config 3 launch shape (4096 workgroups, several resident per CU): eight sampled environments stay
    bit-identical to the CPU oracle for 40 env-steps of the on-device random policy.  Few-environment parity runs
    leave most of the chip idle and do not exercise the timing that exposes intra-workgroup races."""
    from oracle.pyoracle import OracleEnv
    from resco_amd.sim import BatchedSim
    sc = load_scenario('ingolstadt21')
    N, base = 4096, 1000
    picks = [0, 1, 63, 511, 2048, 3000, 4094, 4095]
    sim = BatchedSim(sc, N, seed=9, env_base=base)
    orcs = [OracleEnv(sc, env_index=base + e, seed=9, sigma=-1.0, speed_dev=1) for e in picks]
    for o in orcs:
        o.observe()
    names = INT_BUFS + FLT_BUFS
    for step in range(40):
        sim.act_random(step)
        acts = sim.read('actions')                  # the hashed U{0..G_s-1} actions the kernel is about to use
        sim.step(None)
        for e, o in zip(picks, orcs):
            o.step(acts[e])
        if step % 13 == 0 or step == 39:
            out = {b: sim.read(b) for b in names}
            vg = {g: sim.read(g) for g in ('veh_lane', 'veh_pos', 'veh_speed', 'veh_trip')}
            env = sim.read('env')
            for e, o in zip(picks, orcs):
                ref = o.outputs()
                for b in names:
                    np.testing.assert_array_equal(out[b][e], ref[b], err_msg='%s env %d step %d' % (b, e, step))
                vo = o.vehicles()
                hw = vo['hw']
                assert env[e, 2] == hw and env[e, 1] == vo['next_trip'] and env[e, 0] == o.time
                used = vo['lane'][:hw] != 0xFFFF
                np.testing.assert_array_equal(vg['veh_lane'][e][:hw], vo['lane'][:hw])
                np.testing.assert_array_equal(vg['veh_pos'][e][:hw][used], vo['pos'][:hw][used])
                np.testing.assert_array_equal(vg['veh_speed'][e][:hw][used], vo['speed'][:hw][used])
    st = sim.stats()
    for e, o in zip(picks, orcs):
        so = o.stats()
        assert st['inserted'][e] == so['inserted'] and st['arrived'][e] == so['arrived']
    sim.close()


def test_full_episode_invariants_at_baseline_size():
    """One whole 360-step episode of BASELINE config 3 (4096 envs, on-device random policy): every environment
    ends at t = 3600 with its vehicles conserved, inside their lanes, and with the episode totals consistent."""
    from resco_amd.sim import BatchedSim
    sc = load_scenario('ingolstadt21')
    N = 4096
    sim = BatchedSim(sc, N, seed=21)
    prev_arrived = np.zeros(N, np.int64)
    for k in range(360):
        sim.act_random(k)
        sim.step(None)
        if k in (119, 239, 359):
            st = sim.stats()
            env = sim.read('env')
            assert (env[:, 0] == 10 * (k + 1)).all()
            assert (st['inserted'] == st['arrived'] + st['active']).all()
            assert (env[:, 1] == st['inserted']).all() and (env[:, 3] == st['active']).all() and (env[:, 1] <= sc.n_trips).all()
            assert (st['pending'] >= 0).all() and (st['inserted'] + st['pending'] <= sc.n_trips).all()
            assert (st['arrived'] >= prev_arrived).all()
            prev_arrived = st['arrived'].copy()
            lane, pos, spd = sim.read('veh_lane'), sim.read('veh_pos'), sim.read('veh_speed')
            act = lane != 0xFFFF
            assert (act.sum(axis=1) == st['active']).all() and (st['active'] <= sc.capacity).all()
            assert (pos[act] >= 0).all() and (pos[act] <= sc.lane_len[lane[act]] + 1e-3).all()
            assert (spd[act] >= 0).all() and (spd[act] <= 60.0).all()
            trip = sim.read('veh_trip')
            for e in (0, 1777, 4095):                  # a trip occupies at most one slot
                t_e = trip[e][lane[e] != 0xFFFF]
                assert len(np.unique(t_e)) == len(t_e)
    st = sim.stats()
    assert (st['ticks'] == 3600).all()
    assert (st['sum_duration'] >= st['sum_time_loss_q10'] // 1024).all()       # time loss <= travel time
    assert (st['sum_waiting'] <= st['active_ticks']).all()
    assert st['arrived'].min() > 1000 and len(np.unique(st['arrived'])) > 100   # traffic flows; envs decorrelate
    sim.close()


def test_properties_at_baseline_size():
    """BASELINE config 3 size: ingolstadt21 x 4096 lock-step environments, on-device random policy."""
    from resco_amd.sim import BatchedSim
    sc = load_scenario('ingolstadt21')
    N = 4096
    sim = BatchedSim(sc, N, seed=5)
    for k in range(12):
        sim.act_random(k)
        sim.step(None)
    sim.sync()
    st = sim.stats()
    env = sim.read('env')
    assert (env[:, 0] == 120).all()
    assert (st['inserted'] == st['arrived'] + st['active']).all()          # vehicle conservation per env
    assert (env[:, 1] == st['inserted']).all()
    lane = sim.read('veh_lane')
    assert ((lane != 0xFFFF).sum(axis=1) == st['active']).all()
    assert len(np.unique(st['active'])) > 8                                  # environments decorrelate
    pos, spd = sim.read('veh_pos'), sim.read('veh_speed')
    act = lane != 0xFFFF
    assert (pos[act] <= sc.lane_len[lane[act]] + 1e-3).all() and (spd[act] >= 0).all()
    agg = sim.read('lane_agg')
    q = agg[:, :, 0]
    mp = sim.read('mplight')
    ph = sim.read('phase')
    assert (mp[:, :, 0] == ph).all() and (ph < sc.tls_nphase[None, :]).all()
    # rewards.pressure == -(sum of own queues - sum of downstream queues): recompute from the aggregates
    pr = sim.read('pressure')
    for s in (0, 9, 20):
        own = q[:, sc.sig_obs_start[s]:sc.sig_obs_start[s + 1]].sum(axis=1)
        down = q[:, sc.pr_out_idx[sc.pr_out_start[s]:sc.pr_out_start[s + 1]]].sum(axis=1)
        np.testing.assert_array_equal(pr[:, s], -(own - down).astype(np.int32))
    np.testing.assert_array_equal(sim.read('wait'), -np.add.reduceat(agg[:, :, 2], sc.sig_obs_start[:-1], axis=1))
    h = sim.read('drq_norm_f16').astype(np.float32)
    dn = sim.read('drq_norm')
    s, o0, o1 = 13, sc.sig_obs_start[13], sc.sig_obs_start[14]
    np.testing.assert_allclose(h[:, s, :o1 - o0], dn[:, o0:o1], rtol=1e-3, atol=1e-3)
    assert (h[:, 0, sc.sig_obs_start[1] - sc.sig_obs_start[0]:] == 0).all()         # zero padding
    sim.close()


def test_determinism_sharding_and_snapshot():
    from resco_amd.sim import BatchedSim
    sc = load_scenario('cologne8')
    keys = ['lane_agg', 'mplight', 'veh_pos', 'veh_lane', 'veh_rwait', 'wait']

    def run(n, base, steps, sim=None):
        sim = sim or BatchedSim(sc, n, seed=9, env_base=base)
        for k in range(steps):
            sim.act_random(k)
            sim.step(None)
        return sim, {k: sim.read(k) for k in keys}

    whole, a = run(64, 0, 30)
    _, a2 = run(64, 0, 30)
    lo, b0 = run(32, 0, 30)
    hi, b1 = run(32, 32, 30)
    for k in keys:
        np.testing.assert_array_equal(a[k], a2[k])                                   # same seed -> same batch
        np.testing.assert_array_equal(a[k], np.concatenate([b0[k], b1[k]]))          # env-batch split == whole
    snap = whole.snapshot()
    _, c1 = run(64, 0, 5, whole)
    whole.restore(snap)
    for k in keys:
        np.testing.assert_array_equal(whole.read(k), a[k])
    _, c2 = run(64, 0, 5, whole)
    for k in keys:
        np.testing.assert_array_equal(c1[k], c2[k])
    whole.free_snapshot(snap)


def test_device_random_policy_matches_its_definition():
    from oracle_batch import hashed_random_actions
    from resco_amd.sim import BatchedSim
    sc = load_scenario('cologne8')
    sim = BatchedSim(sc, 6, seed=77, env_base=40)
    for key in (0, 1, 359, 123456):
        sim.act_random(key)
        sim.sync()
        np.testing.assert_array_equal(sim.read('actions'), hashed_random_actions(sc, 77, 40, 6, key))
    sim.close()


def test_handles_of_different_scenarios_coexist():
    """the dynamic-LDS attribute of the step kernel is shared by all handles of a process"""
    from resco_amd.sim import BatchedSim
    big = BatchedSim(load_scenario('ingolstadt21'), 8, seed=1)
    small = BatchedSim(load_scenario('cologne1'), 8, seed=1)
    for k in range(5):
        small.act_random(k); small.step(None)
        big.act_random(k); big.step(None)
    big.sync(); small.sync()
    assert (big.read('env')[:, 0] == 50).all() and (small.read('env')[:, 0] == 50).all()
    big.close(); small.close()


def test_zero_copy_tensors_at_the_agent_boundary():
    import torch
    from resco_amd.multi_signal import VecMultiSignal
    env = VecMultiSignal('cologne1', 1024, states=('drq_norm', 'mplight'), rewards=('pressure', 'wait'), seed=2)
    obs = env.reset()
    assert obs['drq_norm'].shape == (1024, 8, 5) and obs['drq_norm'].is_cuda and obs['mplight'].dtype == torch.int32
    done = False
    for k in range(20):
        acts = torch.randint(0, 4, (1024, 1), dtype=torch.int32, device='cuda')
        torch.cuda.synchronize()
        obs, rew, done, info = env.step(acts)
    env.sync()
    assert not done
    np.testing.assert_array_equal(obs['mplight'].cpu().numpy(), env.sim.read('mplight'))
    np.testing.assert_array_equal(rew['pressure'].cpu().numpy(), env.sim.read('pressure'))
    assert float(rew['wait'].min()) < 0
    env.close()


@pytest.mark.parametrize('side_stream', [False, True])
def test_vec_env_is_ordered_with_torch_stream(side_stream):
    """Actions produced by torch kernels and observations consumed by torch kernels need no host synchronisation:
    VecMultiSignal launches on torch's current stream (the default stream is passed as hipStreamLegacy, because a
    0 handle means 'the handle's own stream' in the C ABI).  An unsynchronised run must equal a fully
    synchronised one."""
    import torch
    from resco_amd.multi_signal import VecMultiSignal
    n, steps = 512, 30
    results = []
    for synced in (True, False):
        env = VecMultiSignal('cologne3', n, states=('mplight',), rewards=('wait',), seed=11)
        S = env.n_signals
        nact = torch.as_tensor(env.n_actions, device='cuda')
        gen = torch.Generator(device='cuda').manual_seed(5)
        ctx = torch.cuda.stream(torch.cuda.Stream()) if (side_stream and not synced) else torch.cuda.stream(torch.cuda.current_stream())
        with ctx:
            obs = env.reset()
            acc = torch.zeros(n, S, device='cuda')
            big = torch.randn(2048, 2048, device='cuda', generator=gen)
            for k in range(steps):
                if synced:
                    torch.cuda.synchronize()
                # a slow producer in front of the actions widens the window a race would need
                slow = (big @ big).sum() * 0.0
                u = torch.rand(n, S, device='cuda', generator=gen) + slow
                a = torch.minimum((u * nact).long(), nact - 1).to(torch.int32)
                if synced:
                    torch.cuda.synchronize()
                obs, rew, done, _ = env.step(a)
                if synced:
                    torch.cuda.synchronize()
                acc += obs['mplight'].float().sum(-1) + rew['wait']        # consumer right behind the kernel
            torch.cuda.current_stream().synchronize()
        results.append((acc.cpu().numpy(), env.sim.read('mplight'), env.sim.stats()['inserted']))
        env.close()
    np.testing.assert_array_equal(results[0][0], results[1][0])
    np.testing.assert_array_equal(results[0][1], results[1][1])
    np.testing.assert_array_equal(results[0][2], results[1][2])


def test_idqn_rollout_on_fp16_observations():
    """BASELINE config 5 interface: the batched IDQN consumes drq_norm_f16 directly; actions stay on device."""
    import torch
    from resco_amd.agents.idqn_rollout import BatchedIDQN
    from resco_amd.multi_signal import VecMultiSignal
    env = VecMultiSignal('ingolstadt21', 256, states=('drq_norm_f16', 'drq_norm'), rewards=('wait_norm',), seed=4)
    net = BatchedIDQN.from_scenario(env.scenario, dtype=torch.float16, device='cuda')
    mods = net.init_like_reference(seed=1)
    obs = env.reset()               # launches ride torch's current stream: no explicit synchronisation needed
    for k in range(25):
        a = net.act(obs['drq_norm_f16'], epsilon=0.2)
        assert a.dtype == torch.int32 and a.is_cuda
        obs, rew, done, _ = env.step(a)
    torch.cuda.synchronize()
    # the fp16 padded tensor is the fp32 drq_norm re-laid out per signal
    sc = env.scenario
    h = obs['drq_norm_f16'].float().cpu().numpy()
    f = obs['drq_norm'].cpu().numpy()
    for s in (0, 13, 20):
        o0, o1 = sc.sig_obs_start[s], sc.sig_obs_start[s + 1]
        np.testing.assert_allclose(h[:, s, :o1 - o0], f[:, o0:o1], rtol=2e-3, atol=2e-3)
    # Q-values of the batched fp16 network vs the per-signal fp32 reference-architecture modules
    q = net(obs['drq_norm_f16']).float().cpu()
    s, L, A = 13, net.lanes[13], net.actions[13]
    ref = mods[s](torch.from_numpy(h[:, s, :L]).unsqueeze(1))
    np.testing.assert_allclose(q[:, s, :A].detach().numpy(), ref.detach().numpy(), rtol=5e-2, atol=5e-2)
    assert float(rew['wait_norm'].min()) >= -4.0
    env.close()


def test_idqn_training_loop_on_device():
    """Replay ring + batched DQN update driven by the simulator's zero-copy tensors (tools/idqn_train.py in
    small): transitions land in the ring as the kernel produced them, the update runs and changes the weights."""
    import torch
    from resco_amd.agents.idqn_learn import BatchedDQNLearner, DeviceReplay
    from resco_amd.agents.idqn_rollout import BatchedIDQN
    from resco_amd.multi_signal import VecMultiSignal
    n = 64
    env = VecMultiSignal('cologne3', n, states=('drq_norm_f16',), rewards=('wait_norm',), seed=2)
    net = BatchedIDQN.from_scenario(env.scenario, dtype=torch.float32, device='cuda')
    net.init_like_reference(seed=0)
    w0, w0_fc2 = net.fc3_w.detach().clone(), net.fc2_w.detach().clone()
    learner = BatchedDQNLearner(net, batch_size=32, target_update=10)
    replay = DeviceReplay(16, n, env.n_signals, net.lmax, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(0)
    obs = env.reset()['drq_norm_f16']
    seen = []
    for k in range(24):
        a = net.act(obs, epsilon=0.5, generator=gen)
        replay.stage(obs)
        o, r, done, _ = env.step(a)
        replay.commit(a, r['wait_norm'], done)
        seen.append((o['drq_norm_f16'].clone(), a.clone(), r['wait_norm'].clone()))
        if learner.n_updates == 0 and len(replay) >= 32 and getattr(learner, '_graph', None) is None:
            learner.capture_update(replay)      # [sample -> loss -> backward -> Adam] replays as one HIP graph from here on
            w_cap = net.fc2_w.detach().clone()
            assert torch.equal(w_cap, w0_fc2)                                # capture leaves the weights where they were
        loss = learner.observe_step(replay, gen)
        obs = o['drq_norm_f16']
    torch.cuda.synchronize()
    assert learner.n_updates == 23 and torch.isfinite(loss) and learner._graph is not None
    assert not torch.equal(w_cap, net.fc2_w.detach())                       # the replayed graph really updates the weights
    assert not torch.equal(w0, net.fc3_w.detach())
    # the ring holds the last 16 steps; slot (k mod 16) = what the agents saw / did / got at step k
    for k in (8, 15, 23):
        i = k % 16
        assert torch.equal(replay.act[i].int(), seen[k][1]) and torch.equal(replay.rew[i], seen[k][2])
        assert torch.equal(replay.obs[(i + 1) % 16] if k < 23 else seen[k][0], seen[k][0])      # successor obs
    assert float(replay.rew.min()) >= -4.0 and float(replay.rew.max()) <= 0.0
    env.close()


@pytest.mark.parametrize('map_name,n', [('ingolstadt21', 200), ('cologne8', 70), ('cologne1', 64)])
def test_fused_idqn_policy_matches_torch_reference(map_name, n):
    """rs_idqn_act (conv features computed inside the MFMA loop of fc1, fp16 operands / fp32 accumulate) against
    the fp32 PyTorch forward of the same reference-architecture networks on the simulator's fp16 observations.
    Tolerance 3e-2 absolute on Q (|Q| ~ 0.1-1, K = 64*H*4 fp16 products); greedy actions agree wherever the top two
    Q-values are further apart than that; epsilon = 1 draws follow the counter hash."""
    import torch
    from oracle_batch import murmur_hash
    from resco_amd.agents.idqn_fused import FusedIDQN
    from resco_amd.agents.idqn_rollout import BatchedIDQN
    from resco_amd.multi_signal import VecMultiSignal
    env = VecMultiSignal(map_name, n, states=('drq_norm_f16',), rewards=('wait_norm',), seed=3)
    net = BatchedIDQN.from_scenario(env.scenario, dtype=torch.float32, device='cuda')
    net.init_like_reference(seed=5)
    with torch.no_grad():                       # make the outputs less degenerate than a fresh initialisation
        net.fc3_b.add_(0.1 * torch.randn_like(net.fc3_b))
    fused = FusedIDQN(net, seed=11)
    obs = env.reset()['drq_norm_f16']
    for k in range(12):
        env.act_random(k)
        obs = env.step(None)[0]['drq_norm_f16']
    acts, q = fused.act(obs, epsilon=0.0, want_q=True)
    torch.cuda.synchronize()
    ref = net(obs).float()
    S = env.n_signals
    for s_ in range(S):
        A = net.actions[s_]
        np.testing.assert_allclose(q[:, s_, :A].cpu().numpy(), ref[:, s_, :A].detach().cpu().numpy(), atol=3e-2, rtol=0)
        assert torch.isinf(q[:, s_, A:]).all()
    top2 = ref.masked_fill(~net.action_mask, float('-inf')).topk(2, dim=-1).values
    clear = (top2[..., 0] - top2[..., 1]) > 6e-2
    assert clear.float().mean() > 0.3
    assert torch.equal(acts[clear].long(), ref.argmax(-1)[clear])
    # epsilon = 1: every action is the hashed draw  hash(seed ^ 0x1D0A17; env, signal, step, 1) mod n_actions
    a1 = fused.act(obs, epsilon=1.0, step_key=77).clone().cpu().numpy()
    for m in (0, n // 2, n - 1):
        for s_ in (0, S - 1):
            assert a1[m, s_] == murmur_hash(11 ^ 0x1D0A17, m, s_, 77, 1) % net.actions[s_]
    # the same draw with epsilon / step key taken from device memory (the HIP-graph replay path), written straight
    # into the simulator's action buffer
    dyn = torch.tensor([np.float32(1.0).view(np.int32), 77], dtype=torch.int32, device='cuda')
    a2 = fused.act(obs, epsilon=0.0, step_key=0, dyn=dyn, out=env.tensor('actions'))
    assert a2.data_ptr() == env.tensor('actions').data_ptr() and np.array_equal(a2.cpu().numpy(), a1)
    env.step(None)
    assert np.array_equal(env.sim.read('actions'), a1)
    # weights changed by a learner are re-packed on the device: same result as a fresh host-side packing
    with torch.no_grad():
        for prm in net.parameters():
            prm.add_(0.05 * torch.randn_like(prm) * (prm != 0))
    fused.refresh_on_device()
    q_dev = fused.act(obs, want_q=True)[1].clone()
    fused.refresh()
    q_host = fused.act(obs, want_q=True)[1]
    torch.cuda.synchronize()
    assert torch.equal(q_dev, q_host) and not torch.allclose(q_dev[:, 0, :2], q[:, 0, :2])
    env.close()


def test_group_step_idqn_and_static_agents_on_the_device():
    """rs_group_step: one call through the ABI launches [agent, step] for every pipe of a GPU.  (i) random policy: two pipes
    through the group == the single batch through rs_act_random + rs_step; (ii) IDQN, greedy (epsilon 0): the group == per pipe
    rs_idqn_act + rs_step; (iii) IDQN with exploration: the draws are keyed by the GLOBAL environment index, so two pipes through
    the group == one handle with the whole batch through the group; several steps per call"""
    import torch
    from resco_amd.agents.idqn_fused import FusedIDQN
    from resco_amd.agents.idqn_rollout import BatchedIDQN
    from resco_amd.sim import BatchedSim, SimGroup
    sc = load_scenario('cologne8')
    n = 96
    mk = lambda cnt, base: BatchedSim(sc, cnt, seed=4, env_base=base)
    # (i)
    whole, pipes = mk(n, 0), [mk(n // 2, 0), mk(n // 2, n // 2)]
    grp = SimGroup(pipes)
    for k in range(0, 20, 5):
        for j in range(5):
            whole.act_random(k + j)
            whole.step(None)
        grp.step('random', step_key=k, n_steps=5)
    grp.sync()
    for name in ('veh_pos', 'veh_lane', 'phase', 'mplight', 'actions', 'drq_norm_f16'):
        np.testing.assert_array_equal(whole.read(name), np.concatenate([p.read(name) for p in pipes]))
    # (ii) + (iii)
    net = BatchedIDQN.from_scenario(sc, dtype=torch.float16, device='cuda')
    net.init_like_reference(seed=5)
    pol = FusedIDQN(net, seed=9)
    solo = [mk(n // 2, 0), mk(n // 2, n // 2)]
    for k in range(20):                         # the same history as the pipes of (i)
        for s_ in solo:
            s_.act_random(k)
            s_.step(None)
    for k in range(6):
        for s_ in solo:
            s_.sync()
            # (greedy for three steps, then exploring: the per-pipe call draws by the global environment index too -- env_base)
            pol.act(s_.tensor('drq_norm_f16'), epsilon=0.0 if k < 3 else 0.4, step_key=k, out=s_.tensor('actions'), env_base=s_.env_base)
            torch.cuda.synchronize()
            s_.step(None)
        grp.step('idqn', step_key=k, policy=pol._h, epsilon=0.0 if k < 3 else 0.4, seed=9)
    grp.sync()
    for s_ in solo:
        s_.sync()
    for name in ('veh_pos', 'phase', 'actions'):
        np.testing.assert_array_equal(np.concatenate([p.read(name) for p in solo]), np.concatenate([p.read(name) for p in pipes]))
    whole2 = mk(n, 0)
    g1, g2 = SimGroup([whole2]), SimGroup([mk(n // 2, 0), mk(n // 2, n // 2)])
    for g in (g1, g2):
        g.step('random', step_key=0, n_steps=8)
        g.step('idqn', step_key=8, n_steps=6, policy=pol._h, epsilon=0.5, epsilon_step=-0.05, seed=9)
        g.sync()
    a1 = whole2.read('actions')
    a2 = np.concatenate([p.read('actions') for p in g2.sims])
    np.testing.assert_array_equal(a1, a2)
    np.testing.assert_array_equal(whole2.read('veh_pos'), np.concatenate([p.read('veh_pos') for p in g2.sims]))
    assert len(np.unique(a1[:, 0])) > 1
    with pytest.raises(RuntimeError, match='switched off'):
        g1.sims[0].set_outputs(('drq_norm',))
        g1.step('idqn', policy=pol._h)
    for s_ in [whole, whole2] + pipes + solo + g2.sims:
        s_.close()


def test_fused_policy_sampling_mode_follows_the_softmax():
    """rs_idqn_act mode 1 (IPPO head on the same trunk): actions are drawn from softmax(logits) with the counter
    hash; over many environments x step keys the empirical action frequencies match the mean probabilities the
    PyTorch forward gives (4 sigma), the draw is a pure function of (seed, env, signal, step key), and an acting /
    PPO-update / device re-pack cycle runs."""
    import torch
    from resco_amd.agents.idqn_fused import FusedIDQN
    from resco_amd.agents.ippo import BatchedIPPO, BatchedPPOLearner
    from resco_amd.multi_signal import VecMultiSignal
    n = 1024
    env = VecMultiSignal('cologne8', n, states=('drq_norm_f16',), rewards=('wait_norm',), seed=1)
    net = BatchedIPPO.from_scenario(env.scenario, dtype=torch.float32, device='cuda')
    net.init_like_reference(seed=2)
    with torch.no_grad():
        net.fc3_w.mul_(40.0)                    # the fresh 1e-2 head is almost uniform: make the policy informative
    policy = FusedIDQN(net, seed=5)
    obs = env.reset()['drq_norm_f16']
    for k in range(10):
        env.act_random(k)
        obs = env.step(None)[0]['drq_norm_f16']
    p = torch.softmax(net(obs)[0].float(), -1).detach()                  # [n, S, Amax]
    S, A = env.n_signals, net.amax
    counts = torch.zeros(S, A, device='cuda')
    keys = 24
    for key in range(keys):
        a = policy.act(obs, step_key=key, sample=True).long()
        assert all(int(a[:, s].max()) < net.actions[s] for s in range(S))
        counts.scatter_add_(1, a.t().contiguous(), torch.ones(S, n, device='cuda'))
    a_again = policy.act(obs, step_key=keys - 1, sample=True).long()
    assert torch.equal(a, a_again)
    freq = (counts / (n * keys)).cpu().numpy()
    mean_p = p.mean(0).cpu().numpy()
    sigma = np.sqrt(np.maximum(mean_p * (1 - mean_p), 1e-4) / (n * keys))
    assert np.all(np.abs(freq - mean_p) < 4 * sigma + 2e-3), np.abs(freq - mean_p).max()
    assert mean_p.max() > 0.4                                             # the test policy is not uniform
    # one short acting / learning cycle
    learner = BatchedPPOLearner(net, minibatch=2048, epochs=2)
    T = 4
    ob = torch.zeros(T, n, S, net.lmax, 5, dtype=torch.float16, device='cuda')
    ac = torch.zeros(T, n, S, dtype=torch.int32, device='cuda')
    rw = torch.zeros(T, n, S, device='cuda')
    for t in range(T):
        ob[t].copy_(obs)
        policy.act(obs, step_key=100 + t, out=env.tensor('actions'), sample=True)
        o, r, _, _ = env.step(None)
        ac[t].copy_(env.tensor('actions'))
        rw[t].copy_(r['wait_norm'])
        obs = o['drq_norm_f16']
    w0 = net.fc2_w.detach().clone()
    loss = learner.update(ob, ac, rw, torch.zeros(T, dtype=torch.bool, device='cuda'), obs)
    policy.refresh_on_device()
    policy.act(obs, step_key=999, sample=True)
    torch.cuda.synchronize()
    assert torch.isfinite(loss) and not torch.equal(w0, net.fc2_w.detach()) and learner.n_updates == 2 * 2
    env.close()


# ------------------------------------------------------------------------------------------------ model fidelity bands
# The reference ships no tests, but it holds measured results of its own SUMO runs: per-episode averages in
# resco_benchmark/utils/avg_timeLoss.py, avg_duration.py, avg_waitingTime.py, avg_queue.py (produced by utils/readXML.py:16-114
# and utils/readCSV.py:30-80).  tests/golden/make_ref_bands.py reduces them (build container) to tests/golden/ref_bands.json:
#   FIXED / MAXWAVE / MAXPRESSURE   delay = median over the published episodes of avg_timeLoss.py (:49-51, 60-62, 83-85, ...)
#   STOCHASTIC                      delay / duration / waiting / queue of EPISODE 1 of the IDQN rows of the four arrays: epsilon
#                                   decays linearly from 1 over 80 episodes (agents/pfrl_dqn.py:65-70, main.py:91-92), so episode 1
#                                   is the uniform random policy seen through IDQN's 200 m detectors
#   free_flow_residual              duration - delay of the trained IDQN episodes: the travel time of the routes at the speed
#                                   limits -- independent of the controller, it pins routing, lane lengths and speed limits
# The dynamics are this build's own model (PARITY-UNPINNED vs SUMO); these are the only reference-held
# numbers that depend on them.  ONE band: +-35 % of the reference figure.  What is known to lie outside it is listed in KNOWN_GAPS
# and tested as an expected failure -- not banded around:
#  * ingolstadt21 FIXED (1.73 x; 2.1 x before the router's minor-link penalty, round 5): profiles/r04_ingolstadt21_approaches.txt,
#    profiles/r05_route_sensitivity.txt.  40 % of the delay sits on ONE movement -- 490 veh/h on one lane of -201201945#0.78 with
#    20 s of green per 86 s, turning left 12 m later through gneJ257's permissive internal junction -- and 22 % on the S approach.
#    Both are at or beyond their physical capacity under the net's own programme in any Krauss simulation; none of the 535 trips of
#    that movement has an alternative route within 2 % of its best (371 within 10 %), so the assignment does not relieve it.
#  * ingolstadt21 MAXWAVE / MAXPRESSURE as configured (5.4 x / 3.9 x): the reference's valid_acts['243641585'] = {2: 0, 4: 1, 7: 2}
#    maps the wave of the S approach (841 trips/h) to green 0 = 'rGgG', in which that approach is red (phase order = the tlLogic's
#    file order exactly as multi_signal.py:52-59 extracts it): 0 of its 841 trips arrive in the hour.  No simulator reaches the
#    published 69.6 s with that mapping (the vehicles stored on the approach alone are worth 59 s per tripinfo entry); with the
#    entry rotated to {4: 0, 7: 1, 2: 2} MAXWAVE gives 1.11 x (default band), MAXPRESSURE 1.55 x.
#  * STOCHASTIC on cologne3 (0.35 x): two pairs of junctions 10 m and 13 m apart; whether a lane change is possible on those two
#    edges decides the cell (RM_MIN_LC_LEN 5 m: 0.35 x, 12.5 m: 2.9 x); the reference's own figure is the mean of trials of which one
#    gridlocked (279 s +- 202 s over the trials; IPPO's and MPLight's first episodes: 185 s and 132 s).
#  * cologne3 / cologne8 MAXPRESSURE and cologne3 MAXWAVE are bimodal in the reference itself (means 162 / 48 / 91 s against
#    medians 28 / 30 / 22 s): medians are compared.
# What trafficlight.setPhase leaves behind is a parameter (rs_params.tls_hold / tls_expiry, include/resco_sim.h): every cell is measured
# and asserted in BOTH modes.  tls_expiry = 1 (the library's default since round 6: the phase runs for its programme duration and the
# programme continues -- SUMO's documented setPhase): the random-policy figures of cologne1 / cologne8 fall below the band (a 6 s green
# chosen for a 10 s step hands its 7th second to the next phase of the list -- cologne1's W-E greens gain 40 % of capacity that way),
# ingolstadt21's are inside it.  tls_expiry = 0 (round 5's default, the calibration variant): cologne1 / cologne8 inside, ingolstadt21's
# random policy at 1.13-1.31 x.  test_tls_expiry_evidence holds the two answers against each other; profiles/r06_reference_bands_both_modes.txt
# is the table.  The held-out counterpart (figures no model constant was tuned on) is tests/test_gpu_heldout.py, both modes as well.
BAND = (0.65, 1.35)
# Cells that are KNOWN to be outside the default band, per mode (key: tls_expiry).  They are NOT part of the pass criterion of
# test_reference_result_bands (the default band is the only one); test_reference_result_known_gaps asserts the default band for each of
# them as an expected failure (xfail, non-strict: a model that closes a gap turns it into an XPASS).  The value is the ratio this build
# measures (64 environments, seed 0: the simulation is deterministic, every box gives the same figure): test_reference_result_bands holds
# each of them within +-15 % of it -- a TWO-SIDED drift guard, so that a model change that moves a gap either way is seen.
_GAPS_BOTH = {
    ('ingolstadt21', 'FIXED', 'delay'): 1.73,           # (the net's own programme: no setPhase, the same in both modes)
}
KNOWN_GAPS = {
    0: {**_GAPS_BOTH, **{
        ('ingolstadt21', 'MAXWAVE', 'delay'): 5.42,         # as configured: the S approach of TLS 243641585 is never served
        ('ingolstadt21', 'MAXPRESSURE', 'delay'): 4.00,     #   (valid_acts maps its wave to a phase in which it is red); unreachable on SUMO too
        ('ingolstadt21', 'MAXPRESSURE*', 'delay'): 1.55,    # with the entry rotated (MAXWAVE*: 1.11 x, inside the default band)
        ('cologne3', 'STOCHASTIC', 'delay'): 0.35, ('cologne3', 'STOCHASTIC', 'duration'): 0.39,
        ('cologne3', 'STOCHASTIC', 'waiting'): 0.24,
    }},
    1: {**_GAPS_BOTH, **{
        ('ingolstadt21', 'MAXWAVE', 'delay'): 5.37,
        ('ingolstadt21', 'MAXPRESSURE', 'delay'): 3.93,
        ('ingolstadt21', 'MAXPRESSURE*', 'delay'): 1.52,
        ('cologne3', 'STOCHASTIC', 'delay'): 0.25, ('cologne3', 'STOCHASTIC', 'duration'): 0.32,
        ('cologne3', 'STOCHASTIC', 'waiting'): 0.15, ('cologne3', 'STOCHASTIC', 'queue'): 0.42,
        ('cologne1', 'STOCHASTIC', 'duration'): 0.58, ('cologne1', 'STOCHASTIC', 'waiting'): 0.46, ('cologne1', 'STOCHASTIC', 'queue'): 0.51,
        ('cologne8', 'STOCHASTIC', 'delay'): 0.60, ('cologne8', 'STOCHASTIC', 'waiting'): 0.56, ('cologne8', 'STOCHASTIC', 'queue'): 0.59,
    }},
}
DRIFT = 0.15
MAXD = {'FIXED': 200, 'MAXWAVE': 50, 'MAXPRESSURE': 200, 'STOCHASTIC': 200}


def _ref_bands():
    with open(os.path.join(os.path.dirname(os.path.abspath(__file__)), 'golden', 'ref_bands.json')) as f:
        return json.load(f)


def _episode_metrics(sc, policy, n_envs=64, seed=0, tls_expiry=0):
    """one whole episode of `policy` on the device for n_envs environments: the reference's four per-episode figures"""
    from resco_amd.sim import BatchedSim
    sim = BatchedSim(sc, n_envs, seed=seed, max_distance=MAXD[policy.rstrip('*')], fixed_program=1 if policy == 'FIXED' else 0,
                     tls_expiry=tls_expiry)
    q = np.zeros(n_envs)
    for k in range(360):
        if policy.startswith('MAX'):
            sim.act_maxwave(1 if policy.startswith('MAXPRESSURE') else 0)
        elif policy == 'STOCHASTIC':
            sim.act_random(k)
        sim.step(None)
        q += sim.read('queue_sum').sum(axis=1) / (sc.n_signals + 1.0)      # utils/readCSV.py:32-46
    m = sim.trip_metrics()
    m['queue'] = q / 360.0
    sim.close()
    return m


_BAND_CACHE = {}


def _band_cells(name, mode):
    """every (policy, metric, value, reference figure) cell of one map under tls_expiry = mode: 64 environments x one whole episode per
    controller (the FIXED programme never calls setPhase: measured once)"""
    if (name, mode) in _BAND_CACHE:
        return _BAND_CACHE[(name, mode)]
    import copy
    sc = load_scenario(name)
    ref = _ref_bands()[name]
    cells, med = [], {}
    for policy in ('FIXED', 'MAXWAVE', 'MAXPRESSURE', 'STOCHASTIC'):
        if policy == 'FIXED' and (name, 'FIXED') in _BAND_CACHE:
            med[policy] = _BAND_CACHE[(name, 'FIXED')]
        else:
            m = _episode_metrics(sc, policy, tls_expiry=mode)
            med[policy] = {k: float(np.median(v)) for k, v in m.items()}
            if policy == 'FIXED':
                _BAND_CACHE[(name, 'FIXED')] = med[policy]
        cells.append((policy, 'delay', med[policy]['delay'], ref[policy]['delay']))
        if policy == 'STOCHASTIC':
            for metric in ('duration', 'waiting', 'queue'):
                cells.append((policy, metric, med[policy][metric], ref[policy][metric]))
    if name == 'ingolstadt21':
        sc2 = copy.copy(sc)
        sc2.valid_acts = dict(sc.valid_acts)
        sc2.valid_acts['243641585'] = {4: 0, 7: 1, 2: 2}
        for policy in ('MAXWAVE*', 'MAXPRESSURE*'):
            m = _episode_metrics(sc2, policy, tls_expiry=mode)
            med[policy] = {k: float(np.median(v)) for k, v in m.items()}
            cells.append((policy, 'delay', med[policy]['delay'], ref[policy.rstrip('*')]['delay']))
    # The travel time of the routes at the speed limits: duration - (timeLoss + departDelay), the reference's figure from
    # its trained IDQN episodes.  Comparable when (nearly) all trips finish: under the map's best static controller, or --
    # ingolstadt21, which no static controller keeps fluid -- with every fourth trip only (FIXED programme).  Band +-10 %.
    if name != 'ingolstadt21':
        best = min(('MAXWAVE', 'MAXPRESSURE'), key=lambda p: med[p]['delay'])
        cells.append((best, 'free_flow', med[best]['duration'] - med[best]['delay'], ref['free_flow_residual']))
    else:
        sc3 = copy.copy(sc)
        sc3.arrays = dict(sc.arrays)
        keep = np.arange(0, sc.n_trips, 4)
        for f in ('trip_depart', 'trip_route', 'trip_vtype'):
            sc3.arrays[f] = np.ascontiguousarray(sc.arrays[f][keep])
        cum = np.zeros(sc.horizon + 1, np.int64)
        np.add.at(cum, sc3.arrays['trip_depart'][sc3.arrays['trip_depart'] <= sc.horizon], 1)
        sc3.arrays['trips_cum'] = np.cumsum(cum).astype(np.int32)
        if (name, 'FIXED/4') not in _BAND_CACHE:
            _BAND_CACHE[(name, 'FIXED/4')] = {k: float(np.median(v)) for k, v in _episode_metrics(sc3, 'FIXED').items()}
        m = _BAND_CACHE[(name, 'FIXED/4')]
        cells.append(('FIXED/4', 'free_flow', m['duration'] - m['delay'], ref['free_flow_residual']))
    _BAND_CACHE[(name, mode)] = cells
    return cells


@pytest.mark.parametrize('mode', [1, 0], ids=['expiry', 'hold'])
@pytest.mark.parametrize('name', ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21'])
def test_reference_result_bands(name, mode):
    """Both answers to what setPhase leaves behind (mode = tls_expiry; 1 is the library's default).  64 environments x one whole episode
    of FIXED / MAXWAVE / MAXPRESSURE / STOCHASTIC on the device against every figure the reference holds for them (delay for all four, duration / waitingTime / queue for the random policy, the free-flow residual
    for the routes).  ONE pass criterion: the default band (+-35 %; +-10 % for the free-flow residual).  The cells listed in
    KNOWN_GAPS are reported and held within +-15 % of their recorded ratio here (a two-sided drift guard);
    test_reference_result_known_gaps holds them to the default band as expected failures."""
    failures, lines = [], []
    for policy, metric, value, target in _band_cells(name, mode):
        ratio = value / target
        key = (name, policy, metric)
        band = (0.9, 1.1) if metric == 'free_flow' else BAND
        if key in KNOWN_GAPS[mode]:
            rec = KNOWN_GAPS[mode][key]
            lines.append('band expiry=%d ' % mode + '%-12s %-12s %-9s %8.2f / %8.2f = %.2f  KNOWN GAP (default band [%.2f, %.2f]; drift guard: %.2f +-%d %%)' % (
                name, policy, metric, value, target, ratio, band[0], band[1], rec, round(100 * DRIFT)))
            if not rec * (1 - DRIFT) <= ratio <= rec * (1 + DRIFT):
                failures.append(lines[-1])
            continue
        lines.append('band expiry=%d ' % mode + '%-12s %-12s %-9s %8.2f / %8.2f = %.2f  [%.2f, %.2f]' % (name, policy, metric, value, target, ratio, band[0], band[1]))
        if not band[0] <= ratio <= band[1]:
            failures.append(lines[-1])
    print('\n'.join(lines))
    assert not failures, failures


@pytest.mark.parametrize('cell', [(mode,) + k for mode in (1, 0) for k in sorted(KNOWN_GAPS[mode])], ids=lambda c: ('expiry-' if c[0] else 'hold-') + '-'.join(c[1:]))
@pytest.mark.xfail(strict=False, reason='known fidelity gap of the own microsimulation model against the reference-held SUMO figures (DESIGN.md section 2)')
def test_reference_result_known_gaps(cell):
    mode, name, policy, metric = cell
    for p, m, value, target in _band_cells(name, mode):
        if (p, m) == (policy, metric):
            assert BAND[0] <= value / target <= BAND[1], (cell, value, target, value / target)
            return
    raise AssertionError('no such cell: %r' % (cell,))


def test_tls_expiry_evidence():
    """What does trafficlight.setPhase leave behind (rs_params.tls_hold; tls_expiry in the Python layer)?  SUMO documents that the phase
    runs for its programme duration and the programme then continues [SUMO-K]: the library's default since round 6.  The reference never
    resets a duration (traffic_signal.py:176-187), so a 6 s green chosen for a 10 s step hands its 7th second to the next phase of the
    list.  No SUMO binary is at hand, but the reference holds 20 figures that depend on the rule: delay / duration / waitingTime / queue
    of a uniformly random policy on five maps with 6 s greens (episode 1 of its IDQN runs, epsilon >= 0.9875).  Both answers on the
    device, 64 environments x one episode each (table: profiles/r06_tls_expiry_bands.txt).  RECORDED, as what it is -- a property of
    this build's own traffic model, not a pin: WITHOUT expiry (round 5's default, the calibration variant tls_hold = 1) at least 15 of
    the 20 figures are closer to the reference's than with it, and |log ratio| summed over the 20 is at most 70 % of what expiry
    gives.  A model change that closes the gap of the documented rule shows up here as a failure -- and is welcome."""
    ref = _ref_bands()
    lines, closer, err = [], 0, [0.0, 0.0]
    for name in ('cologne1', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21'):
        sc = load_scenario(name)
        m = [{k: float(np.median(v)) for k, v in _episode_metrics(sc, 'STOCHASTIC', tls_expiry=x).items()} for x in (0, 1)]
        for metric in ('delay', 'duration', 'waiting', 'queue'):
            t = ref[name]['STOCHASTIC'][metric]
            r = [m[0][metric] / t, m[1][metric] / t]
            closer += abs(np.log(r[0])) < abs(np.log(r[1]))
            err[0] += abs(np.log(r[0]))
            err[1] += abs(np.log(r[1]))
            lines.append('expiry %-12s STOCHASTIC %-9s reference %8.2f   phase stays %8.2f (%.2f)   phase expires %8.2f (%.2f)' % (
                name, metric, t, m[0][metric], r[0], m[1][metric], r[1]))
    lines.append('expiry closer to the reference without expiry: %d of 20; sum |log ratio|: %.2f without, %.2f with' % (closer, err[0], err[1]))
    print('\n'.join(lines))
    assert closer >= 15 and err[0] <= 0.7 * err[1]


# ------------------------------------------------------------------------------------------------ BASELINE configs 4 and 5
def _episode_invariants(sim, sc, k):
    st = sim.stats()
    env = sim.read('env')
    assert (env[:, 0] == 10 * (k + 1)).all()
    assert (st['inserted'] == st['arrived'] + st['active']).all() and (env[:, 1] == st['inserted']).all()
    lane, pos, spd = sim.read('veh_lane'), sim.read('veh_pos'), sim.read('veh_speed')
    act = lane != 0xFFFF
    assert (act.sum(axis=1) == st['active']).all() and (st['active'] <= sc.capacity).all()
    assert (pos[act] >= 0).all() and (pos[act] <= sc.lane_len[lane[act]] + 1e-3).all()
    assert (spd[act] >= 0).all() and (spd[act] <= 60.0).all()
    return st


def _oracle_replay(sc, seed, env_index, actions, max_distance=200.0):
    from oracle.pyoracle import OracleEnv
    o = OracleEnv(sc, env_index=env_index, seed=seed, sigma=-1.0, speed_dev=1, max_distance=max_distance)
    o.observe()
    for a in actions:
        o.step(a)
    return o


def test_step_sim_and_output_mask_on_the_device():
    """rs_step_sim (simulationStep() only: the Signal objects are not observed, multi_signal.py:102-105) and rs_set_outputs
    (only the requested per-lane / per-movement buffers are written) through the C ABI against the oracle."""
    from resco_amd.sim import BatchedSim
    from oracle.pyoracle import OracleEnv
    sc = load_scenario('ingolstadt7')
    n, seed = 3, 6
    sim = BatchedSim(sc, n, seed=seed)
    orcs = [OracleEnv(sc, env_index=e, seed=seed, sigma=-1.0, speed_dev=1) for e in range(n)]
    for o in orcs:
        o.observe()

    def both(k):
        sim.act_random(k)
        sim.step(None)
        for e, o in enumerate(orcs):
            o.step(preroll_actions(sc, seed, e, k))

    for k in range(60):
        both(k)
    before = sim.outputs(INT_BUFS + FLT_BUFS)
    sim.step_sim(5)
    sim.step_sim(6)
    for o in orcs:
        for _ in range(11):
            o.tick()
    after = sim.outputs(INT_BUFS + FLT_BUFS)
    for b in INT_BUFS + FLT_BUFS:
        np.testing.assert_array_equal(before[b], after[b], err_msg=b)
    both(60)
    assert_env_equal(sim, orcs, 60)         # waiting times, arrivals and the departures collected over the quiet ticks
    old = sim.outputs(INT_BUFS + FLT_BUFS)
    sim.set_outputs(['drq_norm', 'mplight'])
    for k in range(61, 75):
        both(k)
    new = sim.outputs(INT_BUFS + FLT_BUFS)
    for e, o in enumerate(orcs):
        ref = o.outputs()
        for b in ('drq_norm', 'mplight', 'phase', 'wait', 'wait_norm', 'pressure', 'queue_sum', 'queue_max', 'arrivals', 'departures'):
            np.testing.assert_array_equal(new[b][e], ref[b], err_msg=b)
    for b in ('lane_agg', 'wave', 'mplight_full'):
        np.testing.assert_array_equal(new[b], old[b], err_msg=b)
    sim.set_outputs(None)
    both(75)
    assert_env_equal(sim, orcs, 75)
    sim.close()


def test_mail_flags_survive_every_launch_boundary():
    """Round 6: a plan reads its cooperation mailboxes only when its record is flagged; between launches the flags travel in
    RS_BUF_VEH_MAIL.  The same congested episode (ingolstadt21, the net's own programme) cut into launches of ONE tick each must end
    in the state of ten-tick launches, both equal to the oracle (which knows no flags); snapshot / restore carries the bitmap."""
    from resco_amd.sim import BatchedSim
    from oracle.pyoracle import OracleEnv
    sc = load_scenario('ingolstadt21')
    n = 3
    a = BatchedSim(sc, n, seed=4, fixed_program=1)
    b = BatchedSim(sc, n, seed=4, fixed_program=1)
    orcs = [OracleEnv(sc, env_index=e, seed=4, sigma=-1.0, speed_dev=1, fixed_program=1) for e in range(n)]
    for o in orcs:
        o.observe()
    mails, snap = 0, None
    for step in range(45):
        a.step(None)
        for i in range(10):
            b.step_sim(1)
            mails += int(np.unpackbits(b.read('veh_mail').view(np.uint8)).sum())
            if step == 30 and i == 4:
                snap = b.snapshot()                  # in the middle of an env-step, requests pending
        for o in orcs:
            for _ in range(10):
                o.tick()
    assert mails > 100, mails

    def same(x, y):
        lane = x.read('veh_lane')
        np.testing.assert_array_equal(lane, y.read('veh_lane'))
        live = lane != 0xFFFF
        for name in ('veh_pos', 'veh_speed', 'veh_cursor', 'veh_swait', 'veh_tloss', 'veh_coop', 'veh_cooplead', 'veh_coop_odd', 'veh_cooplead_odd'):
            np.testing.assert_array_equal(x.read(name)[live], y.read(name)[live], err_msg=name)
        np.testing.assert_array_equal(x.read('veh_mail'), y.read('veh_mail'))
        np.testing.assert_array_equal(x.read('env'), y.read('env'))

    same(a, b)
    for e, o in enumerate(orcs):
        v = o.vehicles()
        live = v['lane'] != 0xFFFF
        np.testing.assert_array_equal(a.read('veh_lane')[e], v['lane'])
        np.testing.assert_array_equal(a.read('veh_pos')[e][live], v['pos'][live])
    # back to the snapshot (tick 305 of the episode) and forward again in other launch sizes: the same end state
    b.restore(snap)
    b.step_sim(5)
    for step in range(31, 45):
        b.step_sim(7)
        b.step_sim(3)
    same(a, b)
    b.free_snapshot(snap)
    a.close(); b.close()


def test_edge_cases_capacity_invalid_actions_and_create_errors():
    """Through the C ABI: (1) the network full -- due trips wait in their lane's backlog, never dropped, bit-identical to the
    oracle; (2) action entries that are no phase index leave the signal alone; actions as a HOST array, a DEVICE tensor and the
    staged buffer give the same step; (3) rs_create refuses what it cannot run and says why."""
    import copy
    import torch
    from resco_amd.sim import BatchedSim
    from oracle.pyoracle import OracleEnv
    sc = copy.copy(load_scenario('cologne1'))
    sc.capacity = 64
    sim = BatchedSim(sc, 2, seed=1)
    orcs = [OracleEnv(sc, env_index=e, seed=1, sigma=-1.0, speed_dev=1) for e in range(2)]
    for o in orcs:
        o.observe()
    a = np.zeros((2, sc.n_signals), np.int32)
    for k in range(150):
        sim.step(a)
        for o in orcs:
            o.step(a[0])
    st = sim.stats()
    assert (st['active'] <= 64).all() and (st['pending'] > 0).all()
    assert_env_equal(sim, orcs, 149)
    sim.close()

    sc = load_scenario('cologne8')
    sims = [BatchedSim(sc, 2, seed=2) for _ in range(3)]
    orcs = [OracleEnv(sc, env_index=e, seed=2, sigma=-1.0, speed_dev=1) for e in range(2)]
    for o in orcs:
        o.observe()
    rng = np.random.default_rng(7)
    for k in range(24):
        a = np.stack([rng.integers(0, sc.tls_ngreen) for _ in range(2)]).astype(np.int32)
        if k % 3 == 1:
            a[0, k % sc.n_signals] = -1
        if k % 3 == 2:
            a[1, (k * 5) % sc.n_signals] = 99
        sims[0].step(a)                                                     # host array
        sims[1].step(torch.as_tensor(a, device='cuda'))                     # device tensor
        sims[2].tensor('actions').copy_(torch.as_tensor(a))                 # staged in RS_BUF_ACTIONS
        torch.cuda.synchronize()
        sims[2].step(None)
        for e, o in enumerate(orcs):
            o.step(a[e])
    for s_ in sims:
        assert_env_equal(s_, orcs, 23)
        s_.close()

    for kw, frag in ((dict(block_threads=100), 'multiple of 64'), (dict(block_threads=-1024), '128-VGPR'), (dict(device=99), 'hipSetDevice')):
        with pytest.raises(RuntimeError) as ei:
            BatchedSim(sc, 1, **kw)
        assert frag in str(ei.value), str(ei.value)
    bad = copy.copy(sc)
    bad.capacity = 100
    with pytest.raises(RuntimeError) as ei:
        BatchedSim(bad, 1)
    assert 'capacity' in str(ei.value)
    with pytest.raises(ValueError):
        BatchedSim(sc, 1, yellow_length=5)                                   # the yellow phases are compiled for 3 s


def test_config2_cologne1_1024_maxpressure_full_episode():
    """BASELINE config 2 at its size: cologne1 x 1024 environments x 360 env-steps with MaxPressure ON THE DEVICE
    (rs_act_maxwave(1), agents/maxpressure.py:13-18): invariants for every environment, and three sampled environments
    replayed on the oracle with the recorded actions -- lane aggregates, rewards.pressure (rewards.py:28-41), rewards.wait
    and the vehicles bit-identical at the end of the episode and half way through."""
    from resco_amd.sim import BatchedSim
    sc = load_scenario('cologne1')
    N, seed, picks = 1024, 2, [0, 511, 1023]
    sim = BatchedSim(sc, N, seed=seed)
    rec, mid = [], None
    for k in range(360):
        sim.act_maxwave(1)
        sim.sync()
        rec.append(sim.read('actions')[picks].copy())
        sim.step(None)
        if k in (179, 359):
            st = _episode_invariants(sim, sc, k)
        if k == 179:
            mid = {b: sim.read(b)[picks].copy() for b in ('lane_agg', 'pressure', 'wait', 'mplight')}
    assert (st['ticks'] == 3600).all() and np.median(st['arrived']) > 1900
    out = {b: sim.read(b) for b in ('lane_agg', 'mplight', 'pressure', 'wait', 'wait_norm', 'queue_sum', 'veh_pos', 'veh_speed', 'veh_lane')}
    for j, e in enumerate(picks):
        o = _oracle_replay(sc, seed, e, [r[j] for r in rec[:180]])
        for b in ('lane_agg', 'pressure', 'wait', 'mplight'):
            np.testing.assert_array_equal(mid[b][j], o.outputs()[b])
        for r in rec[180:]:
            o.step(r[j])
        ref, v = o.outputs(), o.vehicles()
        for b in ('lane_agg', 'mplight', 'pressure', 'wait', 'wait_norm', 'queue_sum'):
            np.testing.assert_array_equal(out[b][e], ref[b])
        np.testing.assert_array_equal(out['veh_lane'][e], v['lane'])
        live = v['lane'] != 0xFFFF
        np.testing.assert_array_equal(out['veh_pos'][e][live], v['pos'][live])
        np.testing.assert_array_equal(out['veh_speed'][e][live], v['speed'][live])
    sim.close()


def test_two_handles_two_threads_two_streams_equal_one_batch():
    """"No global state" (include/resco_sim.h): two handles on ONE device, each driven by its own host thread on its own
    HIP stream, reproduce the two halves of a single 2N-environment handle bit for bit (the way 8 GPUs are driven from 8
    threads or processes)."""
    import threading
    import torch
    from resco_amd.sim import BatchedSim
    sc = load_scenario('cologne8')
    n, seed, steps = 96, 11, 60
    whole = BatchedSim(sc, 2 * n, seed=seed)
    halves = [BatchedSim(sc, n, seed=seed, env_base=0), BatchedSim(sc, n, seed=seed, env_base=n)]
    for k in range(steps):
        whole.act_random(k)
        whole.step(None)
    whole.sync()
    errors = []

    def drive(h):
        try:
            stream = torch.cuda.Stream()
            for k in range(steps):
                h.act_random(k, stream=stream.cuda_stream)
                h.step(None, stream=stream.cuda_stream)
            stream.synchronize()
        except Exception as exc:        # noqa: BLE001
            errors.append(exc)

    threads = [threading.Thread(target=drive, args=(h,)) for h in halves]
    for t in threads:
        t.start()
    for t in threads:
        t.join()
    assert not errors, errors
    for b in ('lane_agg', 'mplight', 'pressure', 'wait', 'phase', 'veh_pos', 'veh_speed', 'veh_lane', 'veh_trip', 'stats', 'env'):
        w = whole.read(b)
        np.testing.assert_array_equal(w[:n], halves[0].read(b))
        np.testing.assert_array_equal(w[n:], halves[1].read(b))
    for h in halves + [whole]:
        h.close()


def test_config4_cologne8_per_gpu_share_full_episode():
    """BASELINE config 4 at its per-GPU size: cologne8 x 2048 environments x 360 env-steps with MaxPressure ON THE DEVICE
    (rs_act_maxwave): invariants for every environment, and two sampled environments replayed on the oracle with the
    recorded actions -- bit-identical at the end of the episode."""
    from resco_amd.sim import BatchedSim
    sc = load_scenario('cologne8')
    N, seed, picks = 2048, 4, [0, 2047]
    sim = BatchedSim(sc, N, seed=seed)
    rec = []
    for k in range(360):
        sim.act_maxwave(1)
        sim.sync()
        rec.append(sim.read('actions')[picks].copy())
        sim.step(None)
        if k in (179, 359):
            st = _episode_invariants(sim, sc, k)
    # (MaxPressure gridlocks cologne8 in a few environments -- the reference's own published episodes are bimodal too)
    assert (st['ticks'] == 3600).all() and np.median(st['arrived']) > 1900 and st['arrived'].min() > 300
    out = {b: sim.read(b) for b in ('lane_agg', 'mplight', 'pressure', 'veh_pos', 'veh_lane')}
    for j, e in enumerate(picks):
        o = _oracle_replay(sc, seed, e, [r[j] for r in rec])
        ref, v = o.outputs(), o.vehicles()
        for b in ('lane_agg', 'mplight', 'pressure'):
            np.testing.assert_array_equal(out[b][e], ref[b])
        np.testing.assert_array_equal(out['veh_lane'][e], v['lane'])
        live = v['lane'] != 0xFFFF
        np.testing.assert_array_equal(out['veh_pos'][e][live], v['pos'][live])
    sim.close()


def test_config5_ingolstadt21_idqn_rollout_full_episode():
    """BASELINE config 5 at its per-GPU size: ingolstadt21 x 1024 environments x 360 env-steps with the fused IDQN policy
    (rs_idqn_act, epsilon-greedy on the fp16 observation tensor) in the loop: invariants, the fp16 tensor against the
    ORACLE's drq_norm for sampled environments (replayed with the recorded actions), rewards equal to the oracle's."""
    import torch
    from resco_amd.agents.idqn_fused import FusedIDQN
    from resco_amd.agents.idqn_rollout import BatchedIDQN
    from resco_amd.sim import BatchedSim, torch_stream
    sc = load_scenario('ingolstadt21')
    N, seed, picks = 1024, 6, [0, 511, 1023]
    sim = BatchedSim(sc, N, seed=seed)
    net = BatchedIDQN.from_scenario(sc, dtype=torch.float32, device='cuda')
    net.init_like_reference(seed=5)
    pol = FusedIDQN(net, seed=1)
    obs = sim.tensor('drq_norm_f16')
    actions = sim.tensor('actions')
    rec = []
    for k in range(360):
        pol.act(obs, epsilon=0.3, step_key=k, out=actions)
        torch.cuda.synchronize()
        rec.append(actions[picks].cpu().numpy().copy())
        sim.step(None, stream=torch_stream())
        if k in (179, 359):
            torch.cuda.synchronize()
            st = _episode_invariants(sim, sc, k)
    assert (st['ticks'] == 3600).all() and st['arrived'].min() > 1000
    h = sim.read('drq_norm_f16').astype(np.float32)
    wn = sim.read('wait_norm')
    lmax = h.shape[2]
    for j, e in enumerate(picks):
        o = _oracle_replay(sc, seed, e, [r[j] for r in rec])
        ref = o.outputs()
        np.testing.assert_array_equal(wn[e], ref['wait_norm'])
        for s_ in range(sc.n_signals):
            o0, o1 = int(sc.sig_obs_start[s_]), int(sc.sig_obs_start[s_ + 1])
            want = ref['drq_norm'][o0:o1].astype(np.float16).astype(np.float32)        # the kernel rounds fp32 -> fp16
            np.testing.assert_array_equal(h[e, s_, :o1 - o0], want)
            assert (h[e, s_, o1 - o0:lmax] == 0).all()
    sim.close()


def test_numa_binding_of_a_rank_is_harmless():
    """bench.py pins every rank of an N > 1 run to the NUMA node of its GPU: on whatever box this runs the call must either
    bind to a non-empty subset of the allowed cores or change nothing"""
    import bench
    before = os.sched_getaffinity(0)
    try:
        node = bench.bind_to_gpu_numa_node(0)
        after = os.sched_getaffinity(0)
        assert after and after <= before
        assert (node is None and after == before) or (node is not None and len(after) >= 2)
    finally:
        os.sched_setaffinity(0, before)


def test_bench_runs_through_rccl_at_one_gpu():
    """bench.py with RESCO_BENCH_FORCE_DIST=1: the N > 1 code path (process group on RCCL, barrier, MAX all-reduce) on one GPU"""
    import json
    import subprocess
    env = dict(os.environ, RESCO_BENCH_FORCE_DIST='1', MASTER_ADDR='127.0.0.1', MASTER_PORT='29533', RANK='0', WORLD_SIZE='1',
               LOCAL_RANK='0')
    out = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--steps', '6', '--warmup', '2', '--envs', '256',
                          '--no-cpu-baseline'], env=env, capture_output=True, text=True, timeout=170)
    assert out.returncode == 0, out.stderr[-2000:]
    lines = [l for l in out.stdout.strip().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1 and out.stdout.strip().splitlines()[-1] == lines[0]          # one JSON line, and it is the last one
    line = json.loads(lines[0])
    assert line['n_gpus'] == 1 and line['steps'] == 6 and line['value'] > 0 and line['roofline']['frac'] > 0
    assert line['config']['pipes'] == 2 and line['roofline']['concurrent_launches'] == 2 and line['roofline']['env_steps_per_launch'] == 128
    assert line['all_outputs']['value'] > 0
    assert line['config']['episode_window'][1] - line['config']['episode_window'][0] == 6


def test_two_ranks_through_bench_on_one_gpu():
    """The N > 1 launch of the contract (`python -m torch.distributed.run --nproc-per-node 2 ... bench.py --gpus 2`) end to end
    through the real HIP path where no multi-GPU node exists: two ranks, both on device 0 (RESCO_BENCH_DEVICE), gloo rendezvous
    (RESCO_BENCH_BACKEND), each with two pipes, the NUMA binding executed.  The union of the two shards (global environment index
    = rank * envs + local index) must equal ONE process stepping the 2N batch: the state digests over all environments agree."""
    import json
    import subprocess
    port = 29600 + (os.getpid() % 300)
    common = ['--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--digest']
    env = dict(os.environ, RESCO_BENCH_DEVICE='0', RESCO_BENCH_BACKEND='gloo')
    two = subprocess.run([sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', '2', '--master-addr', '127.0.0.1',
                          '--master-port', str(port), os.path.join(ROOT, 'bench.py'), '--gpus', '2', '--envs', '128'] + common,
                         env=env, capture_output=True, text=True, timeout=280)
    assert two.returncode == 0, two.stderr[-2000:]
    lines = [l for l in two.stdout.strip().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1                                         # rank 0 prints the one line
    a = json.loads(lines[0])
    assert a['n_gpus'] == 2 and a['config']['envs_per_gpu'] == 128 and a['config']['pipes'] == 2 and a['value'] > 0
    assert a['scaling'] == 'weak' and 'x2' in a['config']['parallelism']
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--envs', '256', '--pipes', '1'] + common,
                         env=dict(os.environ), capture_output=True, text=True, timeout=170)
    assert one.returncode == 0, one.stderr[-2000:]
    b = json.loads([l for l in one.stdout.strip().splitlines() if l.startswith('{"metric"')][0])
    assert a['state_digest'] == b['state_digest']


@pytest.mark.gpu
def test_plain_bench_gpus_2_starts_two_ranks():
    """`python bench.py --gpus 2` without a launcher becomes the contract's two-rank launch by itself (round-5 review: --gpus was never
    read); both ranks on device 0 here (RESCO_BENCH_DEVICE), gloo rendezvous.  n_gpus and the process group's own world size say 2,
    and the digest equals the launcher-started run's.  A launcher whose world differs from --gpus is refused (CPU test:
    tests/test_distributed_cpu.py::test_bench_gpus_flag_means_that_many_ranks)."""
    import json
    import subprocess
    common = ['--gpus', '2', '--envs', '128', '--steps', '6', '--warmup', '2', '--no-cpu-baseline', '--digest']
    env = dict(os.environ, RESCO_BENCH_DEVICE='0', RESCO_BENCH_BACKEND='gloo')
    env.pop('WORLD_SIZE', None)
    r = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + common, env=env, capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    lines = [l for l in r.stdout.strip().splitlines() if l.startswith('{"metric"')]
    assert len(lines) == 1
    a = json.loads(lines[0])
    assert a['n_gpus'] == 2 and a['rccl_ranks'] == 2 and a['config']['envs_per_gpu'] == 128 and a['value'] > 0
    one = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py'), '--envs', '256', '--pipes', '1', '--steps', '6', '--warmup', '2',
                          '--no-cpu-baseline', '--digest'], env=dict(os.environ), capture_output=True, text=True, timeout=170)
    assert one.returncode == 0, one.stderr[-2000:]
    b = json.loads([l for l in one.stdout.strip().splitlines() if l.startswith('{"metric"')][0])
    assert b['n_gpus'] == 1 and b['rccl_ranks'] is None and a['state_digest'] == b['state_digest']
    # without enough devices and without the one-device override the launch is refused, not silently folded onto one GPU
    import torch
    if torch.cuda.device_count() < 2:
        env2 = dict(os.environ, RESCO_BENCH_BACKEND='gloo')
        env2.pop('WORLD_SIZE', None)
        env2.pop('RESCO_BENCH_DEVICE', None)
        r2 = subprocess.run([sys.executable, os.path.join(ROOT, 'bench.py')] + common, env=env2, capture_output=True, text=True, timeout=280)
        assert r2.returncode != 0 and '{"metric"' not in r2.stdout


def test_device_invariant_counter():
    """The short paths of the plan and the move rest on two invariants of the classification (a vehicle without FL_H never needs the
    walk over the links, one without FL_MH never leaves its lane), i.e. on classify() and the plan computing the same floating-point
    expression to the same bits at different inline sites (-ffp-contract=off).  The host emulation asserts them; here the CHECKING
    build of the very kernel (resco_amd/build.py: -DRS_DEVICE_ASSERT) counts violations ON THE DEVICE (rs_stats()[11]) over whole
    episodes of two maps -- congested (random policy) and free-flowing (MAXPRESSURE) -- and the count must be zero; the same run
    must still equal the oracle (the checks change nothing)."""
    import subprocess
    from resco_amd.build import CHECK_LIB, build_check_library
    if not os.path.exists(CHECK_LIB):
        build_check_library()
    code = r"""
import sys, numpy as np
sys.path.insert(0, %r); sys.path.insert(0, %r)
from conftest import load_scenario
from resco_amd.sim import BatchedSim
from oracle.pyoracle import OracleEnv
tot = 0
for name, pol in (('cologne8', 'random'), ('ingolstadt7', 'maxpressure'), ('cologne3', 'random')):
    sc = load_scenario(name)
    sim = BatchedSim(sc, 64, seed=3)
    o = OracleEnv(sc, env_index=5, seed=3, sigma=-1.0, speed_dev=1); o.observe()
    for k in range(360):
        if pol == 'random': sim.act_random(k)
        else: sim.act_maxwave(1)
        if k < 120:
            sim.sync(); o.step(sim.read('actions')[5])
        sim.step(None)
        if k == 119:
            v = o.vehicles(); live = v['lane'] != 0xFFFF
            assert np.array_equal(sim.read('veh_lane')[5], v['lane']) and np.array_equal(sim.read('veh_pos')[5][live], v['pos'][live])
    st = sim.stats()
    assert st['ticks'].min() == 3600 and st['arrived'].min() > 0
    print(name, pol, 'invariant violations', int(st['invariant'].sum()), 'vehicle-ticks', int(st['active_ticks'].sum()))
    tot += int(st['invariant'].sum())
    sim.close()
print('TOTAL', tot)
""" % (ROOT, os.path.join(ROOT, 'tests'))
    r = subprocess.run([sys.executable, '-c', code], env=dict(os.environ, RESCO_SIM_LIB=CHECK_LIB), capture_output=True, text=True, timeout=280)
    assert r.returncode == 0, r.stderr[-2000:]
    print(r.stdout)
    assert r.stdout.strip().splitlines()[-1] == 'TOTAL 0'
    # ... and the production build reports the slot as zero (the checks are compiled out)
    from resco_amd.sim import BatchedSim
    sim = BatchedSim(load_scenario('cologne1'), 4, seed=1)
    for k in range(30):
        sim.act_random(k)
        sim.step(None)
    assert (sim.stats()['invariant'] == 0).all() and 'cap_blocked' in sim.stats()
    sim.close()


def test_arrival_departure_counters_and_mplight_full_batched():
    """the batched outputs behind fma2c / mplight_full (RS_BUF_ARRIVALS / DEPARTURES / MPLIGHT_FULL) for N environments:
    equal to the oracle's for sampled environments, and consistent with the Signal views of environment 0"""
    from oracle.pyoracle import OracleEnv
    from resco_amd.sim import BatchedSim
    sc = load_scenario('cologne8')
    N, seed = 96, 13
    sim = BatchedSim(sc, N, seed=seed)
    orcs = {e: OracleEnv(sc, env_index=e, seed=seed, sigma=-1.0, speed_dev=1) for e in (0, 95)}
    for o in orcs.values():
        o.observe()
    for k in range(90):
        sim.act_random(k)
        sim.sync()
        a = sim.read('actions')
        sim.step(None)
        for e, o in orcs.items():
            o.step(a[e])
        if k % 15 == 14:
            arr, dep, mf = sim.read('arrivals'), sim.read('departures'), sim.read('mplight_full')
            la = sim.read('lane_arrivals')
            for e, o in orcs.items():
                ref = o.outputs()
                np.testing.assert_array_equal(la[e], ref['lane_arrivals'])
                np.testing.assert_array_equal(arr[e], ref['arrivals'])
                np.testing.assert_array_equal(dep[e], ref['departures'])
                np.testing.assert_array_equal(mf[e], ref['mplight_full'])
    assert sim.read('arrivals').sum() > 0 and sim.read('departures').sum() > 0
    sim.close()


def test_batched_drq_state_equals_the_signal_views():
    """VecMultiSignal's derived `drq` state (all environments, torch ops over the device buffers) equals states.drq computed
    through the Signal views of a single-environment MultiSignal run with the same seed and actions (reference
    states.py:9-28)"""
    import torch
    from resco_amd import rewards, states
    from resco_amd.multi_signal import MultiSignal, VecMultiSignal
    # (MultiSignal's first episode runs with seed + 0x9E3779B1, the reference restarts SUMO with --random per episode)
    vec = VecMultiSignal('cologne3', 4, states=('drq', 'drq_norm'), rewards=('wait',), seed=(6 + 0x9E3779B1) & 0xFFFFFFFF)
    env = MultiSignal('t', 'cologne3', None, states.drq, rewards.wait, yellow_length=3, end_time=28800,
                      log_dir=tempfile.mkdtemp() + os.sep, seed=6)
    vec.reset()
    env.reset()
    sc = vec.scenario
    rng = np.random.default_rng(4)
    for k in range(25):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        obs, _, _, _ = vec.step(torch.as_tensor(np.repeat(a[None, :], 4, axis=0), device='cuda'))
        ref, _, _, _ = env.step({sid: int(a[i]) for i, sid in enumerate(env.all_ts_ids)})
    vec.sync()
    got = obs['drq'][0].cpu().numpy()
    o = 0
    for sid in env.all_ts_ids:
        rows = np.asarray(ref[sid])[0]
        np.testing.assert_allclose(got[o:o + len(rows)], rows, rtol=0, atol=1e-4)       # (speed sums: 16.16 fixed point on the device)
        o += len(rows)
    assert o == got.shape[0] and got[:, 1:].sum() > 0
    vec.close()
    env.close()


@pytest.mark.parametrize('map_name,full', [('cologne3', False), ('ingolstadt7', False), ('cologne8', True), ('ingolstadt21', True)])
def test_batched_fma2c_equals_the_signal_views(map_name, full):
    """VecMultiSignal's FMA2C observations and rewards for all environments (gathers / one matrix product over the device
    buffers, incl. RS_BUF_LANE_ARRIVALS) equal states.fma2c / rewards.fma2c evaluated through the Signal views of a
    single-environment MultiSignal with the same seed and actions (reference states.py:162-229, rewards.py:72-136)"""
    import torch
    from resco_amd import rewards, states
    from resco_amd.config.map_config import map_configs
    from resco_amd.config.mdp_config import activate
    from resco_amd.multi_signal import MultiSignal, VecMultiSignal
    name = 'fma2c_full' if full else 'fma2c'
    activate('FMA2CFull' if full else 'FMA2C', map_name)
    mc = map_configs[map_name]
    vec = VecMultiSignal(map_name, 3, states=(name,), rewards=(name,), seed=(8 + 0x9E3779B1) & 0xFFFFFFFF)
    env = MultiSignal('t', map_name, None, getattr(states, name), getattr(rewards, name), yellow_length=3, end_time=mc['end_time'],
                      lights=mc['lights'], log_dir=tempfile.mkdtemp() + os.sep, seed=8)
    vec.reset()
    env.reset()
    sc = vec.scenario
    rng = np.random.default_rng(3)
    seen_arrivals = 0.0
    for k in range(40):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        obs, rew, _, _ = vec.step(torch.as_tensor(np.repeat(a[None, :], 3, axis=0), device='cuda'))
        ro, rr, _, _ = env.step({sid: int(a[i]) for i, sid in enumerate(env.all_ts_ids)})
        if k % 5 == 4:
            vec.sync()
            assert list(obs[name].keys()) == list(ro.keys()) and list(rew[name].keys()) == list(rr.keys())
            for key in ro:
                np.testing.assert_allclose(obs[name][key][0].cpu().numpy(), np.asarray(ro[key], np.float64), rtol=0, atol=1e-5, err_msg=key)
                np.testing.assert_allclose(float(rew[name][key][0]), float(rr[key]), rtol=0, atol=1e-3, err_msg=key)
            seen_arrivals += float(vec.tensor('lane_arrivals').sum())
    assert seen_arrivals > 0
    vec.close()
    env.close()
