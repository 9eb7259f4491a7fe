"""CPU: the step KERNEL's source (resco_amd/csrc/resco_step.h), compiled for the host and executed thread by thread
(tests/hostemu), against the oracle -- bit for bit, in ascending, descending and shuffled thread order.

This pins the kernel's logic, and its independence of the order in which the threads of a workgroup run inside a
phase, without a GPU.  The -m gpu tests pin the real thing (the same source compiled by hipcc) against the same oracle."""
import os

import numpy as np
import pytest

from conftest import load_golden, load_scenario, preroll_actions
from oracle.pyoracle import OracleEnv
from hostemu.emu import EmuSim

OUT = ['lane_agg', 'drq_norm', 'phase', 'mplight', 'wave', 'wait', 'wait_norm', 'pressure', 'queue_sum', 'queue_max',
       'arrivals', 'departures', 'mplight_full', 'lane_arrivals']
VEH = [('veh_lane', 'lane'), ('veh_pos', 'pos'), ('veh_speed', 'speed'), ('veh_cursor', 'cursor'), ('veh_swait', 'sumo_wait'),
       ('veh_tloss', 'time_loss'), ('veh_rwait', 'resco_wait'), ('veh_owner', 'owner'), ('veh_depart', 'depart'), ('veh_accel', 'accel')]


def assert_equal(sim, orcs, step):
    out = sim.outputs(OUT)
    vg = {g: sim.read(g) for g, _ in VEH}
    trip = sim.read('veh_trip')
    envb = sim.read('env')
    for e, o in enumerate(orcs):
        ref = o.outputs()
        for b in OUT:
            np.testing.assert_array_equal(out[b][e], ref[b], err_msg='%s env %d step %d' % (b, e, step))
        v = o.vehicles()
        st = o.stats()
        assert envb[e, 0] == o.time and envb[e, 1] == st['inserted'] and envb[e, 2] == v['hw'] and envb[e, 3] == st['active']
        live = v['lane'] != 0xFFFF
        np.testing.assert_array_equal(trip[e][live].astype(np.int64), v['trip'][live])
        for g, r in VEH:
            a, b_ = vg[g][e], v[r]
            if r != 'lane':
                a, b_ = a[live], b_[live]
            np.testing.assert_array_equal(a, b_, err_msg='%s env %d step %d' % (g, e, step))


@pytest.mark.parametrize('name,steps,order,sigma,speed_dev,fixed', [
    ('cologne1', 60, 0, 0.0, 0, 0),          # parity mode (deterministic), threads ascending
    ('cologne1', 90, 2, -1.0, 1, 0),         # bench mode, shuffled thread order in every phase
    ('cologne1', 40, 1, -1.0, 1, 1),         # FIXED programme, descending
    ('cologne8', 60, 2, -1.0, 1, 0),
    ('cologne3', 50, 1, -1.0, 1, 0),
    ('ingolstadt1', 60, 2, -1.0, 1, 1),
    ('ingolstadt7', 120, 2, -1.0, 1, 0),     # lane-change friction, cooperation requests
    ('ingolstadt21', 70, 2, -1.0, 1, 0),
])
def test_kernel_source_equals_oracle(name, steps, order, sigma, speed_dev, fixed):
    sc = load_scenario(name)
    n, base = 2, 17
    sim = EmuSim(sc, n, order=order, seed=3, sigma=sigma, speed_dev=speed_dev, fixed_program=fixed, env_base=base)
    orcs = [OracleEnv(sc, env_index=base + e, seed=3, sigma=sigma, speed_dev=speed_dev, fixed_program=fixed) for e in range(n)]
    for o in orcs:
        o.observe()
    assert_equal(sim, orcs, -1)
    rng = np.random.default_rng(0)
    for step in range(steps):
        acts = np.stack([rng.integers(0, sc.tls_ngreen) for _ in range(n)]).astype(np.int32)
        if step == 5:
            acts[0, 0] = 99                  # out-of-range action: ignored, phase kept
        sim.step(acts)
        for e, o in enumerate(orcs):
            o.step(acts[e])
        if step % 5 == 4 or step == steps - 1:
            assert_equal(sim, orcs, step)
    st = sim.stats()
    for e, o in enumerate(orcs):
        so = o.stats()
        for k in st:
            assert st[k][e] == so[k], (k, e)
    sim.close()


def _random_configs(n, seed):
    """seeded draws over everything a handle can be created with: map, thread order, driver imperfection, speed factors, detector
    range, programme, step_ratio, block shape, RNG seed, first environment, a warm start"""
    rng = np.random.default_rng(seed)
    maps = ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
    out = []
    for i in range(n):
        name = maps[int(rng.integers(0, len(maps)))]
        out.append(dict(name=name, order=int(rng.integers(0, 3)), sigma=float(rng.choice([-1.0, 0.0, 0.3, 0.9])),
                        speed_dev=int(rng.integers(0, 2)), max_distance=float(rng.choice([1.0, 50.0, 200.0, 9999.0])),
                        fixed=int(rng.random() < 0.25), step_ratio=int(rng.choice([1, 1, 2, 3])), seed=int(rng.integers(0, 2 ** 31)),
                        env_base=int(rng.integers(0, 5000)), warm=int(rng.choice([0, 0, 40, 90])),
                        threads=int(rng.choice([0, 64, 128])), steps=int(rng.integers(12, 28)), case=i,
                        tls_expiry=i % 2))       # both answers to what setPhase leaves behind (rs_params.tls_expiry)
    return out


@pytest.mark.parametrize('cfg', _random_configs(14, 2024), ids=lambda c: '%d-%s' % (c['case'], c['name']))
def test_randomised_parameter_sweep_equals_oracle(cfg):
    """kernel source (host emulation) == oracle on every output and every vehicle field for random combinations of the handle's
    parameters, after an optional warm start under the on-device random policy, under random, repeated and out-of-range actions"""
    sc = load_scenario(cfg['name'])
    kw = dict(seed=cfg['seed'], sigma=cfg['sigma'], speed_dev=cfg['speed_dev'], max_distance=cfg['max_distance'],
              fixed_program=cfg['fixed'], step_ratio=cfg['step_ratio'], tls_expiry=cfg['tls_expiry'])
    bt = cfg['threads'] if cfg['threads'] and cfg['threads'] <= sc.capacity else 0
    sim = EmuSim(sc, 1, order=cfg['order'], env_base=cfg['env_base'], block_threads=bt, **kw)
    o = OracleEnv(sc, env_index=cfg['env_base'], **kw)
    o.observe()
    ratio = cfg['step_ratio']
    for k in range(cfg['warm'] // ratio):
        sim.act_random(k)
        sim.step(None)
        sim.sync()
        o.step(sim.read('actions')[0])
    rng = np.random.default_rng(cfg['case'])
    prev = np.zeros(sc.n_signals, np.int32)
    for step in range(cfg['steps']):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        keep = rng.random(sc.n_signals) < 0.3
        a[keep] = prev[keep]                                # the no-change branch of prep_phase
        if step % 9 == 4:
            a[int(rng.integers(0, sc.n_signals))] = int(rng.choice([-1, 97]))      # not a phase index: the signal is left alone
        prev = a.copy()
        sim.step(a[None, :])
        o.step(a)
        if step % 6 == 5 or step == cfg['steps'] - 1:
            assert_equal(sim, [o], step)
    st, so = sim.stats(), o.stats()
    for k in st:
        assert st[k][0] == so[k], k
    sim.close()


@pytest.mark.parametrize('name,steps,order', [('cologne8', 60, 2), ('ingolstadt7', 80, 1)])
def test_work_lists_overflow(name, steps, order):
    """the work lists of the phases (look-ahead, lane change, lane leavers) are scheduling only: with lists of 8 entries they
    overflow in most ticks and every thread handles its own slots in full -- same results"""
    sc = load_scenario(name)
    sim = EmuSim(sc, 1, order=order, short_lists=True, seed=9, env_base=3)
    o = OracleEnv(sc, env_index=3, seed=9, sigma=-1.0, speed_dev=1)
    o.observe()
    rng = np.random.default_rng(2)
    for step in range(steps):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        sim.step(a[None, :])
        o.step(a)
        if step % 10 == 9:
            assert_equal(sim, [o], step)
    sim.close()


def test_half_the_threads_two_slots_each():
    """block_threads < capacity: a thread owns slots tid, tid + B (the 512-thread launch of a 1024-slot scenario)"""
    sc = load_scenario('cologne8')
    sim = EmuSim(sc, 1, order=2, seed=5, block_threads=64)
    o = OracleEnv(sc, env_index=0, seed=5, sigma=-1.0, speed_dev=1)
    o.observe()
    rng = np.random.default_rng(1)
    for step in range(60):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        sim.step(a[None, :])
        o.step(a)
    assert_equal(sim, [o], 59)
    sim.close()


def test_step_ratio_two_against_the_reference_s_own_loop():
    """MultiSignal(step_ratio=2): step_sim() = two simulation steps (multi_signal.py:102-105), i.e. 6 yellow + 14 further ticks per
    env-step with step_length 10 / yellow_length 3, while Signal.observe still adds step_length to a waiting time.  The golden
    case was produced by the REFERENCE's unmodified step loop (tests/golden/make_golden.py); the kernel source (host emulation) and
    the oracle's orc_step must both reproduce it."""
    meta, g = load_golden('cologne8_d200_sr2')
    assert meta['step_ratio'] == 2
    sc = load_scenario('cologne8')
    sim = EmuSim(sc, 1, order=2, seed=meta['seed'], max_distance=meta['max_distance'], step_ratio=2, tls_expiry=meta.get('tls_expiry', 0))
    o = OracleEnv(sc, env_index=0, seed=meta['seed'], max_distance=meta['max_distance'], sigma=-1.0, speed_dev=1, step_ratio=2, tls_expiry=meta.get('tls_expiry', 0))
    o.observe()
    for k in range(meta['steps']):
        sim.step(g['actions'][k][None, :])
        o.step(g['actions'][k])
        assert int(sim.time()[0]) == 20 * (k + 1) == g['time'][k + 1] - sc.begin
        np.testing.assert_array_equal(sim.read('phase')[0], g['phase'][k + 1])
        np.testing.assert_array_equal(sim.read('mplight')[0].reshape(-1), g['mplight'][k + 1])
        np.testing.assert_array_equal(sim.read('lane_agg')[0][:, :4], g['agg'][k + 1][:, :4])
        np.testing.assert_array_equal(sim.read('wait')[0], g['wait'][k + 1].astype(np.float32))
    assert_equal(sim, [o], meta['steps'] - 1)
    sim.close()


def test_warm_start_ticks_and_reinit_signals():
    """rs_ticks (warm-up / step_sim) and rs_reinit_signals (fresh Signal objects on a loaded network) vs the oracle, and
    the loaded-network golden of the reference's own Python at its first observe"""
    meta, g = load_golden('cologne8_d200_warm180')
    sc = load_scenario('cologne8')
    sim = EmuSim(sc, 1, seed=meta['seed'], max_distance=meta['max_distance'], tls_expiry=meta.get('tls_expiry', 0))
    o = OracleEnv(sc, env_index=0, seed=meta['seed'], max_distance=meta['max_distance'], sigma=-1.0, speed_dev=1, tls_expiry=meta.get('tls_expiry', 0))
    o.observe()
    for k in range(meta['preroll']):
        sim.act_random(k)
        sim.step(None)
        o.step(preroll_actions(sc, meta['seed'], 0, k))
    sim.reinit_signals()
    o.reinit_signals()
    o.observe()
    assert_equal(sim, [o], 0)
    np.testing.assert_array_equal(sim.read('mplight')[0].reshape(-1), g['mplight'][0])
    np.testing.assert_array_equal(sim.read('lane_agg')[0][:, :4], g['agg'][0][:, :4])
    for k in range(10):
        sim.step(g['actions'][k][None, :])
        o.step(g['actions'][k])
        np.testing.assert_array_equal(sim.read('wait')[0], g['wait'][k + 1].astype(np.float32))
    sim.ticks(7)
    for _ in range(7):
        o.tick()
    o.observe()
    assert_equal(sim, [o], 11)
    sim.close()


def test_backlog_accounting():
    """the per-departure-lane backlog (RS_BUF_DEP_NEXT) -> BatchedSim.backlog() / trip_delay() vs the oracle's own count"""
    sc = load_scenario('ingolstadt7')
    sim = EmuSim(sc, 2, seed=9, fixed_program=1)
    orcs = [OracleEnv(sc, env_index=e, seed=9, sigma=-1.0, speed_dev=1, fixed_program=1) for e in range(2)]
    for k in range(150):
        sim.step(None)
        for o in orcs:
            o.step(np.zeros(sc.n_signals, np.int32))
    cnt, waited = sim.backlog()
    for e, o in enumerate(orcs):
        w, c = o.backlog_delay()
        assert cnt[e] == c and waited[e] == w
    assert (sim.trip_delay() > 0).all()
    sim.close()


def test_step_sim_leaves_the_signal_objects_alone():
    """MultiSignal.step_sim() is only sumo.simulationStep() (multi_signal.py:102-105): rs_step_sim advances the simulation and
    touches neither the RESCO waiting times nor the arrival / departure bookkeeping nor any output buffer; vehicles that
    leave meanwhile show up in the departures of the NEXT observe.  Against the oracle (ticks without orc_observe)."""
    sc = load_scenario('cologne8')
    sim = EmuSim(sc, 1, order=2, seed=9)
    o = OracleEnv(sc, env_index=0, seed=9, sigma=-1.0, speed_dev=1)
    o.observe()
    rng = np.random.default_rng(3)
    for step in range(40):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        sim.step(a[None, :])
        o.step(a)
    before = sim.outputs(OUT)
    rw = sim.read('veh_rwait').copy()
    for _ in range(3):                      # 3 x 4 simulation seconds, no observe in between
        sim.step_sim(4)
        for _ in range(4):
            o.tick()
    after = sim.outputs(OUT)
    for b in OUT:
        np.testing.assert_array_equal(before[b], after[b], err_msg=b)
    live = sim.read('veh_lane')[0] != 0xFFFF
    np.testing.assert_array_equal(sim.read('veh_rwait')[0][live & (rw[0] > 0)], rw[0][live & (rw[0] > 0)])
    assert sim.read('env')[0, 0] == o.time == 412
    a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
    sim.step(a[None, :])
    o.step(a)
    assert_equal(sim, [o], 41)              # incl. departures: the vehicles that arrived during the 12 quiet seconds
    sim.close()


def test_output_mask_writes_only_what_was_asked_for():
    """rs_set_outputs: buffers outside the mask keep their contents (and cost no traffic), the others and the per-signal
    scalars are written as always"""
    sc = load_scenario('cologne8')
    sim = EmuSim(sc, 1, seed=4)
    o = OracleEnv(sc, env_index=0, seed=4, sigma=-1.0, speed_dev=1)
    o.observe()
    rng = np.random.default_rng(5)
    for step in range(20):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        sim.step(a[None, :])
        o.step(a)
    old = sim.outputs(OUT)
    sim.set_outputs(['drq_norm', 'mplight'])
    for step in range(20):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        sim.step(a[None, :])
        o.step(a)
    new, ref = sim.outputs(OUT), o.outputs()
    for b in ('drq_norm', 'mplight', 'phase', 'wait', 'wait_norm', 'pressure', 'queue_sum', 'queue_max', 'arrivals', 'departures'):
        np.testing.assert_array_equal(new[b][0], ref[b], err_msg=b)
    for b in ('lane_agg', 'wave', 'mplight_full', 'lane_arrivals'):
        np.testing.assert_array_equal(new[b], old[b], err_msg=b)
        assert not np.array_equal(new[b][0], ref[b])
    # the per-vehicle acceleration (RS_BUF_VEH_ACCEL) is a maskable buffer too: not written since the mask was set
    v = o.vehicles()
    live = v['lane'] != 0xFFFF
    assert live.any() and np.array_equal(sim.read('veh_speed')[0][live], v['speed'][live])
    assert not np.array_equal(sim.read('veh_accel')[0][live], v['accel'][live])
    # consumers of a switched-off buffer fail loudly instead of acting on stale rows: the MAXWAVE agent reads `wave`
    # (off), the MAXPRESSURE agent `mplight` (on)
    with pytest.raises(RuntimeError):
        sim.act_maxwave(0)
    sim._maxwave_ready = False
    sim.act_maxwave(1)
    with pytest.raises(RuntimeError):
        sim.require_output('wave')
    sim.require_output('mplight')
    sim.set_outputs(None)
    sim.require_output('wave')
    a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
    sim.step(a[None, :])
    o.step(a)
    assert_equal(sim, [o], 40)
    with pytest.raises(ValueError):
        sim.set_outputs(['wait'])
    sim.close()


def test_capacity_exhausted_trips_wait_in_their_backlog():
    """The slot capacity is a hard limit of the working memory: when the network holds `capacity` vehicles the due trips stay in
    their lane's backlog (stats['pending']) and enter as slots become free, lower lane first -- never dropped, and identically
    in the kernel and the oracle."""
    import copy
    sc = copy.copy(load_scenario('cologne1'))
    sc.capacity = 64                                    # a quarter of what the jammed map needs
    sim = EmuSim(sc, 1, order=2, seed=1)
    o = OracleEnv(sc, env_index=0, seed=1, sigma=-1.0, speed_dev=1)
    o.observe()
    full = 0
    for step in range(150):
        a = np.zeros(sc.n_signals, np.int32)            # one phase for ever: the other approaches jam
        sim.step(a[None, :])
        o.step(a)
        st = sim.stats()
        assert st['active'][0] <= 64
        full += int(st['active'][0] == 64 and st['pending'][0] > 0)
    assert full > 20                                     # the limit was really hit, with trips waiting
    assert_equal(sim, [o], 149)
    assert sim.stats()['pending'][0] == o.stats()['pending'] > 0
    sim.close()


def test_out_of_range_actions_keep_the_current_phase():
    """An action that is not a phase index of the signal leaves it alone (the reference would raise inside TraCI; the batched
    ABI must not fault on a bad entry of an [N, S] array): kernel == oracle, and the signal's phase does not change."""
    sc = load_scenario('cologne8')
    sim = EmuSim(sc, 1, seed=2)
    o = OracleEnv(sc, env_index=0, seed=2, sigma=-1.0, speed_dev=1)
    o.observe()
    rng = np.random.default_rng(7)
    for step in range(30):
        a = rng.integers(0, sc.tls_ngreen).astype(np.int32)
        if step % 3 == 1:
            a[step % sc.n_signals] = -1
        if step % 3 == 2:
            a[(step * 5) % sc.n_signals] = 99
        sim.step(a[None, :])
        o.step(a)
        assert_equal(sim, [o], step)
    sim.close()


REF_ENV = '/root/reference/resco_benchmark/environments'


@pytest.mark.skipif(not os.path.isdir(REF_ENV), reason='needs the reference net / route files (build container only)')
def test_route_kwarg_one_route_file_per_run(tmp_path, monkeypatch):
    """MultiSignal(route=prefix): the demand of episode k comes from `<prefix>_<k>.rou.xml` over the net file, SUMO begins at 0
    (multi_signal.py:33-38, 123-124 -- how the reference runs the two grid maps).  Two route files cut from cologne1's demand (even /
    odd trips, departures shifted to begin 0); the host emulation stands in for the HIP library."""
    import xml.etree.ElementTree as ET
    import resco_amd.multi_signal as ms
    from resco_amd import rewards, states
    root = ET.parse(os.path.join(REF_ENV, 'cologne1', 'cologne1.rou.xml')).getroot()
    vtypes = [el for el in root if el.tag == 'vType']
    trips = [el for el in root if el.tag == 'trip']
    for run, part in ((1, trips[0::2]), (2, trips[1::2])):
        r = ET.Element('routes')
        for el in vtypes:
            r.append(el)
        for el in part:
            t = ET.SubElement(r, 'trip', dict(el.attrib))
            t.set('depart', '%.2f' % (float(el.get('depart')) - 25200.0))
        ET.ElementTree(r).write(str(tmp_path / ('demand_%d.rou.xml' % run)))
    monkeypatch.setattr(ms, 'BatchedSim', lambda sc, n, **kw: EmuSim(sc, n, **{k: v for k, v in kw.items() if k != 'device'}))
    env = ms.MultiSignal('t', 'cologne1', os.path.join(REF_ENV, 'cologne1', 'cologne1.net.xml'), states.mplight, rewards.wait,
                         route=str(tmp_path / 'demand'), end_time=3600, yellow_length=3, log_dir=str(tmp_path) + os.sep, seed=3,
                         tripinfo=False)
    assert env.scenario.n_trips == len(trips[0::2]) and env.scenario.begin == 0
    n = {}
    for run in (1, 2):
        env.reset()
        assert env.run == run and env.scenario.n_trips == len(trips[run - 1::2])
        for k in range(40):
            obs, rew, done, info = env.step({ts: k % 4 for ts in env.all_ts_ids})
        assert info == {'eps': run} and not done and env.sim_time() == 400.0
        n[run] = int(env.sim.stats()['inserted'][0])
        # the demand of this run, and no other: every trip scheduled in the first 400 s of ITS file is on the network or waiting
        due = int((env.scenario.trip_depart < 400).sum())
        assert 0 < n[run] <= due and n[run] + int(env.sim.backlog()[0][0]) == due
        assert list(env.scenario.trip_ids) == [el.get('id') for el in trips[run - 1::2]]
    with pytest.raises(EnvironmentError):
        env.reset()                                     # there is no demand_3.rou.xml
    env.close()


def test_group_step_equals_the_calls_it_replaces():
    """rs_group_step (one call through the ABI for all pipes of a GPU) == per pipe [rs_act_*, rs_step], for the random policy and
    MAXWAVE / MAXPRESSURE, several steps per call; the union of the pipes is the single batch (env_base keys the RNG)"""
    from resco_amd.sim import SimGroup
    sc = load_scenario('cologne8')
    for agent in ('random', 'maxwave', 'maxpressure'):
        whole = EmuSim(sc, 4, seed=3)
        pipes = [EmuSim(sc, 2, seed=3, env_base=0), EmuSim(sc, 2, seed=3, env_base=2)]
        grp = SimGroup(pipes)
        for k in range(0, 24, 4):
            for j in range(4):
                if agent == 'random':
                    whole.act_random(k + j)
                else:
                    whole.act_maxwave(1 if agent == 'maxpressure' else 0)
                whole.step(None)
            grp.step(agent, step_key=k, n_steps=4)
        for name in ('veh_pos', 'veh_lane', 'phase', 'wait', 'mplight', 'actions'):
            np.testing.assert_array_equal(whole.read(name), np.concatenate([p.read(name) for p in pipes]))
    with pytest.raises(RuntimeError):
        grp.step('idqn')                        # the policy network is device code: the emulation refuses


def test_mail_flags_survive_every_launch_boundary():
    """Round 6: a plan reads its cooperation mailboxes only when its record is flagged, and between launches the flags travel in
    RS_BUF_VEH_MAIL.  The same congested episode (ingolstadt21 under the net's own programme: blocked lane changers every tick) cut
    into launches of ONE tick each must end in the state of ten-tick launches -- every cooperation request written in the last tick
    of a launch is honoured by the first plan of the next one -- and both equal the oracle, which knows no flags."""
    sc = load_scenario('ingolstadt21')
    kw = dict(seed=4, fixed_program=1)
    a = EmuSim(sc, 1, order=2, **kw)
    b = EmuSim(sc, 1, order=1, **kw)
    o = OracleEnv(sc, env_index=0, seed=4, sigma=-1.0, speed_dev=1, fixed_program=1)
    o.observe()
    mails = 0
    for step in range(45):
        a.step(None)
        for _ in range(10):
            b.step_sim(1)
            mails += int(np.unpackbits(b.read('veh_mail').view(np.uint8)).sum())
        for _ in range(10):
            o.tick()
    assert mails > 50, mails                 # the boundary was crossed with requests pending, many times
    v = o.vehicles()
    live = v['lane'] != 0xFFFF
    np.testing.assert_array_equal(a.read('veh_lane'), b.read('veh_lane'))
    for name in ('veh_pos', 'veh_speed', 'veh_cursor', 'veh_swait', 'veh_tloss', 'veh_coop', 'veh_cooplead', 'veh_coop_odd', 'veh_cooplead_odd'):
        np.testing.assert_array_equal(a.read(name)[0][live], b.read(name)[0][live], err_msg=name)        # (a free slot keeps what its last tenant left)
    for name in ('veh_mail', 'env'):
        np.testing.assert_array_equal(a.read(name), b.read(name), err_msg=name)
    np.testing.assert_array_equal(a.read('veh_lane')[0], v['lane'])
    np.testing.assert_array_equal(a.read('veh_pos')[0][live], v['pos'][live])
    a.close(); b.close()


def test_working_memory_of_the_headline_map_fits_a_cu_four_times():
    """Round 6: four 512-thread workgroups per CU on ingolstadt21 (+12.5 % env-steps/s) need <= 40 960 B of LDS per environment (32
    granules of 1280 B).  The layout is computed by host code the emulation shares with rs_create (lds_carve): a field added to the
    working memory that pushes the headline map over the limit costs a workgroup per CU silently -- this makes it loud."""
    sc = load_scenario('ingolstadt21')
    sim = EmuSim(sc, 1, seed=0)
    info = sim.info()
    sim.close()
    assert sc.capacity == 896
    assert info['lds_bytes'] <= 40960, info
