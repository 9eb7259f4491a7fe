#!/usr/bin/env python3
"""Golden case for Signal.generate_config (traffic_signal.py:106-170) -- BUILD CONTAINER ONLY.

The reference falls back to generate_config for a signal that has no entry in signal_configs[map].  All six maps whose demand ships
have entries; the fallback is written for the grid maps, whose net files are in the reference (their demand archives are not).  So:
grid4x4's net, a synthetic demand of this script's own making (600 trips between fringe edges over the first 1500 s), and a map config
WITHOUT any per-signal entry -- every one of the 16 signals goes through generate_config, in the reference (its unmodified MultiSignal /
Signal over oracle/ref_harness.FakeSumo, whose getControlledLinks is served from the net's <connection tl linkIndex> elements) and in
resco_amd.scenario.generate_signal_config.  Stored: the compiled scenario (tests/golden/grid4x4_generated_scenario.npz: tables, no
reference text), the per-signal lanes / lane_sets / downstream the reference derived, and 24 env-steps of its drq / drq_norm / wave
states and wait / wait_norm / pressure rewards.

  python tests/golden/make_generated_config_golden.py
"""
import json
import os
import sys
import tempfile

import numpy as np

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
from oracle import ref_harness                                   # noqa: E402
from oracle.pyoracle import OracleEnv                            # noqa: E402
from resco_amd.scenario import compile_scenario, parse_net       # noqa: E402

TLS_EXPIRY = 0      # the committed fixture was generated under round 5's default (the phase stays); the test passes the recorded value

REF_NET = '/root/reference/resco_benchmark/environments/grid4x4/grid4x4.net.xml'
MAP = 'grid4x4gen'
STEPS, SEED = 24, 11


def synthetic_demand(net, n=600, horizon=1500):
    """trips between fringe edges (ids naming top / bottom / left / right), uniformly at random: (id, vtype, depart, from, to, None)"""
    rng = np.random.default_rng(2024)
    fringe = ('top', 'bottom', 'left', 'right')
    ins = sorted(e for e in net.edges if not net.edges[e].internal and any(e.startswith(f) for f in fringe))
    outs = sorted(e for e in net.edges if not net.edges[e].internal and any(f in e and not e.startswith(f) for f in fringe))
    trips = []
    for i, t in enumerate(np.sort(rng.uniform(0, horizon, n))):
        a = ins[int(rng.integers(0, len(ins)))]
        b = outs[int(rng.integers(0, len(outs)))]
        trips.append(('g%d' % i, 'car', float(np.round(t, 2)), a, b, None))
    return {'car': {'vClass': 'passenger'}}, trips


def main():
    ref_harness.install_stubs(lambda cmd: None)
    ref = ref_harness.import_reference()
    from resco_benchmark.config.signal_config import signal_configs
    cfg = {'phase_pairs': signal_configs['grid4x4']['phase_pairs'], 'valid_acts': None}       # no per-signal entries at all
    signal_configs[MAP] = cfg
    net = parse_net(REF_NET)
    vtypes, trips = synthetic_demand(net)
    sc = compile_scenario(MAP, net, vtypes, trips, 0, 3600, cfg, lights=(), yellow_length=3)
    assert all(sc.signal_meta[s].get('generated') for s in sc.signal_ids)
    sc.save(os.path.join(HERE, 'grid4x4_generated_scenario.npz'))
    state = {'n': 0}

    def factory(cmd):
        orc = OracleEnv(sc, env_index=0, seed=SEED + state['n'], max_distance=200, sigma=-1.0, speed_dev=1, tls_expiry=TLS_EXPIRY)
        state['n'] += 1
        state['orc'] = orc
        return ref_harness.FakeSumo(sc, orc)

    ref_harness._FACTORY['fn'] = factory
    states, rewards = ref['states'], ref['rewards']
    tmp = tempfile.mkdtemp() + os.sep
    env = ref['MultiSignal']('golden', MAP, 'x.net.xml', states.drq_norm, rewards.wait, step_length=10, yellow_length=3, end_time=3600,
                             max_distance=200, lights=(), log_dir=tmp)
    ids = list(env.all_ts_ids)
    rng = np.random.default_rng(5)
    n_green = [len(env.phases[ts]) for ts in ids]
    rec = {k: [] for k in ('drq', 'drq_norm', 'wave', 'wait', 'wait_norm', 'pressure', 'phase', 'agg')}

    def snapshot():
        for fn in ('drq', 'drq_norm', 'wave'):
            out = getattr(states, fn)(env.signals)
            rec[fn].append(np.concatenate([np.asarray(out[ts], dtype=np.float64).reshape(-1) for ts in ids]))
        for fn in ('wait', 'wait_norm', 'pressure'):
            out = getattr(rewards, fn)(env.signals)
            rec[fn].append(np.asarray([float(out[ts]) for ts in ids]))
        rec['phase'].append(np.asarray([env.signals[ts].phase for ts in ids]))
        agg = []
        for ts in ids:
            for lane in env.signals[ts].lanes:
                fo = env.signals[ts].full_observation[lane]
                agg.append([fo['queue'], fo['approach'], fo['total_wait'], fo['max_wait'], sum(v['speed'] for v in fo['vehicles'])])
        rec['agg'].append(np.asarray(agg, dtype=np.float64))

    env.reset()
    snapshot()
    actions = []
    for k in range(STEPS):
        act = [int(rng.integers(0, g)) for g in n_green]
        actions.append(act)
        env.step({ts: a for ts, a in zip(ids, act)})
        snapshot()
    meta = dict(map=MAP, steps=STEPS, seed=SEED + 1, tls_expiry=TLS_EXPIRY, all_ts_ids=ids, n_green=n_green,
                signals={ts: dict(lanes=list(env.signals[ts].lanes), lane_sets=env.signals[ts].lane_sets,
                                  downstream=env.signals[ts].downstream, lane_sets_outbound=env.signals[ts].lane_sets_outbound,
                                  outbound_lanes=list(env.signals[ts].outbound_lanes),
                                  yellow_dict=env.signals[ts].yellow_dict) for ts in ids},
                oracle_stats=state['orc'].stats())
    with open(os.path.join(HERE, 'grid4x4_generated.json'), 'w') as f:
        json.dump(meta, f, indent=0)
    arrays = {k: np.asarray(v) for k, v in rec.items()}
    arrays['actions'] = np.asarray(actions, np.int32)
    np.savez_compressed(os.path.join(HERE, 'grid4x4_generated.npz'), **arrays)
    print('grid4x4 generated config: signals', len(ids), 'lanes', sum(len(env.signals[t].lanes) for t in ids), 'arrived', meta['oracle_stats']['arrived'],
          'max queue', float(arrays['agg'][:, :, 0].max()))


if __name__ == '__main__':
    main()
