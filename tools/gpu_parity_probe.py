#!/usr/bin/env python3
"""GPU-vs-oracle parity probe (development aid; the pytest version lives in tests/test_gpu_parity.py).

Runs the HIP library and the CPU oracle on the same seeded action script and reports the first mismatch
per buffer."""
import os
import sys
import time

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario           # noqa: E402
from resco_amd.sim import BatchedSim              # noqa: E402
from oracle.pyoracle import OracleEnv             # noqa: E402

INT_BUFS = ['phase', 'mplight', 'wave', 'pressure', 'queue_sum', 'queue_max']
FLT_BUFS = ['lane_agg', 'drq_norm', 'wait', 'wait_norm']
VEH = [('veh_lane', 'lane'), ('veh_trip', 'trip'), ('veh_pos', 'pos'), ('veh_speed', 'speed'), ('veh_cursor', 'cursor'),
       ('veh_swait', 'sumo_wait'), ('veh_rwait', 'resco_wait'), ('veh_owner', 'owner'), ('veh_tloss', 'time_loss'),
       ('veh_depart', 'depart'), ('veh_accel', 'accel')]


def run(name, n_envs, steps, sigma, speed_dev, fixed, seed=3):
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    sim = BatchedSim(sc, n_envs, seed=seed, sigma=sigma, speed_dev=speed_dev, fixed_program=fixed)
    print(name, sim.info())
    orcs = [OracleEnv(sc, env_index=e, seed=seed, sigma=sigma, speed_dev=speed_dev, fixed_program=fixed)
            for e in range(n_envs)]
    for o in orcs:
        o.observe()
    rng = np.random.default_rng(seed)
    G = sc.tls_ngreen
    bad = 0
    t_gpu = t_cpu = 0.0
    for step in range(-1, steps):
        if step >= 0:
            acts = np.stack([rng.integers(0, G) for _ in range(n_envs)]).astype(np.int32)
            t0 = time.time(); sim.step(acts); sim.sync(); t_gpu += time.time() - t0
            t0 = time.time()
            for e, o in enumerate(orcs):
                o.step(acts[e])
            t_cpu += time.time() - t0
        out = sim.outputs()
        vg = {g: sim.read(g) for g, _ in VEH}
        env = sim.read('env')
        for e, o in enumerate(orcs):
            ref = o.outputs()
            for b in INT_BUFS + FLT_BUFS:
                if not np.array_equal(out[b][e], ref[b]):
                    idx = np.argwhere(out[b][e] != ref[b])[0]
                    print('MISMATCH step', step, 'env', e, b, 'at', idx, 'gpu', out[b][e][tuple(idx)], 'ref', ref[b][tuple(idx)])
                    bad += 1
            vo = o.vehicles()
            if env[e, 2] != vo['hw'] or env[e, 1] != vo['next_trip']:
                print('MISMATCH step', step, 'env', e, 'hw/next_trip', env[e], vo['hw'], vo['next_trip']); bad += 1
            hw = vo['hw']
            live = vo['lane'][:hw] != 0xFFFF
            for g, r in VEH:
                a, b_ = vg[g][e][:hw], vo[r][:hw]
                if r == 'trip':
                    b_ = b_.astype(np.int64) & 0xFFFF
                    a = a.astype(np.int64)
                if r not in ('lane', 'trip'):
                    a, b_ = a[live], b_[live]
                if r in ('resco_wait', 'owner', 'depart', 'accel'):
                    act = vo['lane'][:hw][live] < 0xFFFE
                    a, b_ = a[act], b_[act]
                if not np.array_equal(a, b_):
                    idx = np.argwhere(a != b_)[0]
                    print('MISMATCH step', step, 'env', e, g, 'slot', idx, 'gpu', a[tuple(idx)], 'ref', b_[tuple(idx)]); bad += 1
        if bad > 20:
            break
    st = sim.stats()
    so = orcs[0].stats()
    for k in st:
        if st[k][0] != so[k]:
            print('STAT MISMATCH', k, st[k][0], so[k]); bad += 1
    print(name, 'n_envs', n_envs, 'steps', steps, 'sigma', sigma, 'fixed', fixed, 'mismatches', bad,
          'gpu s/step %.4f cpu s/step/env %.4f' % (t_gpu / max(1, steps), t_cpu / max(1, steps) / n_envs), st['active'][:4])
    sim.close()
    return bad


if __name__ == '__main__':
    total = 0
    total += run('cologne1', 4, 40, 0.0, 0, 0)
    total += run('cologne1', 4, 60, 0.5, 1, 0)
    total += run('cologne1', 2, 30, 0.5, 1, 1)
    total += run('cologne8', 3, 40, 0.5, 1, 0)
    total += run('ingolstadt21', 2, 60, 0.5, 1, 0)
    print('TOTAL MISMATCHES', total)
    sys.exit(1 if total else 0)
