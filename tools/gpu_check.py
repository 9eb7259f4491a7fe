#!/usr/bin/env python3
"""Quick GPU sanity run (development aid): HIP library vs CPU oracle on a few maps, then a short timing of the bench
workload for one or more block configurations.  python tools/gpu_check.py [--maps a,b] [--steps n] [--blocks 0,-512]"""
import argparse, json, os, sys, time
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario
from resco_amd.sim import BatchedSim


def parity(name, steps, n=4, fixed=0, block=0):
    from oracle.pyoracle import OracleEnv        # checker only (development aid, not a product path)
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    sim = BatchedSim(sc, n, seed=1, sigma=-1.0, speed_dev=1, fixed_program=fixed, block_threads=block)
    orcs = [OracleEnv(sc, env_index=e, seed=1, sigma=-1.0, speed_dev=1, fixed_program=fixed) for e in range(n)]
    for o in orcs: o.observe()
    rng = np.random.default_rng(0)
    for k in range(steps):
        acts = np.stack([rng.integers(0, g, n) for g in sc.tls_ngreen], 1).astype(np.int32)
        sim.step(acts)
        for e, o in enumerate(orcs): o.step(acts[e])
        if k % 10 != 9 and k != steps - 1: continue
        out = sim.outputs(); lanes = sim.read('veh_lane'); poss = sim.read('veh_pos')
        for e, o in enumerate(orcs):
            ref = o.outputs(); v = o.vehicles()
            if not np.array_equal(lanes[e], v['lane']): return '%s: lanes differ at step %d env %d' % (name, k, e)
            live = v['lane'] != 0xFFFF
            if not np.array_equal(poss[e][live], v['pos'][live]): return '%s: positions differ at step %d env %d' % (name, k, e)
            for key in ('lane_agg', 'drq_norm', 'phase', 'mplight', 'wave', 'pressure', 'wait', 'wait_norm'):
                if not np.array_equal(out[key][e], ref[key]): return '%s: %s differs at step %d env %d' % (name, key, k, e)
    st = sim.stats(); so = orcs[0].stats()
    for key in so:
        if st[key][0] != so[key]: return '%s: stat %s %d != %d' % (name, key, st[key][0], so[key])
    sim.close()
    return '%s: %d steps x %d envs bit-identical to the oracle (fixed=%d block=%d)' % (name, steps, n, fixed, block)


def timing(name, n, block, steps=120, warm=60):
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    if os.environ.get('RS_CAPACITY'):
        sc.capacity = int(os.environ['RS_CAPACITY'])
    sim = BatchedSim(sc, n, seed=0, sigma=-1.0, speed_dev=1, block_threads=block)
    for k in range(warm):
        sim.act_random(k); sim.step(None)
    sim.sync(); sim.timing(True)
    t0 = time.perf_counter()
    for k in range(warm, warm + steps):
        sim.act_random(k); sim.step(None)
    sim.sync()
    dt = time.perf_counter() - t0
    ms, launches = sim.timing_read()
    st = sim.stats()
    info = sim.info()
    sim.close()
    return dict(map=name, envs=n, block=info['block_threads'], v128=block < 0, lds=info['lds_bytes'], env_steps_per_s=n * steps / dt,
                kernel_ms=ms / max(1, launches), mean_active=float((st['active_ticks'] / st['ticks']).mean()))


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('--maps', default='cologne1,cologne8,ingolstadt21')
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--blocks', default='0')
    ap.add_argument('--envs', type=int, default=4096)
    ap.add_argument('--no-parity', action='store_true')
    ap.add_argument('--time-map', default='ingolstadt21')
    a = ap.parse_args()
    if not a.no_parity:
        for m in a.maps.split(','):
            print(parity(m, a.steps), flush=True)
        print(parity('ingolstadt7', 40, fixed=1), flush=True)
        print(parity('ingolstadt21', 30, block=-512), flush=True)
    for b in a.blocks.split(','):
        print(json.dumps(timing(a.time_map, a.envs, int(b))), flush=True)
