#!/usr/bin/env python3
"""Model-fidelity study on the CPU oracle (TEST INFRASTRUCTURE): the four figures the reference publishes per episode,
for the static controllers, against the reference-held result arrays (tests/golden/ref_bands.json, produced by
tests/golden/make_ref_bands.py from resco_benchmark/utils/avg_{timeLoss,duration,waitingTime,queue}.py).

  delay     timeLoss + departDelay per tripinfo entry; never-departed demand charged only for
            <vehicle> route files, as the reference's script does                          (utils/readXML.py:16-77)
  duration  tripinfo `duration`, arrived and (--tripinfo-output.write-unfinished) running   (utils/readXML.py:41-44)
  waiting   tripinfo `waitingTime`                                                          (the same loop)
  queue     mean over the steps of sum_signals(queue) / (S + 1)                             (utils/readCSV.py:32-46)

Which array pins which controller:
  FIXED / MAXWAVE / MAXPRESSURE   avg_timeLoss.py rows of the same name (median over the published episodes)
  STOCHASTIC (max_distance 200)   the FIRST episode of the IDQN rows of all four arrays: epsilon decays linearly from 1 over
                                  80 episodes (agents/pfrl_dqn.py:65-70, main.py:91-92), so episode 1 acts uniformly at
                                  random with probability >= 0.9875 -- a random policy observed through IDQN's 200 m detectors
  free-flow residual              duration - delay of the LAST episodes of IDQN (trained; departDelay ~ small): the
                                  controller-independent travel time of the routes, a pin on routing, lengths and speeds

  python oracle/fidelity_eval.py [map ...] [--policies FIXED,MAXWAVE,MAXPRESSURE,STOCHASTIC] [--envs 8]
"""
import argparse
import json
import multiprocessing as mp
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)

MAPS = ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
MAX_DISTANCE = {'FIXED': 200, 'MAXWAVE': 50, 'MAXPRESSURE': 200, 'STOCHASTIC': 200}


def ref_bands():
    with open(os.path.join(ROOT, 'tests', 'golden', 'ref_bands.json')) as f:
        return json.load(f)


def episode(job):
    name, policy, env_index, seed, steps = job[:5]
    sigma = job[5] if len(job) > 5 else -1.0
    from oracle.pyoracle import OracleEnv, lib
    from resco_amd.scenario import Scenario
    from resco_amd.sim import maxwave_tables
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    if policy.endswith('*'):            # ingolstadt21 with the one valid_acts entry repaired (exception 3)
        sc.valid_acts = dict(sc.valid_acts)
        sc.valid_acts['243641585'] = {4: 0, 7: 1, 2: 2}
        policy = policy[:-1]
    env = OracleEnv(sc, env_index=env_index, seed=seed, sigma=sigma, speed_dev=1, max_distance=MAX_DISTANCE[policy],
                    fixed_program=1 if policy == 'FIXED' else 0, trip_log=1)
    env.observe()
    S = sc.n_signals
    G = [int(g) for g in sc.tls_ngreen]
    pairs, valid, order = maxwave_tables(sc)
    L = lib()
    qsum = 0.0
    for k in range(steps):
        a = np.zeros(S, np.int32)
        if policy in ('MAXWAVE', 'MAXPRESSURE'):
            out = env.outputs()
            obs = out['wave'] if policy == 'MAXWAVE' else out['mplight'][:, 1:]
            for s in range(S):
                best, have = 0, False
                for j in range(len(pairs)):
                    p = order[s, j]
                    if p < 0:
                        break
                    act = valid[s, p]
                    if act < 0:
                        continue
                    press = obs[s, pairs[p, 0]] + obs[s, pairs[p, 1]]
                    if not have or press > best:
                        have, best, a[s] = True, press, act
        elif policy == 'STOCHASTIC':
            for s in range(S):
                a[s] = L.orc_hash((seed ^ 0xA5A5A5A5) & 0xFFFFFFFF, env_index, s, k, 7) % G[s]
        env.step(a)
        qsum += float(env.outputs()['queue_sum'].sum()) / (S + 1)
    st = env.stats()
    v = env.vehicles()
    act = v['lane'] < 0xFFFE
    now = env.time
    waited, n_wait = env.backlog_delay()
    trips = st['inserted'] + n_wait
    n_info = st['arrived'] + int(act.sum())                # tripinfo children: arrived + still running
    loss = st['sum_time_loss_q10'] / 1024.0 + float((v['time_loss'] * act).sum())
    # utils/readXML.py:59-68 adds the demand that never departed only for rou.xml files made of <vehicle> elements
    if sc.demand_tag == 'vehicle':
        delay = (loss + st['sum_depart_delay'] + waited) / max(1, trips)
    else:
        delay = (loss + st['sum_depart_delay']) / max(1, n_info)
    duration = (st['sum_duration'] + float(((now - v['depart'].astype(np.int64)) * act).sum())) / max(1, n_info)
    waiting = st['sum_waiting'] / max(1, n_info)
    time_loss = (st['sum_time_loss_q10'] / 1024.0 + float((v['time_loss'] * act).sum())) / max(1, n_info)
    return dict(delay=delay, duration=duration, waiting=waiting, queue=qsum / steps, time_loss=time_loss, arrived=st['arrived'],
                inserted=st['inserted'], pending=st['pending'], mean_active=st['active_ticks'] / max(1, st['ticks']),
                depart_delay=(st['sum_depart_delay'] + waited) / max(1, trips))


def run(name, policy, envs=8, seed=0, steps=360, pool=None, sigma=-1.0):
    jobs = [(name, policy, e, seed, steps, sigma) for e in range(envs)]
    rows = pool.map(episode, jobs) if pool is not None else [episode(j) for j in jobs]
    out = dict(map=name, policy=policy, envs=envs)
    for k in rows[0]:
        x = np.array([r[k] for r in rows], float)
        out[k] = float(np.median(x))
        out[k + '_mean'] = float(x.mean())
    out['delays_sorted'] = [round(float(x), 1) for x in np.sort([r['delay'] for r in rows])]
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('maps', nargs='*', default=MAPS)
    ap.add_argument('--policies', default='FIXED,MAXWAVE,MAXPRESSURE,STOCHASTIC')
    ap.add_argument('--envs', type=int, default=8)
    ap.add_argument('--steps', type=int, default=360)
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--json', default=None)
    ap.add_argument('--compact', action='store_true', help='one line per map: delay ratios only')
    ap.add_argument('--sigma', type=float, default=-1.0)
    args = ap.parse_args()
    from oracle.pyoracle import build
    build()
    RB = ref_bands()
    allrows = []
    with mp.get_context('fork').Pool(min(args.envs, os.cpu_count() or 1)) as pool:
        ratios = []
        for m in args.maps:
            line = []
            for pol in args.policies.split(','):
                r = run(m, pol, args.envs, args.seed, args.steps, pool, args.sigma)
                if args.compact:
                    ref0 = RB.get(m, {}).get(pol.rstrip('*'), {})
                    q = r['delay'] / ref0['delay'] if 'delay' in ref0 else float('nan')
                    line.append('%s %6.1f (%.2f)' % (pol[:2] + pol[-1:], r['delay'], q)); ratios.append(q)
                    allrows.append(r)
                    continue
                ref = RB.get(m, {}).get(pol.rstrip('*'), {})
                cells = []
                for key in ('delay', 'duration', 'waiting', 'queue'):
                    s = '%s %7.1f' % (key, r[key])
                    if key in ref:
                        s += ' /%7.1f (%.2f)' % (ref[key], r[key] / ref[key])
                    cells.append(s)
                resid = r['duration'] - r['time_loss']
                extra = ''
                if pol in ('MAXPRESSURE', 'FIXED') and 'free_flow_residual' in RB.get(m, {}):
                    extra = '  resid %.1f / %.1f' % (resid, RB[m]['free_flow_residual'])
                print('%-13s %-11s %s  | arr %5.0f pend %4.0f V %6.1f dd %5.1f%s' % (m, pol, '  '.join(cells), r['arrived'], r['pending'], r['mean_active'], r['depart_delay'], extra), flush=True)
                allrows.append(r)
            if args.compact:
                print('%-13s %s' % (m, '  '.join(line)), flush=True)
        if args.compact:
            import math
            rr = np.array([x for x in ratios if x == x])
            print('median ratio %.3f   sum|log| %.3f   in +-35%%: %d / %d' % (np.median(rr), np.abs(np.log(rr)).sum(), ((rr > 0.65) & (rr < 1.35)).sum(), len(rr)))
    if args.json:
        with open(args.json, 'w') as f:
            json.dump(allrows, f, indent=1)


if __name__ == '__main__':
    main()
