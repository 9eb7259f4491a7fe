#!/usr/bin/env python3
"""Where is the time lost?  Time-loss seconds (sum of 1 - v / vmax, what SUMO's timeLoss integrates) per lane and per plan()
reason over one oracle episode under any controller (study tool, TEST INFRASTRUCTURE).

  python oracle/study/loss_map.py ingolstadt21 FIXED|MAXWAVE|MAXPRESSURE|STOCHASTIC [--repair] [--keep 0.5] [--top 25]
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleEnv, lib
from resco_amd.scenario import Scenario
from resco_amd.sim import maxwave_tables
from oracle.delay_eval import MAX_DISTANCE

ap = argparse.ArgumentParser()
ap.add_argument('map'); ap.add_argument('policy')
ap.add_argument('--repair', action='store_true', help="ingolstadt21: valid_acts['243641585'] = {4: 0, 7: 1, 2: 2}")
ap.add_argument('--keep', type=float, default=1.0, help='fraction of the trips kept (demand scaling)')
ap.add_argument('--top', type=int, default=25)
ap.add_argument('--env', type=int, default=0)
args = ap.parse_args()
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', args.map + '.npz')); A = sc.arrays
if args.repair:
    sc.valid_acts = dict(sc.valid_acts); sc.valid_acts['243641585'] = {4: 0, 7: 1, 2: 2}
if args.keep < 1.0:
    keep = np.random.RandomState(1).rand(sc.n_trips) < args.keep
    for k in ('trip_depart', 'trip_route', 'trip_vtype'):
        A[k] = np.ascontiguousarray(A[k][keep])
    A['trips_cum'] = np.searchsorted(A['trip_depart'], np.arange(sc.horizon + 2), side='right').astype(np.int32)
pol = args.policy
env = OracleEnv(sc, env_index=args.env, seed=0, sigma=-1.0, speed_dev=1, max_distance=MAX_DISTANCE[pol], fixed_program=1 if pol == 'FIXED' else 0, trip_log=1)
env.observe()
names = ['free', 'leader', 'wronglane', 'red', 'foe', 'nextlead', 'speedlim', 'minorvis', 'coop', 'cooplead']
nl = sc.n_lanes
halt = np.zeros(nl); loss = np.zeros(nl); rl = np.zeros((nl, 10)); lossr = np.zeros(10)
S = sc.n_signals; G = [int(g) for g in sc.tls_ngreen]
pairs, valid, order = maxwave_tables(sc); L = lib()
def act(k):
    a = np.zeros(S, np.int32)
    if pol in ('MAXWAVE', 'MAXPRESSURE'):
        out = env.outputs(); obs = out['wave'] if pol == 'MAXWAVE' else out['mplight'][:, 1:]
        for s in range(S):
            best, have = 0, False
            for j in range(len(pairs)):
                p = order[s, j]
                if p < 0: break
                ac = valid[s, p]
                if ac < 0: continue
                press = obs[s, pairs[p, 0]] + obs[s, pairs[p, 1]]
                if not have or press > best: have, best, a[s] = True, press, ac
    elif pol == 'STOCHASTIC':
        for s in range(S): a[s] = L.orc_hash((0 ^ 0xA5A5A5A5) & 0xFFFFFFFF, args.env, s, k, 7) % G[s]
    return a
Y = sc.yellow_length
for k in range(360):
    a = act(k)
    # the step, tick by tick (prep -> Y ticks -> set -> T - Y ticks -> observe), so that every tick can be sampled
    if pol != 'FIXED':
        for s in range(S):
            cur = env.get_phase(s)
            if cur != a[s] and cur < G[s]:
                y = sc.arrays['tls_yellow'][sc.arrays['tls_yel_off'][s] + cur * G[s] + a[s]]
                if y >= 0: env.set_phase(s, int(y))
    for tick in range(10):
        if tick == Y and pol != 'FIXED':
            for s in range(S): env.set_phase(s, int(a[s]))
        env.tick()
        v = env.vehicles(); r, b_ = env.debug(); hw = v['hw']
        lane = v['lane'][:hw].astype(np.int64); actv = lane < 0xFFFE; sp = v['speed'][:hw]
        la = lane[actv]
        np.add.at(halt, lane[actv & (sp <= 0.1)], 1)
        ls = np.clip(1 - sp[actv] / A['lane_vmax'][la], 0, 1)
        np.add.at(loss, la, ls); np.add.at(lossr, r[:hw][actv], ls); np.add.at(rl, (la, r[:hw][actv]), ls)
    env.observe()
st = env.stats(); v = env.vehicles(); actv = v['lane'] < 0xFFFE
n = st['arrived'] + actv.sum()
print('%s %s trips %d: delay %.1f s (timeLoss %.1f + departDelay %.1f), arrived %d running %d' % (args.map, pol + ('*' if args.repair else ''), len(A['trip_depart']),
      (st['sum_time_loss_q10'] / 1024 + float((v['time_loss'] * actv).sum()) + st['sum_depart_delay']) / n,
      (st['sum_time_loss_q10'] / 1024 + float((v['time_loss'] * actv).sum())) / n, st['sum_depart_delay'] / n, st['arrived'], actv.sum()))
print('time loss by what limited the vehicle (s per trip):', {nm: round(x / n, 1) for nm, x in zip(names, lossr)})
for l in np.argsort(-loss)[:args.top]:
    print('%-30s len %6.1f int=%d loss %7.0f halt %7.0f  %s' % (sc.lane_ids[l], A['lane_len'][l], A['lane_internal'][l], loss[l], halt[l], ' '.join('%s:%d' % (names[i][:4], rl[l, i]) for i in range(10) if rl[l, i] > loss[l] * 0.1)))
