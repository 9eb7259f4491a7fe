#!/usr/bin/env python3
"""Who carries the delay?  Per-trip delay (timeLoss + departDelay, what utils/readXML.py:41-58 sums per tripinfo entry) of one
oracle episode, split by the approach a trip's route uses (study tool, TEST INFRASTRUCTURE).

  python oracle/study/approach_split.py ingolstadt21 FIXED --groups "E:-201201945#0.78>-174800513" "S:23166741#5>*" ...

A group is NAME:FROM_EDGE>TO_EDGE ('*' = any); a trip belongs to the first group its route matches; 'victims' are the other
trips whose route shares an edge with a route of group 1 upstream of that group's approach (they stand in its queue); 'rest'.
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.delay_eval import run_env

ap = argparse.ArgumentParser()
ap.add_argument('map'); ap.add_argument('policy')
ap.add_argument('--groups', nargs='*', default=[])
ap.add_argument('--env', type=int, default=0)
ap.add_argument('--steps', type=int, default=360)
args = ap.parse_args()
os.environ['ORC_TRIP_LOG'] = '1'
env, sc = run_env(args.map, args.policy, args.env, 0, args.steps, trip_log=1)
A = sc.arrays
eid = {e: i for i, e in enumerate(sc.edge_ids)}
routes = [A['route_edge'][A['route_start'][r]:A['route_start'][r + 1]].tolist() for r in range(sc.n_routes)]
groups = []
for g in args.groups:
    name, spec = g.split(':', 1); a, b = spec.split('>')
    groups.append((name, eid[a], None if b == '*' else eid[b]))
def match(r, a, b):
    for i in range(len(r) - 1):
        if r[i] == a and (b is None or r[i + 1] == b):
            return i
    return -1
rgroup = np.full(sc.n_routes, -1)
for ri, r in enumerate(routes):
    for gi, (_, a, b) in enumerate(groups):
        if match(r, a, b) >= 0:
            rgroup[ri] = gi; break
# victims of group 0: share an upstream edge with one of its routes
up = set()
if groups:
    for ri, r in enumerate(routes):
        if rgroup[ri] == 0:
            i = match(r, groups[0][1], groups[0][2]); up.update(r[:i + 1])
nG = len(groups)
for ri, r in enumerate(routes):
    if rgroup[ri] < 0:
        rgroup[ri] = nG if (set(r) & up) else nG + 1
names = [g[0] for g in groups] + ['victims(of %s)' % groups[0][0] if groups else 'victims', 'rest']
log = env.trip_log(); v = env.vehicles(); now = env.time
act = v['lane'] < 0xFFFE
delay = np.full(sc.n_trips, np.nan); state = np.zeros(sc.n_trips, int)      # 0 never departed, 1 running, 2 arrived
for k in range(sc.n_trips):
    if log[k, 1] > 0:
        delay[k] = log[k, 2] / 1024.0 + (log[k, 0] - 1 - A['trip_depart'][k]); state[k] = 2
for s in np.nonzero(act)[0]:
    k = v['trip'][s]
    delay[k] = v['time_loss'][s] + (int(v['depart'][s]) - 1 - A['trip_depart'][k]); state[k] = 1
tg = rgroup[A['trip_route']]
tot = np.nansum(delay); n_info = int((state > 0).sum())
print('%s %s: %d tripinfo entries (%d arrived, %d running), %d not departed; mean delay %.1f s' % (args.map, args.policy, n_info, (state == 2).sum(), (state == 1).sum(), (state == 0).sum(), tot / n_info))
print('%-28s %6s %6s %6s %6s %9s %7s %7s' % ('group', 'trips', 'arr', 'run', 'nodep', 'mean dly', 'share', 'of-avg'))
for gi, nm in enumerate(names):
    m = tg == gi
    d = delay[m]
    print('%-28s %6d %6d %6d %6d %9.1f %6.1f%% %7.1f' % (nm, m.sum(), (state[m] == 2).sum(), (state[m] == 1).sum(), (state[m] == 0).sum(), np.nanmean(d) if np.isfinite(d).any() else 0.0, 100 * np.nansum(d) / tot, np.nansum(d) / n_info))
