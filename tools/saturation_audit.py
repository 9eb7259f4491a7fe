#!/usr/bin/env python3
"""CPU (oracle, test infrastructure): per-approach saturation audit of a map under the net's own FIXED programme -- the round-5 review's
item 7 (ingolstadt21 FIXED: 1.73 x the reference's delay).  For every signal-controlled approach lane, tick by tick:

  demand       vehicles that entered the lane in the hour
  green        seconds the lane's first link showed G / g
  sat s        green seconds that began with a platoon to discharge (>= 3 vehicles on the lane, the first one within 15 m of the stop line)
  crossed      vehicles that left the lane forward in those seconds
  h model      sat s / crossed: the discharge headway the model produces on this approach
  h Krauss     tau + (length + minGap) / v, v = min(speed limit of the lane, of the junction lane behind the stop line) x speedFactor 1: what
               a platoon at speed discharges at (the published car-following model; start-up lost time comes on top, ~2 s per green)
  cap          3600 / h model x green / 3600: what the approach can serve per hour at its own headway;  v/c = demand / cap
  blocked s    sat s in which nobody crossed although the light was green and the first vehicle stood: spill-back from downstream or a
               yielding minor link -- capacity the signal handed out and the approach could not use

  python tools/saturation_audit.py [map] [--env 0] [--top 14] > profiles/r06_saturation_audit.txt
"""
import argparse
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleEnv, build       # noqa: E402
from resco_amd.scenario import Scenario             # noqa: E402


def audit(name, envi, seed=0, scale=None):
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    A = sc.arrays
    env = OracleEnv(sc, env_index=envi, seed=seed, sigma=-1.0, speed_dev=1, max_distance=200, fixed_program=1, trip_log=1)
    env.observe()
    nl = sc.n_lanes
    lanes = [l for l in range(nl) if not A['lane_internal'][l] and A['lane_link_cnt'][l] > 0 and A['link_tls'][A['lane_link_start'][l]] >= 0]
    link0 = {l: int(A['lane_link_start'][l]) for l in lanes}
    R = {l: dict(demand=0, green=0, sat=0, crossed_sat=0, crossed=0, blocked=0, stand=0.0, occ=0.0) for l in lanes}
    prev_on = {}
    seen = {l: set() for l in lanes}

    def link_state(link):
        s = int(A['link_tls'][link])
        ph = env.get_phase(s)
        return int(A['fix_states'][A['fix_state_off'][s] + ph * A['tls_nlinks'][s] + A['link_tls_pos'][link]])

    for t in range(3600):
        states = {l: link_state(link0[l]) for l in lanes}
        v = env.vehicles()
        hw = v['hw']
        on, per_lane = {}, {}
        for sl in range(hw):
            if v['lane'][sl] >= 0xFFFE:
                continue
            l = int(v['lane'][sl])
            on[int(v['trip'][sl])] = l
            per_lane.setdefault(l, []).append((float(v['pos'][sl]), float(v['speed'][sl])))
        env.tick()
        v2 = env.vehicles()
        on2 = {int(v2['trip'][sl]): int(v2['lane'][sl]) for sl in range(v2['hw']) if v2['lane'][sl] < 0xFFFE}
        for l in lanes:
            here = per_lane.get(l, [])
            for tr, pl in on.items():
                if pl == l and tr not in seen[l]:
                    seen[l].add(tr)
                    R[l]['demand'] += 1
            crossed = sum(1 for tr, pl in on.items() if pl == l and (tr not in on2 or A['lane_edge'][on2[tr]] != A['lane_edge'][l] or A['lane_internal'][on2[tr]]))
            R[l]['crossed'] += crossed
            R[l]['occ'] += len(here)
            R[l]['stand'] += sum(1 for p, s in here if s <= 0.1)
            if states[l] >= 2:
                R[l]['green'] += 1
                if len(here) >= 3:
                    head = max(here)
                    if A['lane_len'][l] - head[0] <= 15.0:
                        R[l]['sat'] += 1
                        R[l]['crossed_sat'] += crossed
                        if crossed == 0 and head[1] <= 0.1:
                            R[l]['blocked'] += 1
    st = env.stats()
    env.close()
    return sc, R, st


def loss_by_edge(name, envi, seed=0):
    """where the vehicles lose their time: sum over the ticks of (1 - v / speed limit) per vehicle, by edge (junction lanes together)"""
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    A = sc.arrays
    env = OracleEnv(sc, env_index=envi, seed=seed, sigma=-1.0, speed_dev=1, fixed_program=1)
    env.observe()
    nl = sc.n_lanes
    stand, occ, slow = np.zeros(nl), np.zeros(nl), np.zeros(nl)
    vmax = np.asarray(A['lane_vmax'])
    for t in range(3600):
        env.tick()
        v = env.vehicles()
        hw = v['hw']
        ln = v['lane'][:hw]
        m = ln < 0xFFFE
        li = ln[m].astype(int)
        np.add.at(occ, li, 1)
        np.add.at(stand, ln[m & (v['speed'][:hw] <= 0.1)].astype(int), 1)
        np.add.at(slow, li, np.clip(1 - v['speed'][:hw][m] / np.maximum(vmax[li], 0.1), 0, 1))
    env.close()
    by = {}
    for l in range(nl):
        e = int(A['lane_edge'][l])
        k = sc.edge_ids[e] if e >= 0 and not A['lane_internal'][l] else '(junction lanes)'
        d = by.setdefault(k, [0.0, 0.0, 0.0])
        d[0] += slow[l]; d[1] += stand[l]; d[2] += occ[l]
    return by, slow.sum(), stand.sum(), occ.sum()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('map', nargs='?', default='ingolstadt21')
    ap.add_argument('--env', type=int, default=0)
    ap.add_argument('--top', type=int, default=14)
    a = ap.parse_args()
    build()
    sc, R, st = audit(a.map, a.env)
    A = sc.arrays
    vt = np.asarray(A['vtype_params']).reshape(-1, 10)
    unit = float(np.median(vt[:, 0] + vt[:, 1]))
    tau = float(np.median(vt[:, 4]))
    print('# %s, FIXED programme, env %d: inserted %d arrived %d active %d pending %d; vehicle unit %.1f m, tau %.1f s' % (
        a.map, a.env, st['inserted'], st['arrived'], st['active'], st['pending'], unit, tau))
    print('%-22s %-12s %6s %6s %6s %8s %8s %8s %6s %5s %9s %7s %7s' % ('lane', 'tls', 'demand', 'green', 'sat s', 'crossed', 'h model', 'h Krauss', 'cap', 'v/c', 'blocked s', 'occ', 'stand'))
    rows = sorted(R.items(), key=lambda kv: -kv[1]['demand'])[:a.top]
    for l, r in rows:
        k = int(A['lane_link_start'][l])
        tls = sc.signal_ids[int(A['link_tls'][k])]
        via = int(A['link_via1'][k])
        vlim = float(A['lane_vmax'][l])
        if via >= 0:
            vlim = min(vlim, float(A['lane_vmax'][via]))
        hk = tau + unit / max(vlim, 0.1)
        hm = r['sat'] / r['crossed_sat'] if r['crossed_sat'] else float('inf')
        cap = 3600.0 / hm * r['green'] / 3600.0 if hm < 1e9 else 0.0
        print('%-22s %-12s %6d %6d %6d %8d %8.2f %8.2f %6.0f %5.2f %9d %7.1f %7.1f' % (sc.lane_ids[l], tls, r['demand'], r['green'], r['sat'], r['crossed_sat'], hm, hk, cap,
                                                                                r['demand'] / cap if cap else float('inf'), r['blocked'], r['occ'] / 3600.0, r['stand'] / 3600.0))
    by, tot, st_tot, occ_tot = loss_by_edge(a.map, a.env)
    print('\n# where the time is lost (sum over the ticks of 1 - v / speed limit per vehicle): %.0f veh-s in all, %.0f of them standing, %.0f vehicle-seconds on the network' % (tot, st_tot, occ_tot))
    for k, d in sorted(by.items(), key=lambda kv: -kv[1][0])[:20]:
        print('%-24s loss %8.0f (%4.1f %%)   standing %8.0f   vehicle-seconds %8.0f' % (k, d[0], 100 * d[0] / tot, d[1], d[2]))


if __name__ == '__main__':
    main()
