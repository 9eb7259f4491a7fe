#!/usr/bin/env python3
"""Volume / capacity of every signal-controlled approach lane under the net's own (FIXED) programme, next to what the
oracle serves there (study tool, TEST INFRASTRUCTURE).

  demand    trips per hour whose route uses the lane's connection(s) (static fastest paths, like SUMO's <trip> routing)
  green     seconds of G / g per cycle for the lane's busiest connection
  cap       green / cycle * 3600 / H veh/h, H = saturation headway (default 2.0 s: Krauss tau 1 s + length + minGap at ~8 m/s)
  served    vehicles that left the lane in the oracle's episode
  halt      halted vehicle-seconds on the lane (v <= 0.1 m/s)
  loss      time-loss seconds collected on the lane  (sum (1 - v / vmax))

  python oracle/study/vc_table.py ingolstadt21 [--policy FIXED] [--top 30]
"""
import argparse, os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleEnv
from resco_amd.scenario import Scenario

ap = argparse.ArgumentParser()
ap.add_argument('map')
ap.add_argument('--top', type=int, default=30)
ap.add_argument('--headway', type=float, default=2.0)
ap.add_argument('--env', type=int, default=0)
args = ap.parse_args()
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', args.map + '.npz')); A = sc.arrays
nl = sc.n_lanes
# demand per link: every trip walks its route; the link is the one choose_link would take is lane dependent, so count per (edge -> edge)
edge_dem = {}
for k in range(sc.n_trips):
    rt = A['trip_route'][k]; r = A['route_edge'][A['route_start'][rt]:A['route_start'][rt + 1]]
    for i in range(len(r) - 1):
        edge_dem[(int(r[i]), int(r[i + 1]))] = edge_dem.get((int(r[i]), int(r[i + 1])), 0) + 1
env = OracleEnv(sc, env_index=args.env, seed=0, sigma=-1.0, speed_dev=1, fixed_program=1)
halt = np.zeros(nl); loss = np.zeros(nl); served = np.zeros(nl); occ = np.zeros(nl)
prev_lane = {}
T = sc.horizon
for t in range(T):
    env.tick()
    v = env.vehicles(); hw = v['hw']
    lane = v['lane'][:hw].astype(np.int64); act = lane < 0xFFFE
    sp = v['speed'][:hw]; trip = v['trip'][:hw]
    la = lane[act]
    np.add.at(occ, la, 1)
    np.add.at(halt, lane[act & (sp <= 0.1)], 1)
    vm = A['lane_vmax'][la]
    np.add.at(loss, la, np.clip(1.0 - sp[act] / vm, 0, 1))
    cur = dict(zip(trip[act].tolist(), la.tolist()))
    for k, l in prev_lane.items():
        if cur.get(k, -1) != l:
            served[l] += 1
    prev_lane = cur
hours = T / 3600.0
rows = []
for l in range(nl):
    if A['lane_internal'][l]:
        continue
    ls, lc = A['lane_link_start'][l], A['lane_link_cnt'][l]
    tl = [i for i in range(ls, ls + lc) if A['link_tls'][i] >= 0]
    if not tl:
        continue
    e = A['lane_edge'][l]
    s = A['link_tls'][tl[0]]
    m = sc.signal_meta[sc.signal_ids[s]]
    prog = m['orig_program']
    cyc = sum(d for d, _ in prog)
    # demand of the lane = demand of its edge pairs split evenly over the edge's lanes that have that connection
    dem = 0.0; green = 0
    for i in tl:
        pair = (int(e), int(A['link_to_edge'][i]))
        # lanes of the edge that connect to this next edge
        n_share = 0
        for l2 in range(A['edge_lane0'][e], A['edge_lane0'][e] + A['edge_nlanes'][e]):
            if any(A['link_to_edge'][j] == pair[1] for j in range(A['lane_link_start'][l2], A['lane_link_start'][l2] + A['lane_link_cnt'][l2])):
                n_share += 1
        # several links of one lane to the same edge (two destination lanes) must not double-count
        same = [j for j in tl if A['link_to_edge'][j] == pair[1]]
        dem += edge_dem.get(pair, 0) / max(1, n_share) / len(same)
        g = sum(d for d, st in prog if st[A['link_tls_pos'][i]] in 'Gg')
        green = max(green, g)
    cap = green / cyc * 3600.0 / args.headway
    rows.append((dem / hours / max(cap, 1e-9), sc.lane_ids[l], sc.signal_ids[s], dem / hours, green, cyc, cap, served[l] / hours, halt[l], loss[l], occ[l] / T))
rows.sort(reverse=True)
print('%-26s %-14s %7s %5s %4s %6s %5s %7s %8s %8s %5s' % ('lane', 'tls', 'demand', 'green', 'cyc', 'cap', 'v/c', 'served', 'halt', 'loss', 'occ'))
for vc, lid, sid, dem, g, cyc, cap, srv, h, ls_, oc in rows[:args.top]:
    print('%-26s %-14s %7.0f %5d %4d %6.0f %5.2f %7.0f %8.0f %8.0f %5.1f' % (lid, sid[:14], dem, g, cyc, cap, vc, srv, h, ls_, oc))
st = env.stats()
print('network: time loss %.0f veh-s, halted %.0f veh-s, inserted %d arrived %d' % (loss.sum(), halt.sum(), st['inserted'], st['arrived']))
tot = loss.sum()
tl_loss = sum(r[9] for r in rows)
print('time loss on signal-approach lanes: %.0f (%.0f%%); on lanes with v/c > 0.9: %.0f (%.0f%%)' % (tl_loss, 100 * tl_loss / tot, sum(r[9] for r in rows if r[0] > 0.9), 100 * sum(r[9] for r in rows if r[0] > 0.9) / tot))
