#!/usr/bin/env python3
"""Where does the delay come from?  Per-lane halted seconds and the plan() reason of halted vehicles (oracle study tool)."""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleEnv
from resco_amd.scenario import Scenario

name = sys.argv[1] if len(sys.argv) > 1 else 'ingolstadt7'
fixed = int(sys.argv[2]) if len(sys.argv) > 2 else 1
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
env = OracleEnv(sc, env_index=0, seed=0, sigma=-1.0, speed_dev=1, fixed_program=fixed)
A = sc.arrays
halt = np.zeros(sc.n_lanes)
reason_halt = np.zeros((sc.n_lanes, 12))
occ = np.zeros(sc.n_lanes)
T = 3600
for t in range(T):
    env.tick()
    v = env.vehicles(); r, b = env.debug()
    hw = v['hw']
    lane = v['lane'][:hw]; act = lane < 0xFFFE
    sp = v['speed'][:hw]
    h = act & (sp <= 0.1)
    np.add.at(halt, lane[h], 1)
    np.add.at(occ, lane[act], 1)
    np.add.at(reason_halt, (lane[h], r[:hw][h]), 1)
st = env.stats()
print(st)
order = np.argsort(-halt)[:25]
print('lane id, len, nlanes(edge), internal, halted veh-s, mean occ, reasons[free,leader,wronglane,red,foe,nextlead,speedlim]')
for l in order:
    e = A['lane_edge'][l]
    print('%-28s len %6.1f k=%d/%d int=%d halt %7.0f occ %5.2f  %s' % (sc.lane_ids[l], A['lane_len'][l], l - A['edge_lane0'][e] if e >= 0 else -1, A['edge_nlanes'][e] if e >= 0 else 0, A['lane_internal'][l], halt[l], occ[l] / T, reason_halt[l][:9].astype(int)))
print('total halted veh-s', halt.sum(), 'by reason', reason_halt.sum(0).astype(int))
