#!/usr/bin/env python3
"""lane_info.py map lane_id...: links, TLS programme and route demand of a lane (study tool)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario
name = sys.argv[1]
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz')); A = sc.arrays
# link demand
dem = np.zeros(len(A['link_to_lane']), int)
edge_dem = {}
for k in range(sc.n_trips):
    rt = A['trip_route'][k]; r = A['route_edge'][A['route_start'][rt]:A['route_start'][rt + 1]]
    for i in range(len(r) - 1):
        edge_dem[(r[i], r[i + 1])] = edge_dem.get((r[i], r[i + 1]), 0) + 1
for lid in sys.argv[2:]:
    l = sc.lane_ids.index(lid); e = A['lane_edge'][l]
    print(lid, 'len %.1f vmax %.1f' % (A['lane_len'][l], A['lane_vmax'][l]), 'edge lanes', A['edge_nlanes'][e])
    for i in range(A['lane_link_start'][l], A['lane_link_start'][l] + A['lane_link_cnt'][l]):
        t = A['link_tls'][i]
        print('   link', i, '->', sc.lane_ids[A['link_dest_lane'][i]], 'tls', t, sc.signal_ids[t] if t >= 0 else '-', 'pos', A['link_tls_pos'][i], 'minor', A['link_minor'][i],
              'cont', A['link_cont'][i], 'foes', A['link_foe_cnt'][i], 'edge-demand', edge_dem.get((e, A['link_to_edge'][i]), 0))
        if t >= 0:
            m = sc.signal_meta[sc.signal_ids[t]]
            print('      orig:', ' '.join('%d:%s' % (d, s[A['link_tls_pos'][i]]) for d, s in m['orig_program']))
