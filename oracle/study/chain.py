#!/usr/bin/env python3
"""follow the chain of blockers of the longest-waiting vehicle (oracle study tool)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleEnv
from resco_amd.scenario import Scenario
from oracle.delay_eval import run_env
name = sys.argv[1]; policy = sys.argv[2]; T = int(sys.argv[3]) if len(sys.argv) > 3 else 3600
env, sc = run_env(name, policy, int(os.environ.get('ENV', '0')), 0, T // 10)
A = sc.arrays
v = env.vehicles(); r, b = env.debug(); hw = v['hw']
names = ['free', 'leader', 'wronglane', 'red', 'foe', 'nextlead', 'speedlim', 'minorvis', 'coop', 'cooplead']
def desc(s):
    l = v['lane'][s]; e = A['lane_edge'][l]; k = v['trip'][s]; rt = A['trip_route'][k]; rs = A['route_start'][rt]
    kk = l - A['edge_lane0'][e] if e >= 0 else -1
    m2 = np.round(A['route_cont'][rs + v['cursor'][s]], 0).astype(int).tolist()
    return 'slot %d trip %d lane %s(k=%d/%d) pos %.1f/%.1f v %.1f wait %d reason %s blk %d cont %s' % (s, k, sc.lane_ids[l], kk, A['edge_nlanes'][e] if e >= 0 else 0, v['pos'][s], A['lane_len'][l], v['speed'][s], v['sumo_wait'][s], names[r[s]], b[s], m2)
act = v['lane'][:hw] < 0xFFFE
w = np.where(act, v['sumo_wait'][:hw].astype(np.int64), -1)
s = int(np.argmax(w))
seen = set()
while s >= 0 and s not in seen:
    seen.add(s)
    if v['lane'][s] >= 0xFFFE:
        print('slot', s, 'is free now'); break
    print(desc(s))
    if r[s] in (1, 5, 8): s = int(b[s])
    elif r[s] in (3, 4, 7):
        print('   link', b[s], 'tls', A['link_tls'][b[s]], 'pos', A['link_tls_pos'][b[s]], 'minor', A['link_minor'][b[s]], 'cont', A['link_cont'][b[s]], 'to', sc.lane_ids[A['link_to_lane'][b[s]]]); break
    else: break
w_, c_, pl = env.backlog_delay(True)
print('backlog', c_, 'trips;', [(sc.lane_ids[l], int(pl[l])) for l in np.argsort(-pl)[:5] if pl[l] > 0])
# longest waiters
for s in np.argsort(-w)[:12]:
    if w[s] > 60: print('  ', desc(int(s)))
if len(sys.argv) > 4:
    sig = int(sys.argv[4])
    print('signal', sig, 'phase', env.outputs()['phase'][sig], 'wave', env.outputs()['wave'][sig])
    o0, o1 = A['sig_obs_start'][sig], A['sig_obs_start'][sig + 1]
    for oi in range(o0, o1):
        l = A['obs_lane'][oi]
        print(' lane', sc.lane_ids[l], 'agg', env.outputs()['lane_agg'][oi])
        for s in range(hw):
            if v['lane'][s] == l: print('      ', desc(s))
