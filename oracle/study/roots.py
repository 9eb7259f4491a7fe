#!/usr/bin/env python3
"""root causes of queues under FIXED: for every halted vehicle follow its blocker chain to the head and charge the second
to (head lane, head reason) (study tool)"""
import os, sys, collections
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleEnv
from resco_amd.scenario import Scenario
name = sys.argv[1]
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz')); A = sc.arrays
env = OracleEnv(sc, env_index=0, seed=0, sigma=-1.0, speed_dev=1, fixed_program=1)
names = ['free', 'leader', 'wronglane', 'red', 'foe', 'nextlead', 'speedlim', 'minorvis', 'coop', 'cooplead']
cause = collections.Counter()
for t in range(3600):
    env.tick()
    if t % 5: continue
    v = env.vehicles(); r, b = env.debug(); hw = v['hw']
    lane = v['lane'][:hw]; act = lane < 0xFFFE
    for s in np.nonzero(act & (v['speed'][:hw] <= 0.1))[0]:
        c = int(s); n = 0
        while r[c] in (1, 5, 8, 9) and n < 200:
            nb = int(b[c])
            if nb < 0 or nb >= hw or lane[nb] >= 0xFFFE or v['speed'][nb] > 0.1: break
            c = nb; n += 1
        cause[(sc.lane_ids[lane[c]], names[r[c]], n >= 200)] += 5
tot = sum(cause.values())
print('halted veh-s', tot)
for (l, why, loop), n in cause.most_common(25):
    print('%7d %5.1f%%  %-40s %s %s' % (n, 100.0 * n / tot, l, why, 'LOOP' if loop else ''))
