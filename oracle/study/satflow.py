#!/usr/bin/env python3
"""saturation flow of one lane under the FIXED programme: vehicles leaving per second while a long queue stands (study tool)"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleEnv
from resco_amd.scenario import Scenario
name, lid = sys.argv[1], sys.argv[2]
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz')); A = sc.arrays
l = sc.lane_ids.index(lid)
env = OracleEnv(sc, env_index=0, seed=0, sigma=-1.0, speed_dev=1, fixed_program=1)
prev = set(); leave = []; occ = []
for t in range(3600):
    env.tick()
    v = env.vehicles(); hw = v['hw']
    cur = set(v['trip'][:hw][(v['lane'][:hw] == l)].tolist())
    leave.append(len(prev - cur)); occ.append(len(prev))
    prev = cur
leave = np.array(leave); occ = np.array(occ)
# green periods = runs of ticks with leave>0 separated by >=8 ticks without
TH = int(os.environ.get("TH", "8")); busy = occ >= TH
print('total left', leave.sum(), 'mean occ', occ.mean())
# per cycle (90s) profile
cyc = int(sys.argv[3]) if len(sys.argv) > 3 else 90
prof = np.zeros(cyc); cnt = np.zeros(cyc)
for t in range(3600):
    if busy[t]: prof[t % cyc] += leave[t]; cnt[t % cyc] += 1
print('discharge profile over the cycle while occ>=8 (veh/s):')
print(np.round(prof / np.maximum(cnt, 1), 2))
