#!/usr/bin/env python3
"""vehicles leaving a lane forward per signal cycle under the FIXED programme (study tool): python oracle/study/percycle.py map lane cycle_s"""
import os, sys
import numpy as np
ROOT='/root/repo'; sys.path.insert(0, ROOT)
from oracle.pyoracle import OracleEnv
from resco_amd.scenario import Scenario
name=sys.argv[1]; lid=sys.argv[2]; cyc=int(sys.argv[3]); lid2=sys.argv[4] if len(sys.argv)>4 else None
sc = Scenario.load(os.path.join(ROOT,'resco_amd','scenarios',name+'.npz')); A=sc.arrays
l=sc.lane_ids.index(lid)
env = OracleEnv(sc, env_index=0, seed=0, sigma=-1.0, speed_dev=1, fixed_program=1)
prev=set(); cnt=np.zeros(3600//cyc+1,int); q=np.zeros(3600//cyc+1,int)
for t in range(3600):
    env.tick(); v=env.vehicles(); hw=v['hw']
    cur=set(v['trip'][:hw][v['lane'][:hw]==l].tolist())
    # left forward = not on any lane of same edge now
    e=A['lane_edge'][l]; same=set(v['trip'][:hw][(v['lane'][:hw]<0xFFFE)&(A['lane_edge'][np.minimum(v['lane'][:hw],sc.n_lanes-1)]==e)].tolist())
    cnt[t//cyc]+=len([k for k in prev-cur if k not in same]); q[t//cyc]=max(q[t//cyc],len(cur))
    prev=cur
print('served per cycle:', cnt.tolist(), 'total', cnt.sum())
