#!/usr/bin/env python3
"""In-kernel phase timers of the step kernel (rs_phase_profile): share of every phase in a workgroup's wall time.
python tools/phase_profile.py [map] [envs] [block] [--sections]
--sections: the library in RESCO_SIM_LIB is a study build (tools/ab.py build "sec:-DRS_STUDY_SECTIONS"): the role counters hold
the time a wave spends in the sections of the long path of the plan (one look-ahead chunk) instead."""
import os, sys
SECTIONS = '--sections' in sys.argv
if SECTIONS:
    sys.argv.remove('--sections')
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario
from resco_amd.sim import BatchedSim
name = sys.argv[1] if len(sys.argv) > 1 else 'ingolstadt21'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
n_envs = n
block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
if os.environ.get('RS_CAPACITY'):
    sc.capacity = int(os.environ['RS_CAPACITY'])
sim = BatchedSim(sc, n, seed=0, sigma=-1.0, speed_dev=1, block_threads=block)
for k in range(100):
    sim.act_random(k); sim.step(None)
sim.phase_profile(True)
for k in range(100, 160):
    sim.act_random(k); sim.step(None)
acc = sim.phase_profile(False)
names = ['L0 init', 'L1 load+prep', 'L2 register', '-', 'P plan+lc', 'C insert?+tls', 'M move+insert', '-', '-', '-', '-', 'O0', 'O1 observe', 'O2 outputs', 'O3', '-']
if SECTIONS:
    roles = {7: 'H plan: own records', 8: 'H plan: leader on the lane', 9: 'H plan: leader found + cooperation', 10: 'H plan: walk over the links', 3: 'H plan: safe speed + dawdling', 15: 'H plan: mover flag'}
else:
    roles = {7: 'P look-ahead list', 8: 'P lane-change list', 9: 'P slots', 10: 'M leavers list', 3: 'M slots', 15: 'M whole body'}
tot = float(sum(a for i, a in enumerate(acc) if i not in roles)) or 1.0
for i, (nm, a) in enumerate(zip(names, acc)):
    if i not in roles and nm != '-':
        print('%-14s %6.2f %%' % (nm, 100.0 * a / tot))
for i, nm in roles.items():
    cnt, ticks = acc[i] >> 40, acc[i] & ((1 << 40) - 1)
    if cnt:
        print('wave role %-20s %6.2f us per wave per tick (%.2f waves per tick)' % (nm, ticks / cnt / 100.0, cnt / (60.0 * ((n_envs + 15) // 16) * 10)))
print('total ticks (100 MHz) per env-step per workgroup: %.0f' % (tot / 60 / n))
