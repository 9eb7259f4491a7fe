#!/usr/bin/env python3
"""In-kernel phase timers of the step kernel (rs_phase_profile): share of every phase in a workgroup's wall time.
python tools/phase_profile.py [map] [envs] [block]"""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario
from resco_amd.sim import BatchedSim
name = sys.argv[1] if len(sys.argv) > 1 else 'ingolstadt21'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
sim = BatchedSim(sc, n, seed=0, sigma=-1.0, speed_dev=1, block_threads=block)
for k in range(100):
    sim.act_random(k); sim.step(None)
sim.phase_profile(True)
for k in range(100, 160):
    sim.act_random(k); sim.step(None)
acc = sim.phase_profile(False)
names = ['L0 init', 'L1 load+prep', 'L2 register', '-', 'P plan+lc', 'C insert?+tls', 'M move+insert', '-', '-', '-', '-', 'O0', 'O1 observe', 'O2 outputs', 'O3']
tot = float(sum(acc)) or 1.0
for nm, a in zip(names, acc):
    print('%-12s %6.2f %%' % (nm, 100.0 * a / tot))
print('total ticks (100 MHz) per env-step per workgroup: %.0f' % (tot / 60 / n))
