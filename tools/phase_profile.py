#!/usr/bin/env python3
"""Per-phase time split of rs_step_kernel (in-kernel wall_clock64 timers, summed over workgroups)."""
import os, sys
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario
from resco_amd.sim import BatchedSim
name = sys.argv[1] if len(sys.argv) > 1 else 'ingolstadt21'
n = int(sys.argv[2]) if len(sys.argv) > 2 else 4096
block = int(sys.argv[3]) if len(sys.argv) > 3 else 0
sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
sim = BatchedSim(sc, n, seed=0, block_threads=block)
for k in range(100):
    sim.act_random(k); sim.step(None)
sim.phase_profile(True)
for k in range(100, 140):
    sim.act_random(k); sim.step(None)
acc = sim.phase_profile(False)
names = ['load', 'prologue', 'A cand+approach', 'B insert', 'C plan', 'clear heads', 'D move', 'E lane change', 'F rebuild', 'observe+store', 'outputs']
tot = sum(acc[:11])
for nm, v in zip(names, acc):
    print('%-18s %6.2f %%   %.1f us per WG per env-step' % (nm, 100.0 * v / tot, v / 100.0 / (n * 40)))
print('total per WG per env-step: %.1f us' % (tot / 100.0 / (n * 40)))
