#!/usr/bin/env python3
"""GPU box: the held-out IDQN check (tests/test_gpu_heldout.py: the reference's network and hyper-parameters trained from scratch on this
simulator, three learner seeds per map, best training episode / the reference's best of 100) under BOTH values of tls_expiry.
  python tools/heldout_both_modes.py [maps] > profiles/r06_heldout_both_modes.txt"""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
sys.path.insert(0, os.path.join(ROOT, 'tools'))
sys.path.insert(0, os.path.join(ROOT, 'tests'))
import idqn_train                                   # noqa: E402
from test_gpu_heldout import RUNS, SEEDS            # noqa: E402

with open(os.path.join(ROOT, 'tests', 'golden', 'ref_bands.json')) as f:
    REF = json.load(f)
maps = sys.argv[1].split(',') if len(sys.argv) > 1 else list(RUNS)
print('# map | tls_expiry | best training episode of seeds %s (s) | reference best of 100 | ratio median (min, max) | mean of last five episodes / reference | random policy (s)' % (SEEDS,))
for name in maps:
    envs, episodes = RUNS[name]
    target = REF[name]['trained_best']['delay']
    for mode in (0, 1):
        best, last5, rnd = [], [], []
        for seed in SEEDS:
            rows, final = idqn_train.main(name, envs, episodes, 256, 1, True, 10000, 0.0, evaluate=False, quiet=True, seed=seed, tls_expiry=bool(mode))
            best.append(final['best_training_episode_delay_s'])
            last5.append(float(np.mean([r['avg_delay_s'] for r in rows[-5:]])))
            rnd.append(final['random_avg_delay_s'])
        r = [b / target for b in best]
        print('%-13s | %d | %s | %.1f | %.2f (%.2f, %.2f) | %s | %.0f' % (name, mode, ' / '.join('%.1f' % b for b in best), target, float(np.median(r)), min(r), max(r),
                                                                      ' / '.join('%.2f' % (x / target) for x in last5), float(np.mean(rnd))), flush=True)
