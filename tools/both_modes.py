#!/usr/bin/env python3
"""GPU box: every reference-held result cell under BOTH answers to "what does trafficlight.setPhase leave behind"
(tls_expiry 0 = rs_params.tls_hold 1: the phase stays until the next setPhase -- round 5's calibration variant; tls_expiry 1 = the library's
default: it expires after its programme duration and the programme continues -- what SUMO's MSSimpleTrafficLightLogic::changeStepAndDuration does).  64 environments x one whole episode per cell, the
median over the environments / the reference's figure (tests/golden/ref_bands.json).  FIXED does not depend on the parameter (the
net's own programme always runs on its durations) and is printed once.

  python tools/both_modes.py [--envs 64] > profiles/r06_reference_bands_both_modes.txt
"""
import argparse
import copy
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario      # noqa: E402
from resco_amd.sim import BatchedSim         # noqa: E402

MAPS = ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
MAXD = {'FIXED': 200, 'MAXWAVE': 50, 'MAXPRESSURE': 200, 'STOCHASTIC': 200}


def episode(sc, policy, n_envs, tls_expiry, seed=0):
    sim = BatchedSim(sc, n_envs, seed=seed, max_distance=MAXD[policy.rstrip('*')], fixed_program=1 if policy == 'FIXED' else 0,
                     tls_expiry=tls_expiry)
    q = np.zeros(n_envs)
    for k in range(360):
        if policy.startswith('MAX'):
            sim.act_maxwave(1 if policy.startswith('MAXPRESSURE') else 0)
        elif policy == 'STOCHASTIC':
            sim.act_random(k)
        sim.step(None)
        q += sim.read('queue_sum').sum(axis=1) / (sc.n_signals + 1.0)
    m = sim.trip_metrics()
    m['queue'] = q / 360.0
    blocked = int(sim.stats()['cap_blocked'].sum())
    sim.close()
    out = {k: float(np.median(v)) for k, v in m.items()}
    out['cap_blocked'] = blocked
    return out


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--envs', type=int, default=64)
    ap.add_argument('--maps', default=','.join(MAPS))
    a = ap.parse_args()
    with open(os.path.join(ROOT, 'tests', 'golden', 'ref_bands.json')) as f:
        ref = json.load(f)
    print('# map policy metric | reference | phase stays (tls_expiry 0: round 5\'s default): value (ratio) | phase expires (tls_expiry 1, SUMO\'s documented setPhase: the default): value (ratio)   [cap!: insertions were refused because all vehicle slots were taken]')
    err = {0: [], 1: []}
    inband = {0: 0, 1: 0}
    ncell = 0
    for name in a.maps.split(','):
        sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
        pols = [('FIXED', sc), ('MAXWAVE', sc), ('MAXPRESSURE', sc), ('STOCHASTIC', sc)]
        if name == 'ingolstadt21':
            sc2 = copy.copy(sc)
            sc2.valid_acts = dict(sc.valid_acts)
            sc2.valid_acts['243641585'] = {4: 0, 7: 1, 2: 2}
            pols += [('MAXWAVE*', sc2), ('MAXPRESSURE*', sc2)]
        for policy, s in pols:
            modes = (0,) if policy == 'FIXED' else (0, 1)
            res = {x: episode(s, policy, a.envs, x) for x in modes}
            metrics = ('delay', 'duration', 'waiting', 'queue') if policy == 'STOCHASTIC' else ('delay',)
            for metric in metrics:
                t = ref[name][policy.rstrip('*')][metric]
                cols = []
                for x in (0, 1):
                    r = res[x if x in res else 0]
                    ratio = r[metric] / t
                    cols.append('%8.2f (%.2f)%s' % (r[metric], ratio, ' cap!' if r['cap_blocked'] else ''))
                    if not policy.endswith('*'):
                        err[x].append(abs(np.log(ratio)))
                        inband[x] += 0.65 <= ratio <= 1.35
                ncell += not policy.endswith('*')
                print('%-13s %-12s %-9s | %8.2f | %s | %s' % (name, policy, metric, t, cols[0], cols[1]), flush=True)
    print('# cells (as configured): %d; inside +-35 %%: %d without expiry, %d with; sum |log ratio|: %.2f without, %.2f with; median |log ratio|: %.3f / %.3f'
          % (ncell, inband[0], inband[1], sum(err[0]), sum(err[1]), float(np.median(err[0])), float(np.median(err[1]))))


if __name__ == '__main__':
    main()
