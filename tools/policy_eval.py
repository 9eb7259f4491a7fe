#!/usr/bin/env python3
"""Control-quality sanity band: average trip delay (timeLoss + departDelay) of the
static controllers on the HIP simulator, N environments x one full 360-step episode each, bench mode
(sigma 0.5, speedFactor dev 0.1).  The reference's published numbers come from SUMO; this build's
dynamics are its own model (parity unpinned), so only the order of magnitude / ranking is expected to
agree."""
import json
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario      # noqa: E402
from resco_amd.sim import BatchedSim         # noqa: E402

REF = {  # avg delay (s) over the published episodes of utils/avg_timeLoss.py: (mean, median)
    ('cologne1', 'FIXED'): (56.61, 56.85), ('cologne1', 'MAXWAVE'): (27.81, 27.94), ('cologne1', 'MAXPRESSURE'): (65.85, 31.09),
    ('cologne3', 'FIXED'): (46.35, 39.04), ('cologne3', 'MAXWAVE'): (90.67, 21.95), ('cologne3', 'MAXPRESSURE'): (162.28, 28.05),
    ('cologne8', 'FIXED'): (63.77, 64.21), ('cologne8', 'MAXWAVE'): (21.87, 21.85), ('cologne8', 'MAXPRESSURE'): (47.73, 29.71),
    ('ingolstadt1', 'FIXED'): (39.40, 39.47), ('ingolstadt1', 'MAXWAVE'): (28.32, 27.99), ('ingolstadt1', 'MAXPRESSURE'): (23.62, 23.61),
    ('ingolstadt7', 'FIXED'): (91.31, 91.45), ('ingolstadt7', 'MAXWAVE'): (80.56, 80.31), ('ingolstadt7', 'MAXPRESSURE'): (46.82, 46.41),
    ('ingolstadt21', 'FIXED'): (133.10, 130.37), ('ingolstadt21', 'MAXWAVE'): (76.32, 69.61),
    ('ingolstadt21', 'MAXPRESSURE'): (136.72, 115.61),
}


def run(name, policy, n=64, seed=0):
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    md = {'FIXED': 200, 'MAXWAVE': 50, 'MAXPRESSURE': 200, 'STOCHASTIC': 1}[policy]
    sim = BatchedSim(sc, n, seed=seed, max_distance=md, fixed_program=1 if policy == 'FIXED' else 0)
    for k in range(360):
        if policy == 'MAXWAVE':
            sim.act_maxwave(0)
        elif policy == 'MAXPRESSURE':
            sim.act_maxwave(1)
        elif policy == 'STOCHASTIC':
            sim.act_random(k)
        sim.step(None)
    st = sim.stats()
    arrived = np.maximum(1, st['arrived'])
    delay = sim.trip_delay()        # timeLoss + departDelay per inserted trip, unfinished trips included
    out = dict(map=name, policy=policy, envs=n, avg_delay=float(delay.mean()), median_delay=float(np.median(delay)),
               avg_duration=float((st['sum_duration'] / arrived).mean()), mean_active=float((st['active_ticks'] / st['ticks']).mean()),
               arrived=float(st['arrived'].mean()), inserted=float(st['inserted'].mean()), pending=float(st['pending'].mean()),
               trips=sc.n_trips, reference_delay=REF.get((name, policy)))
    sim.close()
    return out


if __name__ == '__main__':
    maps = sys.argv[1:] or ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
    res = []
    for m in maps:
        for pol in ('FIXED', 'MAXWAVE', 'MAXPRESSURE', 'STOCHASTIC'):
            r = run(m, pol)
            res.append(r)
            print(json.dumps(r), flush=True)
