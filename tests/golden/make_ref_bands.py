#!/usr/bin/env python3
"""Extract the reference-held result figures into tests/golden/ref_bands.json (BUILD CONTAINER ONLY).

The reference ships no tests, but it does hold measured results of its own SUMO runs: per-episode averages in
resco_benchmark/utils/avg_timeLoss.py, avg_duration.py, avg_waitingTime.py, avg_queue.py (written by
utils/readXML.py:16-114 and utils/readCSV.py:30-80 from tripinfo_<run>.xml / metrics_<run>.csv).  They are the only
reference-held numbers that depend on SUMO's dynamics, so they are what pins this build's microsimulation model (the
arrays themselves do not travel; this script reduces them to the handful of figures the tests band against):

  <map>.FIXED / MAXWAVE / MAXPRESSURE .delay   median over the published episodes of avg_timeLoss.py
  <map>.STOCHASTIC.{delay,duration,waiting,queue}
        episode 1 of the IDQN row of each array: IDQN's epsilon decays linearly from 1 over 80 episodes
        (agents/pfrl_dqn.py:65-70, config/agent_config.py:83-94, main.py:91-92), so in episode 1 every signal acts
        uniformly at random with probability >= 0.9875 -- the uniform random policy seen through IDQN's 200 m detectors;
        `spread` = the relative spread of episodes 1-3 of IDQN / IPPO / FMA2C / MPLight (all near-random at that point)
  <map>.free_flow_residual
        median of (duration - delay) over the last 10 IDQN episodes (a trained policy: little departDelay): the travel
        time of the routes at the speed limits, independent of the controller

  python tests/golden/make_ref_bands.py
"""
import json
import os

import numpy as np

REF = '/root/reference/resco_benchmark/utils'
HERE = os.path.dirname(os.path.abspath(__file__))
MAPS = ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
FILES = {'delay': ('avg_timeLoss.py', 'delays'), 'duration': ('avg_duration.py', 'durations'),
         'waiting': ('avg_waitingTime.py', 'waiting'), 'queue': ('avg_queue.py', 'queue')}


def load(fname, var):
    ns = {}
    with open(os.path.join(REF, fname)) as f:
        exec(f.read(), {'np': np, 'array': np.array, '__name__': 'ref'}, ns)     # the files are plain dict literals
    return ns[var]


def row(d, prefix):
    for k, v in d.items():
        if k.startswith(prefix) and not k.endswith('_yerr'):
            a = np.asarray(v, float)
            if a.size:
                return a
    return None


def main():
    D = {m: load(*fv) for m, fv in FILES.items()}
    out = {}
    for mp in MAPS:
        o = {}
        for pol in ('FIXED', 'MAXWAVE', 'MAXPRESSURE'):
            a = row(D['delay'], '%s %s ' % (pol, mp))
            o[pol] = dict(delay=round(float(np.median(a)), 2), delay_mean=round(float(a.mean()), 2), episodes=int(a.size))
        st = {}
        for metric in FILES:
            a = row(D[metric], 'IDQN %s ' % mp)
            st[metric] = round(float(a[0]), 2)
            early = [row(D[metric], '%s %s ' % (ag, mp)) for ag in ('IDQN', 'IPPO', 'FMA2C', 'MPLight')]
            early = np.concatenate([e[:3] for e in early if e is not None])
            st[metric + '_early_min'] = round(float(early.min()), 2)
            st[metric + '_early_max'] = round(float(early.max()), 2)
        o['STOCHASTIC'] = st
        dur, dly = row(D['duration'], 'IDQN %s ' % mp), row(D['delay'], 'IDQN %s ' % mp)
        o['free_flow_residual'] = round(float(np.median(dur[-10:] - dly[-10:])), 2)
        o['trained_best'] = {m: round(float(row(D[m], 'IDQN %s ' % mp).min()), 2) for m in FILES}
        out[mp] = o
    with open(os.path.join(HERE, 'ref_bands.json'), 'w') as f:
        json.dump(out, f, indent=1, sort_keys=True)
    print(json.dumps(out, indent=1, sort_keys=True))


if __name__ == '__main__':
    main()
