#!/usr/bin/env python3
"""Random search over the lane-change / junction constants of the oracle model against the reference's published
delays (utils/avg_timeLoss.py medians).  Study tool; results go to oracle/study/calib.log."""
import json, os, random, subprocess, sys, math
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SPACE = dict(X_LAV=[0, 1], X_LAT=[5, 8, 10, 14], X_LAB=[10, 30, 60], X_LAMIN=[2, 5, 8], X_URG=[30, 50, 80], X_GOOD=[200, 450, 900],
             X_SGA=[5, 10, 20], X_EXTRA=[1, 2, 3], X_COOPR=[30, 50, 80], X_FOE=[25, 30, 40], X_VIS=[0, 4.5], X_SWAPW=[10, 20, 40], X_ALT=[0, 1])
CASES = [(m, p) for m in ('cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7') for p in ('FIXED', 'MAXWAVE', 'MAXPRESSURE')] + [('ingolstadt21', 'FIXED')]
def evaluate(params, envs=8):
    env = dict(os.environ); env.update({k: str(v) for k, v in params.items()})
    rows = {}
    for m in sorted(set(c[0] for c in CASES)):
        pols = ','.join(p for mm, p in CASES if mm == m)
        out = subprocess.run([sys.executable, os.path.join(ROOT, 'oracle', 'delay_eval.py'), m, '--policies', pols, '--envs', str(envs)],
                             env=env, capture_output=True, text=True)
        for line in out.stderr.splitlines():
            try: r = json.loads(line)
            except Exception: continue
            rows[(r['map'], r['policy'])] = (r['avg_delay'], r['reference_delay'][1])
    score = 0.0; worst = 0.0
    for k, (d, ref) in rows.items():
        e = abs(math.log(max(d, 1e-3) / ref)); w = 2.0 if k[1] == 'FIXED' else 1.0
        score += w * e; worst = max(worst, e)
    return score, worst, rows
if __name__ == '__main__':
    random.seed(int(sys.argv[1]) if len(sys.argv) > 1 else 0)
    n = int(sys.argv[2]) if len(sys.argv) > 2 else 60
    log = open(os.path.join(ROOT, 'oracle', 'study', 'calib.log'), 'a')
    for it in range(n):
        params = {} if it == 0 else {k: random.choice(v) for k, v in SPACE.items()}
        score, worst, rows = evaluate(params)
        rec = dict(score=score, worst=worst, params=params, rows={'%s/%s' % k: round(v[0] / v[1], 3) for k, v in rows.items()})
        log.write(json.dumps(rec) + '\n'); log.flush()
