#!/usr/bin/env python3
"""Coordinate search over the constants of the microsimulation model (include/resco_model.h, all overridable with -DRM_...)
against EVERY reference-held figure (tests/golden/ref_bands.json): the 18 FIXED / MAXWAVE / MAXPRESSURE delay medians (ingolstadt21's
greedy controllers with the repaired valid_acts entry) and the six random-policy delays.  Study tool (oracle only, 8 s per
evaluation on 8 cores); results go to oracle/study/calib2.log.  The shipped constants are changed by hand, if at all.

  python oracle/study/calibrate2.py [rounds]
"""
import json
import math
import os
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
SPACE = {
    'RM_LOOK_TIME': ['6.0f', '8.0f', '10.0f', '12.0f'], 'RM_LOOK_BASE': ['5.0f', '10.0f', '20.0f'],
    'RM_LOOK_MIN_SPEED': ['3.0f', '5.0f', '8.0f'], 'RM_OCC_FACTOR': ['0.7f', '1.0f', '1.3f'],
    'RM_URGENT_DIST': ['50.0f', '80.0f', '120.0f'], 'RM_COOP_RANGE': ['40.0f', '80.0f', '120.0f'],
    'RM_FOE_GAP_Q': ['25', '30', '40', '50'], 'RM_VIS_DIST': ['4.5f', '8.0f'], 'RM_STOP_OFFSET': ['0.5f', '1.0f'],
    'RM_SG_ADVANTAGE': ['10.0f', '20.0f', '40.0f'], 'RM_SG_EXTRA_LANES': ['1', '2', '3'], 'RM_GOOD_CONT': ['120.0f', '200.0f', '400.0f'],
    'RM_SWAP_WAIT': ['10', '20', '40'], 'RM_MIN_LC_LEN': ['5.0f', '12.5f'],
}
DEFAULT = {'RM_LOOK_TIME': '8.0f', 'RM_LOOK_BASE': '10.0f', 'RM_LOOK_MIN_SPEED': '5.0f', 'RM_OCC_FACTOR': '1.0f', 'RM_URGENT_DIST': '80.0f',
           'RM_COOP_RANGE': '80.0f', 'RM_FOE_GAP_Q': '40', 'RM_VIS_DIST': '4.5f', 'RM_STOP_OFFSET': '1.0f', 'RM_SG_ADVANTAGE': '20.0f',
           'RM_SG_EXTRA_LANES': '2', 'RM_GOOD_CONT': '200.0f', 'RM_SWAP_WAIT': '20', 'RM_MIN_LC_LEN': '5.0f'}
CELLS = [(m, p) for m in ('cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7') for p in ('FIXED', 'MAXWAVE', 'MAXPRESSURE', 'STOCHASTIC')] + \
        [('ingolstadt21', p) for p in ('FIXED', 'MAXWAVE*', 'MAXPRESSURE*', 'STOCHASTIC')]
WEIGHT = {'STOCHASTIC': 0.5}


def evaluate(params):
    defs = ' '.join('-D%s=%s' % kv for kv in params.items() if DEFAULT[kv[0]] != kv[1])
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-B', '-s', 'ORC_DEFS=' + defs], stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    code = ('import sys, json; sys.path.insert(0, %r); import multiprocessing as mp; import oracle.fidelity_eval as F\n'
            'RB = F.ref_bands(); out = {}\n'
            'with mp.get_context("fork").Pool(8) as pool:\n'
            '    for m, p in %r:\n'
            '        r = F.run(m, p, 8, 0, 360, pool); out[m + "/" + p] = r["delay"] / RB[m][p.rstrip("*")]["delay"]\n'
            'print(json.dumps(out))\n' % (ROOT, CELLS))
    res = subprocess.run([sys.executable, '-c', code], capture_output=True, text=True)
    rows = json.loads(res.stdout.strip().splitlines()[-1])
    score = sum(WEIGHT.get(k.split('/')[1], 1.0) * abs(math.log(max(v, 1e-3))) for k, v in rows.items())
    return score, rows


def main():
    rounds = int(sys.argv[1]) if len(sys.argv) > 1 else 2
    log = open(os.path.join(ROOT, 'oracle', 'study', 'calib2.log'), 'a')
    cur = dict(DEFAULT)
    best, rows = evaluate(cur)
    log.write(json.dumps(dict(score=best, params={}, rows=rows)) + '\n'); log.flush()
    print('default score %.3f' % best, flush=True)
    for rd in range(rounds):
        improved = False
        for name, values in SPACE.items():
            for v in values:
                if v == cur[name]:
                    continue
                trial = dict(cur); trial[name] = v
                sc, rows = evaluate(trial)
                log.write(json.dumps(dict(score=sc, params={k: x for k, x in trial.items() if DEFAULT[k] != x}, rows=rows)) + '\n'); log.flush()
                print('%s=%s score %.3f%s' % (name, v, sc, '  <-- better' if sc < best - 0.02 else ''), flush=True)
                if sc < best - 0.02:
                    best, cur, improved = sc, trial, True
        if not improved:
            break
    print('best', best, {k: x for k, x in cur.items() if DEFAULT[k] != x})
    subprocess.check_call(['make', '-C', os.path.join(ROOT, 'oracle'), '-B', '-s'])


if __name__ == '__main__':
    main()
