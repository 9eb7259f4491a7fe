#!/usr/bin/env python3
"""Queue discharge and what a random policy does to it -- the study behind the round-5 model change (CPU oracle = test
infrastructure; nothing here is on the product path).

  1. known answers: a standing queue released by a green of 7 / 17 / 27 s (oracle/discharge.py: oracle == an independent float64
     restatement of the Krauss formulas, asserted in tests/test_discharge.py)
  2. the capacity of ONE lane of cologne1 under the uniformly random policy, for both answers to "what does setPhase leave behind"
     (rs_params.tls_expiry)
  3. the running cologne1 / cologne3 episodes under that policy, sampled per green window and lane: vehicles per green, start-up
     loss, stranded lane changers, permissive-left stalls, the inventory against what the reference's figures imply
  4. does the model ever gridlock on cologne3 (SUMO runs with --time-to-teleport -1, multi_signal.py:127)?

  python tools/discharge_study.py > profiles/r05_discharge_study.txt
"""
import os
import sys

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from oracle import discharge as D                                # noqa: E402
from oracle.pyoracle import OracleEnv, lib                       # noqa: E402
from resco_amd.scenario import Scenario                          # noqa: E402

REASON = ['free', 'leader', 'wrong lane', 'red / yellow', 'foe (permissive)', 'leader beyond the junction', 'speed limit ahead',
          'minor link: not yet visible', 'lets a lane changer in', 'falls in behind its target']


def load(name):
    return Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))


def ref_bands():
    import json
    with open(os.path.join(ROOT, 'tests', 'golden', 'ref_bands.json')) as f:
        return json.load(f)


def random_action(seed, env, s, k, G):
    return lib().orc_hash((seed ^ 0xA5A5A5A5) & 0xFFFFFFFF, env, s, k, 7) % G


def part1():
    print('## 1. standing queue of 20 cars (length 4.3, minGap 1.5, accel 2.6, decel 4.5, tau 1) on cologne1 -32038056#3_0, right turn on G')
    print('#     second of the green (0 = its first tick) in which car k crosses the stop line; y = crossed under the 3 s yellow; - = stopped')
    for green in (7, 17, 27):
        o, ref, worst = D.compare(20, green)
        row = ' '.join(('%2d%s' % (c, 'y' if c >= green else ' ')) if c is not None else ' - ' for c in o['cross'])
        n = len([c for c in o['cross'] if c is not None])
        print('green %2d s, sigma 0  : %s   -> %2d cars; oracle == restatement: %s (largest position difference %.4f m)' % (green, row, n, o['cross'] == ref, worst))
    for green in (7, 17, 27):
        ns = []
        for seed in range(20):
            o = D.oracle_discharge(30, green, sigma=0.5, seed=seed)
            ns.append(len([c for c in o['cross'] if c is not None]))
        print('green %2d s, sigma 0.5: %.2f cars on average over 20 seeds (min %d, max %d)' % (green, np.mean(ns), min(ns), max(ns)))
    o, _, _ = D.compare(20, 27)
    c = [x for x in o['cross'] if x is not None]
    print('saturation headway at sigma 0 from the 8th car on: %.2f s; the first four cars need 7 s (start-up loss %.1f s)' %
          ((c[-1] - c[7]) / (len(c) - 8), 7 - 4 * (c[-1] - c[7]) / (len(c) - 8)))
    print('last car at yellow onset: stops iff its distance to the line >= brakeGap(v) (tests/test_discharge.py::test_last_vehicle_at_yellow_onset)')


def part2():
    print('\n## 2. capacity of ONE lane of cologne1 under the uniformly random policy (4 greens: N-S 29 s, N-S left 6 s, W-E 29 s, W-E left 6 s;')
    print('#     action every 10 s, 3 s yellow): 1700 cars offered on one route, cars through in the hour, 4 environments')
    for app, tgt, label in (('-32038056#3', '32038051#0', 'W approach lane 0, right turn (green in W-E = phase 2)'),
                            ('23429231#1', '32038056#0', 'N approach lane 0, right turn (green in N-S = phase 0)')):
        for sigma in (0.0, 0.5):
            row = []
            for expiry in (0, 1):
                sc = D.queue_scenario(1700, 2, approach=app, target=tgt)
                arr = []
                for envi in range(4):
                    env = OracleEnv(sc, env_index=envi, seed=0, sigma=sigma, speed_dev=1, max_distance=200, tls_expiry=expiry)
                    env.observe()
                    for k in range(360):
                        env.step(np.array([random_action(0, envi, 0, k, 4)], np.int32))
                    arr.append(env.stats()['arrived'])
                    env.close()
                row.append(np.mean(arr))
            print('%-58s sigma %.1f: phase stays %4.0f /h   phase expires %4.0f /h  (%+.0f %%)' % (label, sigma, row[0], row[1], 100 * (row[1] / row[0] - 1)))
    print('# with expiry the 6 s N-S-left green (index 1) hands its 7th second to index 2 = the W-E green: one car per lane starts, and when')
    print('# the next action is 2 the green simply continues.  cologne1 demand: W 571 /h on two lanes, N 688 /h, E 438 /h, S 316 /h.')


def episode(name, envi, expiry, sample_lanes=None, steps=360, seed=0):
    """the reference's step loop (prep -> yellow ticks -> set -> green ticks -> observe) tick by tick with per-lane sampling"""
    sc = load(name)
    A = sc.arrays
    env = OracleEnv(sc, env_index=envi, seed=seed, sigma=-1.0, speed_dev=1, max_distance=200, trip_log=1, tls_expiry=expiry)
    env.observe()
    S = sc.n_signals
    G = [int(g) for g in A['tls_ngreen']]
    nl = sc.n_lanes
    lanes = [l for l in range(nl) if not A['lane_internal'][l] and A['lane_link_cnt'][l] > 0 and
             A['link_tls'][A['lane_link_start'][l]] >= 0] if sample_lanes is None else sample_lanes
    main_link = {l: int(A['lane_link_start'][l]) for l in lanes}
    occ = np.zeros(nl)
    stand = np.zeros(nl)
    head_reason = np.zeros((nl, 10))
    windows = {l: [] for l in lanes}          # (length, queue at start, crossed during the green + the yellow after it)
    cur_win = {l: None for l in lanes}
    prev_on = {}
    max_stand, long_standers = 0, 0
    qsum = 0.0

    def link_state(link):
        s = int(A['link_tls'][link])
        ph = env.get_phase(s)
        return int(A['tls_states'][A['tls_state_off'][s] + ph * A['tls_nlinks'][s] + A['link_tls_pos'][link]])

    for k in range(steps):
        acts = [random_action(seed, envi, s, k, G[s]) for s in range(S)]
        for s in range(S):                                  # Signal.prep_phase
            cur = env.get_phase(s)
            if cur != acts[s] and cur < G[s]:
                y = int(A['tls_yellow'][A['tls_yel_off'][s] + cur * G[s] + acts[s]])
                if y >= 0:
                    env.set_phase(s, y)
        for tick in range(10):
            if tick == 3:
                for s in range(S):
                    env.set_phase(s, acts[s])               # Signal.set_phase
            states = {l: link_state(main_link[l]) for l in lanes}       # what this tick's plan will see
            env.tick()
            v = env.vehicles()
            hw = v['hw']
            r, _ = env.debug()
            on = {}
            heads = {}
            for sl in range(hw):
                if v['trip'][sl] < 0 or v['lane'][sl] >= 0xFFFE:
                    continue
                l = int(v['lane'][sl])
                occ[l] += 1
                if v['speed'][sl] <= 0.1:
                    stand[l] += 1
                on[int(v['trip'][sl])] = l
                if l not in heads or v['pos'][sl] > v['pos'][heads[l]]:
                    heads[l] = sl
                w = int(v['sumo_wait'][sl])
                if w > max_stand:
                    max_stand = w
            for l, sl in heads.items():
                head_reason[l, r[sl]] += 1
            for l in lanes:
                crossed = sum(1 for t, pl in prev_on.items() if pl == l and (t not in on or A['lane_edge'][on[t]] != A['lane_edge'][l]
                                                                                 or A['lane_internal'][on[t]]))
                green = states[l] >= 2
                w = cur_win[l]
                if green:
                    if w is None or w['closed']:
                        if w is not None:
                            windows[l].append((w['len'], w['q0'], w['crossed'], w['first']))
                        q0 = sum(1 for t, pl in prev_on.items() if pl == l)
                        w = cur_win[l] = dict(len=0, q0=q0, crossed=0, closed=False, tail=0, first=None)
                    w['len'] += 1
                    if crossed and w['first'] is None:
                        w['first'] = w['len'] - 1
                    w['crossed'] += crossed
                elif w is not None:
                    if not w['closed'] and w['tail'] < 3:
                        w['tail'] += 1
                        w['crossed'] += crossed
                    else:
                        w['closed'] = True
            prev_on = on
        env.observe()
        qsum += float(env.outputs()['queue_sum'].sum()) / (S + 1)
    st = env.stats()
    v = env.vehicles()
    long_standers = int(((v['lane'][:v['hw']] < 0xFFFE) & (v['sumo_wait'][:v['hw']] >= 300)).sum())
    out = dict(sc=sc, occ=occ / (steps * 10), stand=stand / (steps * 10), head_reason=head_reason, windows=windows, stats=st,
               queue=qsum / steps, max_stand=max_stand, long_standers=long_standers,
               backlog=env.backlog_delay(per_lane=True)[2], mean_active=st['active_ticks'] / max(1, st['ticks']))
    env.close()
    return out


def part3(name):
    RB = ref_bands()[name]['STOCHASTIC']
    sc = load(name)
    n_trips = sc.n_trips
    implied = n_trips / 3600.0 * RB['duration']
    print('\n## 3. %s under the uniformly random policy, environment 0, one episode, sampled every tick' % name)
    print('#     reference (episode 1 of its IDQN runs): duration %.0f s -> %.0f vehicles in the network on average (Little); queue %.1f x (S + 1) = %.0f queued within 200 m' %
          (RB['duration'], implied, RB['queue'], RB['queue'] * (sc.n_signals + 1)))
    for expiry in (0, 1):
        e = episode(name, 0, expiry)
        A = sc.arrays
        print('\n### tls_expiry = %d (%s): %.0f vehicles in the network on average, %.0f queued (same rule), %d trips waiting to depart at the end' %
              (expiry, 'phase expires' if expiry else 'phase stays', e['mean_active'], e['queue'] * (sc.n_signals + 1), e['stats']['pending']))
        print('lane               len   mean occ standing  windows  cars/7s-green(q>=5)  first car after  cars/window(all)  what limits the car at the head of the lane, ticks (red / yellow excluded)')
        for l, ws in e['windows'].items():
            if not ws:
                continue
            w7 = [w for w in ws if w[0] == 7 and w[1] >= 5]
            firsts = [w[3] for w in w7 if w[3] is not None]
            hr = e['head_reason'][l]
            top = ', '.join('%s %d' % (REASON[i], hr[i]) for i in np.argsort(-hr)[:4] if hr[i] > 0 and i != 3)
            print('%-16s %6.1f %8.1f %8.1f %8d %12s %18s %17.2f   %s' % (
                sc.lane_ids[l], A['lane_len'][l], e['occ'][l], e['stand'][l], len(ws),
                ('%.2f (n=%d)' % (np.mean([w[2] for w in w7]), len(w7))) if w7 else '-',
                ('%.1f s' % np.mean(firsts)) if firsts else '-', np.mean([w[2] for w in ws]), top))
        wl = int(sum(e['head_reason'][l, 2] for l in e['windows']))
        foe = int(sum(e['head_reason'][l, 4] + e['head_reason'][l, 7] for l in e['windows']))
        print('lane-ticks with the head car stranded on a lane that does not continue its route: %d; waiting for a permissive gap: %d (of %d lane-ticks)' %
              (wl, foe, 3600 * len(e['windows'])))
    return


def part4():
    print('\n## 4. cologne3 under the random policy: does this model ever gridlock?  16 environments x one episode')
    rows = []
    for expiry in (0, 1):
        ms, ls, arr = [], [], []
        for envi in range(16):
            e = episode('cologne3', envi, expiry, sample_lanes=[])
            ms.append(e['max_stand'])
            ls.append(e['long_standers'])
            arr.append(e['stats']['arrived'])
        print('tls_expiry %d: longest standstill of any vehicle %d s (median over environments %d s); vehicles standing >= 300 s at the end: %d in total; arrived %d..%d of %d' %
              (expiry, max(ms), int(np.median(ms)), sum(ls), min(arr), max(arr), load('cologne3').n_trips))
    print('# no environment gridlocks.  The reference figure (279 s) is the mean over trials with a spread of +-202 s: one of its trials did.')
    print('# What decides this cell is whether a lane change is possible on the two edges between its paired junctions (200818108#0: 9.7 m,')
    print('# 319261593#16: 12.6 m).  RM_MIN_LC_LEN = 5 m (shipped): 0.35 x the reference; 12.5 m (no change on either edge): 2.9 x (coordinate search of round 3).')


if __name__ == '__main__':
    print('# tools/discharge_study.py (CPU oracle, model v5)')
    part1()
    part2()
    part3('cologne1')
    part3('cologne3')
    part4()
