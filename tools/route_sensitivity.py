#!/usr/bin/env python3
"""Route-assignment sensitivity of ingolstadt21 FIXED (build container only: reads the reference's net.xml / rou.xml).

The shipped scenario routes every <trip> over edge length / speed + junction-lane time (resco_amd/scenario.py).  SUMO's router
adds terms this cost model leaves out [SUMO-K, none pinned]: `--weights.minor-penalty` (1.5 s per junction lane entered over a link
that is neither traffic-light controlled nor has priority) and, in later versions, `--weights.turnaround-penalty`.  This tool

  1. re-routes the 900 OD pairs with those terms and reports which routes (and how many trips) move,
  2. lists every OD pair whose best and second-best route (the best route that avoids at least one edge of the best) are within 2 %,
  3. says how many of the trips of the E approach of TLS 243641585 (`-201201945#0.78` -> `-174800513`, 40 % of the FIXED delay,
     profiles/r04_ingolstadt21_approaches.txt) have such an alternative or are re-routed,
  4. runs the FIXED programme (CPU oracle, test infrastructure, 8 environments x one episode) on every re-routed assignment.

  python tools/route_sensitivity.py > profiles/r05_route_sensitivity.txt
"""
import heapq
import importlib.util
import multiprocessing as mp
import os
import sys
import tempfile

import numpy as np

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = '/root/reference'
MAP = 'ingolstadt21'
E_EDGE, E_NEXT = '-201201945#0.78', '-174800513'


def load_ref(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def dijkstra(succ, cost, src, dst, banned=None):
    dist = {src: cost[src]}
    prev = {}
    heap = [(cost[src], src)]
    done = set()
    while heap:
        d, u = heapq.heappop(heap)
        if u in done:
            continue
        done.add(u)
        if u == dst:
            break
        for v, w in succ.get(u, ()):
            if banned is not None and v == banned:
                continue
            nd = d + w + cost[v]
            if v not in dist or nd < dist[v]:
                dist[v] = nd
                prev[v] = u
                heapq.heappush(heap, (nd, v))
    if dst not in done:
        return None, float('inf')
    path = [dst]
    while path[-1] != src:
        path.append(prev[path[-1]])
    return path[::-1], dist[dst]


def fixed_delay(job):
    path, env_index = job
    from oracle.pyoracle import OracleEnv
    from resco_amd.scenario import Scenario
    import oracle.fidelity_eval as F
    _orig = Scenario.load
    Scenario.load = staticmethod(lambda p: _orig(path))
    try:
        return F.episode((MAP, 'FIXED', env_index, 0, 360))
    finally:
        Scenario.load = staticmethod(_orig)


def main():
    from resco_amd.scenario import compile_from_sumocfg
    scm = load_ref('resco_benchmark/config/signal_config.py', '_rs_sig')
    mcm = load_ref('resco_benchmark/config/map_config.py', '_rs_map')
    mc = mcm.map_configs[MAP]
    cfg = os.path.join(REF, 'resco_benchmark', mc['net'])
    variants = [('shipped cost model', 0.0, 0.0), ('minor-penalty 1.5 s', 1.5, 0.0), ('turnaround-penalty 5 s', 0.0, 5.0),
                ('minor 1.5 s + turnaround 5 s', 1.5, 5.0), ('minor-penalty 3 s', 3.0, 0.0), ('minor 5 s + turnaround 10 s', 5.0, 10.0)]
    compiled = {}
    for label, mp_, tp in variants:
        compiled[label] = compile_from_sumocfg(MAP, cfg, scm.signal_configs[MAP], lights=mc['lights'],
                                               yellow_length=mc['yellow_length'], minor_penalty=mp_, turnaround_penalty=tp)
    base = compiled['shipped cost model']
    od = base.router['od_route']
    # trips per OD
    from resco_amd.scenario import parse_routes, parse_sumocfg
    _, rou, begin, _ = parse_sumocfg(cfg)
    _, trips = parse_routes(rou)
    per_od = {}
    for (tid, vt, depart, frm, to, explicit) in trips:
        if depart >= begin:
            per_od[(frm, to)] = per_od.get((frm, to), 0) + 1
    n_trips = sum(per_od.values())
    e_ods = {k for k, r in od.items() if r and any(a == E_EDGE and b == E_NEXT for a, b in zip(r[:-1], r[1:]))}
    e_trips = sum(per_od[k] for k in e_ods)
    print('# %s: %d trips, %d OD pairs; E approach of 243641585 (%s -> %s): %d OD pairs, %d trips' %
          (MAP, n_trips, len(od), E_EDGE, E_NEXT, len(e_ods), e_trips))
    print('\n## 1. re-routing with router cost terms the shipped compiler omits')
    print('%-32s %10s %10s %14s %14s' % ('cost model', 'ODs moved', 'trips', 'E-approach ODs', 'E trips moved'))
    for label, _, _ in variants[1:]:
        o2 = compiled[label].router['od_route']
        moved = [k for k in od if od[k] != o2.get(k)]
        print('%-32s %10d %10d %14d %14d' % (label, len(moved), sum(per_od[k] for k in moved), len([k for k in moved if k in e_ods]),
                                           sum(per_od[k] for k in moved if k in e_ods)))
        still = sum(per_od[k] for k, r in o2.items() if r and any(a == E_EDGE and b == E_NEXT for a, b in zip(r[:-1], r[1:])))
        print('%-32s   trips over the E approach after re-routing: %d' % ('', still))

    print('\n## 2. near ties under the shipped cost model: second-best = the best route that avoids one edge of the best')
    cost, succ = base.router['cost'], base.router['succ']
    near = []
    for k, r in od.items():
        if not r:
            continue
        _, c0 = dijkstra(succ, cost, k[0], k[1])
        best2, alt = float('inf'), None
        for e in r[1:-1]:
            p, c = dijkstra(succ, cost, k[0], k[1], banned=e)
            if p is not None and c < best2:
                best2, alt = c, p
        if alt is not None and (best2 - c0) / c0 < 0.02:
            near.append((k, c0, best2, alt))
    print('OD pairs with an alternative within 2 %%: %d of %d (%d trips of %d)' % (len(near), len(od), sum(per_od[k] for k, *_ in near), n_trips))
    e_near = [x for x in near if x[0] in e_ods]
    print('... of them over the E approach of 243641585: %d OD pairs, %d of the %d trips' % (len(e_near), sum(per_od[x[0]] for x in e_near), e_trips))
    for k, c0, c2, alt in sorted(near, key=lambda x: -per_od[x[0]])[:25]:
        on_e = any(a == E_EDGE and b == E_NEXT for a, b in zip(alt[:-1], alt[1:]))
        print('  %-22s -> %-22s trips %4d  best %7.2f s  second %7.2f s (+%.2f %%)%s%s' %
              (k[0], k[1], per_od[k], c0, c2, 100 * (c2 - c0) / c0, '  [E approach]' if k in e_ods else '',
               '  [alternative joins the E approach]' if on_e and k not in e_ods else ''))
    # the E-approach ODs: how far is their best alternative that AVOIDS the E movement?
    print('\nE-approach OD pairs: cost of the best route that avoids %s -> %s' % (E_EDGE, E_NEXT))
    succ_noe = {a: [(b, w) for b, w in lst if not (a == E_EDGE and b == E_NEXT)] for a, lst in succ.items()}
    hist = []
    for k in sorted(e_ods, key=lambda k: -per_od[k]):
        _, c0 = dijkstra(succ, cost, k[0], k[1])
        _, c1 = dijkstra(succ_noe, cost, k[0], k[1])
        hist.append((per_od[k], (c1 - c0) / c0 if c1 < float('inf') else float('inf')))
    for thr in (0.02, 0.05, 0.10, 0.25):
        print('  detour <= %3.0f %%: %4d of %d trips' % (100 * thr, sum(n for n, d in hist if d <= thr), e_trips))
    print('  no route without it: %d trips' % sum(n for n, d in hist if d == float('inf')))

    print('\n## 3. FIXED programme on every assignment (CPU oracle, 8 environments, median; reference 130.37 s)')
    tmp = tempfile.mkdtemp()
    with mp.get_context('fork').Pool(8) as pool:
        for label, _, _ in variants:
            p = os.path.join(tmp, label.replace(' ', '_').replace('.', '_') + '.npz')
            compiled[label].save(p)
            rows = pool.map(fixed_delay, [(p, e) for e in range(8)])
            d = np.median([r['delay'] for r in rows])
            print('%-32s routes %4d  delay %6.1f s  (%.2f x)' % (label, compiled[label].n_routes, d, d / 130.37))


if __name__ == '__main__':
    main()
