#!/usr/bin/env python3
"""Junction conflict lists for the junction study (TEST INFRASTRUCTURE, build container only: reads the reference's net.xml).

For every first-stage link (normal lane -> junction) of a compiled scenario: the first-stage links whose paths CONFLICT with it
inside the junction -- the `foes` bit string of its <request> (crossing and merging movements, whatever their priority), not only the
prohibitors of `response` the shipped model uses.  Written to oracle/study/_conflicts_<map>.npz (git-ignored); the study build of the
oracle (-DRM_STUDY_JUNCTION) reads it through orc_study_set_conflicts.

  python oracle/study/conflicts.py [map ...]
"""
import os, sys
import numpy as np
ROOT = os.path.dirname(os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario, parse_net
REF = '/root/reference/resco_benchmark/environments'


def build(name):
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz')); A = sc.arrays
    net = parse_net(os.path.join(REF, name, name + '.net.xml'))
    lane_idx = {l: i for i, l in enumerate(sc.lane_ids)}
    # first-stage links by (from lane id, first via lane id)
    link_by_via = {}
    for k in range(sc.n_links):
        fl = int(A['link_from_lane'][k])
        if A['lane_internal'][fl]:
            continue
        v1 = int(A['link_via1'][k])
        if v1 >= 0:
            link_by_via.setdefault(sc.lane_ids[v1], []).append(k)
    via_junction = {}
    for j in net.junctions.values():
        if j.type == 'internal':
            continue
        for i, l in enumerate(j.int_lanes):
            via_junction[l] = (j, i)
    start, cnt, lst = np.zeros(sc.n_links, np.int32), np.zeros(sc.n_links, np.int32), []
    for via, links in link_by_via.items():
        if via not in via_junction:
            continue
        j, ridx = via_junction[via]
        if ridx >= len(j.requests):
            continue
        foes = j.requests[ridx][1]
        n = len(foes)
        cf = []
        for b in range(n):
            if foes[n - 1 - b] == '1' and b != ridx and b < len(j.int_lanes):
                cf.extend(link_by_via.get(j.int_lanes[b], ()))
        for k in links:
            start[k] = len(lst); cnt[k] = len(cf); lst.extend(cf)
    out = os.path.join(ROOT, 'oracle', 'study', '_conflicts_%s.npz' % name)
    np.savez(out, start=start, cnt=cnt, links=np.asarray(lst, np.int32))
    print(name, 'first-stage links', sum(len(v) for v in link_by_via.values()), 'with conflicts', int((cnt > 0).sum()), 'entries', len(lst))


if __name__ == '__main__':
    for m in (sys.argv[1:] or ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']):
        build(m)
