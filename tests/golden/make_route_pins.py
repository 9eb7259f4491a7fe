#!/usr/bin/env python3
"""An INDEPENDENT cross-check of the tables the oracle, the kernel and the host emulation all share (BUILD CONTAINER ONLY).

resco_amd/scenario.py compiles routes (own Dijkstra over net.xml; cost = edge length / speed + junction-lane time + SUMO's
minor-link penalty of 1.5 s [SUMO-K]) and route_tlsdist (the distance from the end of a route
edge to the next TLS stop line, what vehicle.getNextTLS feeds Signal.get_vehicles, traffic_signal.py:238-247).  A wrong
table there is invisible to every bit-exact test because all three implementations read the same one.  This script recomputes
both from the reference's net.xml / rou.xml with its own code -- a separate XML walk, a label-correcting (Bellman-Ford /
SPFA) search instead of the heap Dijkstra, a backward recursion for the distances -- and stores per map:
    od_cost      [n_od] the optimal free-flow travel time of every origin / destination pair of the demand
    od_from/to   the edge ids of the pairs
    tls_lo/hi    per OD pair and route step the smallest / largest distance from the end of that edge to the next TLS stop
                 line over the parallel lane-level connections (they differ by a few metres of junction geometry)
in tests/golden/route_pins_<map>.npz.  tests/test_scenario.py compares the shipped scenarios with them.

  python tests/golden/make_route_pins.py
"""
import collections
import os
import xml.etree.ElementTree as ET

import numpy as np

REF = '/root/reference/resco_benchmark/environments'
HERE = os.path.dirname(os.path.abspath(__file__))
MAPS = ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
NO_CAR = {'pedestrian', 'bicycle', 'tram', 'rail_urban', 'rail', 'rail_electric', 'rail_fast', 'ship'}


def car_allowed(lane):
    allow, dis = lane.get('allow'), lane.get('disallow')
    if allow is not None:
        return 'passenger' in allow.split()
    if dis is not None:
        return 'passenger' not in dis.split()
    return True


def read_net(path):
    edge_len, edge_speed, lane_len, lane_speed, lane_ok, internal = {}, {}, {}, {}, {}, set()
    conns = []
    for el in ET.parse(path).getroot():
        if el.tag == 'edge':
            lanes = el.findall('lane')
            for ln in lanes:
                lane_len[ln.get('id')] = float(ln.get('length'))
                lane_speed[ln.get('id')] = float(ln.get('speed'))
                lane_ok[ln.get('id')] = car_allowed(ln)
            if el.get('function') == 'internal':
                internal.add(el.get('id'))
                continue
            ok = [ln for ln in lanes if car_allowed(ln)]
            if ok:
                edge_len[el.get('id')] = float(lanes[0].get('length'))
                edge_speed[el.get('id')] = max(float(ln.get('speed')) for ln in ok)
        elif el.tag == 'connection':
            conns.append(el.attrib)
    return edge_len, edge_speed, lane_len, lane_speed, lane_ok, internal, conns


MINOR_PENALTY = 1.5     # [SUMO-K] --weights.minor-penalty: seconds per junction lane entered over an uncontrolled link without priority


def via_chain(c, conns_from):
    """the internal lanes of a connection, in order (a left turn: two), and the router's minor-link penalty over them"""
    out, via, guard, pen = [], c.get('via'), 0, 0.0
    cur = c
    while via and guard < 4:
        out.append(via)
        if 'tl' not in cur and not cur.get('state', 'M')[:1].isupper():
            pen += MINOR_PENALTY
        e, i = via.rsplit('_', 1)
        nxt = conns_from.get((e, i))
        cur = nxt[0] if nxt else None
        via = cur.get('via') if cur is not None else None
        guard += 1
    return out, pen


def main():
    for m in MAPS:
        edge_len, edge_speed, lane_len, lane_speed, lane_ok, internal, conns = read_net(os.path.join(REF, m, m + '.net.xml'))
        conns_from = collections.defaultdict(list)
        for c in conns:
            conns_from[(c['from'], c['fromLane'])].append(c)
        # edge graph: cost of entering b from a = cheapest junction passage + the travel time of b
        hop = {}                    # (a, b) -> list of (via time, via length, controlled by a TLS)
        for c in conns:
            a, b = c['from'], c['to']
            if a in internal or b in internal or a not in edge_len or b not in edge_len:
                continue
            if not lane_ok.get('%s_%s' % (a, c['fromLane'])) or not lane_ok.get('%s_%s' % (b, c['toLane'])):
                continue
            ch, pen = via_chain(c, conns_from)
            hop.setdefault((a, b), []).append((sum(lane_len[v] / max(lane_speed[v], 0.1) for v in ch) + pen,
                                               sum(lane_len[v] for v in ch), 'tl' in c))
        succ = collections.defaultdict(list)
        for (a, b), alts in hop.items():
            succ[a].append((b, min(t for t, _, _ in alts)))
        # demand
        ods = []
        for el in ET.parse(os.path.join(REF, m, m + '.rou.xml')).getroot():
            if el.tag == 'trip':
                ods.append((el.get('from'), el.get('to')))
            elif el.tag == 'vehicle' and float(el.get('depart')) >= 0:        # explicit routes (cologne3): nothing to search
                pass
        ods = sorted(set(ods))
        cost_from = {}
        for src in sorted(set(o for o, _ in ods)):
            # label-correcting search (queue based Bellman-Ford): no heap, no settled set
            dist = {src: edge_len[src] / edge_speed[src]}
            queue = collections.deque([src])
            while queue:
                u = queue.popleft()
                du = dist[u]
                for v, w in succ.get(u, ()):
                    nd = du + w + edge_len[v] / edge_speed[v]
                    if nd < dist.get(v, float('inf')) - 1e-12:
                        dist[v] = nd
                        queue.append(v)
            cost_from[src] = dist
        od_cost = np.array([cost_from[o].get(d, np.inf) for o, d in ods])
        np.savez_compressed(os.path.join(HERE, 'route_pins_%s.npz' % m), od_from=np.array([o for o, _ in ods]),
                            od_to=np.array([d for _, d in ods]), od_cost=od_cost,
                            hop_a=np.array([a for a, _ in hop]), hop_b=np.array([b for _, b in hop]),
                            hop_len_lo=np.array([min(x[1] for x in v) for v in hop.values()]),
                            hop_len_hi=np.array([max(x[1] for x in v) for v in hop.values()]),
                            hop_tls=np.array([any(x[2] for x in v) for v in hop.values()]),
                            hop_time=np.array([min(x[0] for x in v) for v in hop.values()]),
                            edge_ids=np.array(sorted(edge_len)), edge_len=np.array([edge_len[e] for e in sorted(edge_len)]),
                            edge_speed=np.array([edge_speed[e] for e in sorted(edge_len)]))
        print(m, 'OD pairs', len(ods), 'unreachable', int(np.isinf(od_cost).sum()), 'edge pairs', len(hop))


if __name__ == '__main__':
    main()
