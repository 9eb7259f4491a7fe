#!/usr/bin/env python3
"""Second half of the independent cross-check of the shared tables (BUILD CONTAINER ONLY; see make_route_pins.py): the
lane-level tables of the junction model, recomputed from the reference's net.xml with this script's own code:
    link_*    per normal -> normal connection the demand uses ("<from lane id>><to lane id>"): traffic light and link index,
              minor (state m / =), waits inside the junction (request cont), and its prohibitors = the connections whose bit is
              set in its <request response="..."> (SUMO's right-of-way matrix)
    cont_*    per route step and lane of the compiled routes: how far the route can be followed from that lane without a lane
              change (SUMO's bestLanes length [SUMO-K]); the ROUTES are read from the shipped scenario (they are pinned as
              fastest paths by route_pins_*.npz), everything else from net.xml
-> tests/golden/junction_pins_<map>.json.gz ; tests/test_scenario.py compares the shipped scenarios with it.

  python tests/golden/make_junction_pins.py
"""
import collections
import gzip
import json
import os
import sys
import xml.etree.ElementTree as ET

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference/resco_benchmark/environments'
MAPS = ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
BIG = 1.0e6


def main():
    from resco_amd.scenario import Scenario          # only to read the ROUTES (edge id sequences) of the shipped scenario
    for m in MAPS:
        root = ET.parse(os.path.join(REF, m, m + '.net.xml')).getroot()
        lane_len, internal_edges, edge_lanes = {}, set(), {}
        for e in root.iter('edge'):
            if e.get('function') == 'internal':
                internal_edges.add(e.get('id'))
            edge_lanes[e.get('id')] = [ln.get('id') for ln in e.findall('lane')]
            for ln in e.findall('lane'):
                lane_len[ln.get('id')] = float(ln.get('length'))
        conns = [c.attrib for c in root.iter('connection')]
        by_from = collections.defaultdict(list)
        for c in conns:
            by_from['%s_%s' % (c['from'], c['fromLane'])].append(c)

        def chain(c):
            out, via = [], c.get('via')
            while via and len(out) < 4:
                out.append(via)
                nxt = by_from.get(via)
                via = nxt[0].get('via') if nxt else None
            return out

        # right of way: junction -> list of (response string, cont) per request index; internal lane -> (junction, index)
        req, where = {}, {}
        for j in root.iter('junction'):
            if j.get('type') == 'internal':         # (the waiting position inside a junction: its intLanes are its FOES' lanes)
                continue
            rows = [(r.get('response'), int(r.get('cont', '0'))) for r in j.findall('request')]
            if rows:
                req[j.get('id')] = rows
            for i, il in enumerate((j.get('intLanes') or '').split()):
                where[il] = (j.get('id'), i)
        first = {}          # key -> record of a normal -> normal connection
        by_req = collections.defaultdict(list)
        for c in conns:
            if c['from'] in internal_edges or c['to'] in internal_edges:
                continue
            key = '%s_%s>%s_%s' % (c['from'], c['fromLane'], c['to'], c['toLane'])
            ch = chain(c)
            jid, ridx = None, -1
            for v in reversed(ch):
                if v in where:
                    jid, ridx = where[v]
                    break
            first[key] = dict(tl=c.get('tl'), idx=int(c['linkIndex']) if 'tl' in c else -1, minor=int(c.get('state', 'M') in 'm='),
                              via=ch, jid=jid, ridx=ridx)
            if jid is not None:
                by_req[(jid, ridx)].append(key)
        for key, r in first.items():
            foes, cont = [], 0
            if r['jid'] in req and 0 <= r['ridx'] < len(req[r['jid']]):
                resp, cont = req[r['jid']][r['ridx']]
                n = len(resp)
                for b in range(n):
                    if resp[n - 1 - b] == '1' and b != r['ridx']:
                        foes += by_req.get((r['jid'], b), [])
            r['foes'] = sorted(set(foes))
            r['cont'] = cont if len(r['via']) >= 2 else 0
        # continuation lengths along the shipped routes
        sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', m + '.npz'))
        A = sc.arrays
        cont_rows = []
        for r in range(sc.n_routes):
            es = [sc.edge_ids[e] for e in A['route_edge'][A['route_start'][r]:A['route_start'][r + 1]]]
            nxt = None
            rows = []
            for c in range(len(es) - 1, -1, -1):
                cur = {}
                for lid in edge_lanes[es[c]]:
                    if c == len(es) - 1:
                        cur[lid] = BIG
                        continue
                    best = 0.0
                    for cn in by_from.get(lid, ()):
                        if cn['to'] != es[c + 1]:
                            continue
                        to = '%s_%s' % (cn['to'], cn['toLane'])
                        v = sum(lane_len[x] for x in chain(cn)) + nxt.get(to, 0.0)
                        best = max(best, v)
                    cur[lid] = min(BIG, lane_len[lid] + best)
                rows.append(cur)
                nxt = cur
            cont_rows.append(rows[::-1])
        used = {}
        for key, r in first.items():
            used[key] = dict(tl=r['tl'], idx=r['idx'], minor=r['minor'], cont=r['cont'], foes=r['foes'], n_via=len(r['via']))
        with gzip.open(os.path.join(HERE, 'junction_pins_%s.json.gz' % m), 'wt') as f:
            json.dump(dict(links=used, cont=cont_rows), f, separators=(',', ':'))
        print(m, 'connections', len(used), 'routes', len(cont_rows))


if __name__ == '__main__':
    main()
