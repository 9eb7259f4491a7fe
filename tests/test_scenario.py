"""CPU: scenario compiler tables vs the reference's Signal.__init__ / create_yellows (golden) and the
survey's probe numbers."""
import numpy as np
import pytest

from conftest import load_golden, load_scenario
from resco_amd.scenario import build_yellow_program, green_phases


@pytest.mark.parametrize('tag', ['cologne1_d200', 'cologne8_d200', 'ingolstadt21_d200'])
def test_signal_tables_match_reference(tag):
    meta, _ = load_golden(tag)
    sc = load_scenario(meta['map'])
    assert sc.signal_ids == meta['all_ts_ids'] == meta['ts_order']
    for i, sid in enumerate(sc.signal_ids):
        ref, mine = meta['signals'][sid], sc.signal_meta[sid]
        assert mine['lanes'] == ref['lanes']
        assert mine['outbound_lanes'] == ref['outbound_lanes']
        assert mine['out_lane_to_signalid'] == ref['out_lane_to_signalid']
        assert mine['inbounds_fr_direction'] == ref['inbounds_fr_direction']
        assert {k: sorted(v) for k, v in mine['lane_sets_outbound'].items()} == ref['lane_sets_outbound']
        assert list(mine['lane_sets'].keys()) == list(ref['lane_sets'].keys())
        assert mine['phases'] == ref['phases']                      # create_yellows output incl. durations
        assert mine['yellow_dict'] == ref['yellow_dict']
        assert mine['phases'][:mine['n_green']] == ref['green_phases']
        assert meta['n_green'][i] == mine['n_green'] == int(sc.tls_ngreen[i])
        assert meta['obs_shape'][sid] == [13]


def test_survey_sizes():
    # SURVEY.md 8 / appendix D
    want = {'cologne1': (1, 8, 8, [4], [14]), 'cologne8': (8, 33, 6, None, None), 'ingolstadt21': (21, 163, 17, None, None)}
    for name, (S, O, lmax, greens, phases) in want.items():
        sc = load_scenario(name)
        assert sc.n_signals == S and sc.n_obs == O
        assert int(np.diff(sc.sig_obs_start).max()) == lmax
        if greens:
            assert list(sc.tls_ngreen) == greens and list(sc.tls_nphase) == phases
    assert int(load_scenario('cologne8').tls_ngreen.sum()) == 25
    assert int(load_scenario('ingolstadt21').tls_ngreen.sum()) == 67
    assert load_scenario('cologne1').n_routes == 23 and load_scenario('cologne8').n_routes == 579
    assert load_scenario('ingolstadt21').n_routes == 900 and load_scenario('ingolstadt21').n_trips == 4283


def test_cologne1_yellow_dict_known_answer():
    # SURVEY.md 8(a) A5 probe
    sc = load_scenario('cologne1')
    yd = sc.signal_meta[sc.signal_ids[0]]['yellow_dict']
    assert yd == {'0_1': 4, '0_2': 5, '0_3': 6, '1_2': 7, '1_3': 8, '2_0': 9, '2_1': 10, '2_3': 11, '3_0': 12, '3_1': 13}
    assert sc.signal_meta[sc.signal_ids[0]]['green_durations'] == [29, 6, 29, 6]


def test_yellow_program_rules():
    greens = green_phases([(30, 'GGrr'), (4, 'yyrr'), (30, 'rrGG'), (4, 'rryy'), (5, 'rrrr')])
    assert greens == [(30, 'GGrr'), (30, 'rrGG')]
    phases, yd = build_yellow_program(greens, 3)
    assert phases == [(30, 'GGrr'), (30, 'rrGG'), (3, 'yyrr'), (3, 'rryy')] and yd == {'0_1': 2, '1_0': 3}
    # a pair that needs no yellow gets no entry; 'g' -> 's' counts like 'r'
    phases, yd = build_yellow_program([(10, 'Gg'), (10, 'GG'), (10, 'sG')], 2)
    assert '0_1' not in yd and '1_0' not in yd and yd['0_2'] == 3 and phases[3] == (2, 'yg')


def test_table_consistency():
    for name in ('cologne1', 'cologne8', 'ingolstadt21'):
        sc = load_scenario(name)
        A = sc.arrays
        assert (A['lane_len'] > 0).all() and (A['lane_vmax'] > 0).all()
        assert (A['link_to_lane'] >= 0).all() and (A['link_to_lane'] < sc.n_lanes).all()
        assert (A['foe_link'] >= 0).all() and (A['foe_link'] < sc.n_links).all()
        assert (np.diff(A['trip_depart']) >= 0).all() and A['trips_cum'][-1] == sc.n_trips
        assert (A['obs_lane'] < sc.n_lanes).all()
        # route continuation lengths (SUMO's bestLanes): every step has a lane with a positive length, the last step
        # continues "to the end" on all its lanes, entries beyond the edge's lane count are 0, and a lane's
        # continuation is at least its own length
        cont = A['route_cont']
        assert cont.shape == (len(A['route_edge']), sc.kmax) and sc.kmax == int(A['edge_nlanes'].max())
        nl = A['edge_nlanes'][A['route_edge']]
        assert (cont.max(axis=1) > 0).all()
        assert all((cont[q, nl[q]:] == 0).all() for q in range(len(nl)))
        assert (cont[A['route_start'][1:] - 1, 0] == np.float32(1.0e6)).all()
        assert (cont[np.arange(len(nl)), 0] >= A['lane_len'][A['edge_lane0'][A['route_edge']]] - 1e-3).all()
        # internal lanes have exactly one outgoing link
        assert (A['lane_link_cnt'][A['lane_internal'] == 1] == 1).all()
        assert sc.capacity % 64 == 0 and 64 <= sc.capacity <= 1984


# ------------------------------------------------------------------------------------------------ tables every checker shares
# The oracle, the kernel, the host emulation and FakeSumo all read the routes and detector distances resco_amd/scenario.py
# compiled: a wrong table there is invisible to the bit-exact tests.  tests/golden/route_pins_<map>.npz is an independent
# recomputation from the reference's net.xml / rou.xml (own XML walk, a label-correcting search instead of the heap Dijkstra;
# tests/golden/make_route_pins.py, build container).
def _pins(name):
    import os
    from conftest import ROOT
    z = np.load(os.path.join(ROOT, 'tests', 'golden', 'route_pins_%s.npz' % name))
    edge = {e: (float(l), float(v)) for e, l, v in zip(z['edge_ids'], z['edge_len'], z['edge_speed'])}
    hop = {(a, b): (float(t), float(lo), float(hi), bool(tl)) for a, b, t, lo, hi, tl in
           zip(z['hop_a'], z['hop_b'], z['hop_time'], z['hop_len_lo'], z['hop_len_hi'], z['hop_tls'])}
    od = {(a, b): float(c) for a, b, c in zip(z['od_from'], z['od_to'], z['od_cost'])}
    return edge, hop, od


@pytest.mark.parametrize('name', ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21'])
def test_routes_and_detector_distances_against_an_independent_recomputation(name):
    sc = load_scenario(name)
    edge, hop, od = _pins(name)
    A = sc.arrays
    n_checked = 0
    for r in range(sc.n_routes):
        es = [sc.edge_ids[e] for e in A['route_edge'][A['route_start'][r]:A['route_start'][r + 1]]]
        # every edge and every transition of the route exists in the network, with the length the lane table carries
        for e in es:
            assert abs(edge[e][0] - float(A['lane_len'][A['edge_lane0'][sc.edge_ids.index(e)]])) < 1e-3
        cost = sum(edge[e][0] / edge[e][1] for e in es) + sum(hop[(a, b)][0] for a, b in zip(es[:-1], es[1:]))
        if (es[0], es[-1]) in od:          # <trip> demand: the compiled route is a fastest path (ties may differ in the edges)
            assert abs(cost - od[(es[0], es[-1])]) <= 1e-6 * cost, (name, es[0], es[-1], cost, od[(es[0], es[-1])])
            n_checked += 1
        # route_tlsdist[q]: from the end of edge q to the next TLS stop line along the route (vehicle.getNextTLS, which
        # Signal.get_vehicles compares with max_distance, traffic_signal.py:238-247)
        lo = hi = None
        q0 = int(A['route_start'][r])
        for q in range(len(es) - 1, -1, -1):
            if q == len(es) - 1:
                lo = hi = np.inf
            else:
                t, l_lo, l_hi, tl = hop[(es[q], es[q + 1])]
                lo, hi = (0.0, 0.0) if tl else (l_lo + edge[es[q + 1]][0] + lo, l_hi + edge[es[q + 1]][0] + hi)
            mine = float(A['route_tlsdist'][q0 + q])
            if np.isinf(lo):
                assert mine >= 1e9          # no signal ahead: never detectable
            else:
                assert lo - 0.05 <= mine <= hi + 0.05, (name, es[q], mine, lo, hi)
    if sc.demand_tag == 'trip':
        assert n_checked == sc.n_routes == len(od)


def test_survey_appendix_tables():
    """SURVEY.md Appendix D (per-signal sizing, a probe over the reference's own Signal objects) and Appendix A (junction
    passages: one internal lane for straight / right, two for the left turns that wait inside the junction)"""
    D = {'cologne1': [(8, 0, 4, 14, 20, [29, 6, 29, 6])],
         'cologne8': [(6, 2, 4, 14, 18, [33, 6, 33, 6]), (4, 3, 2, 4, 16, [33, 33]), (3, 1, 3, 8, 9, [38, 6, 37]),
                      (6, 3, 4, 14, 18, [33, 6, 33, 6]), (4, 3, 3, 8, 9, [38, 6, 37]), (2, 1, 2, 3, 8, [78, 6]),
                      (4, 3, 3, 8, 9, [38, 6, 37]), (4, 1, 4, 14, 16, [33, 6, 33, 6])],
         'ingolstadt21': [(7, 5, 3, 8, 7, [35, 6, 34]), (6, 4, 3, 8, 6, [35, 6, 34]), (4, 10, 3, 8, 6, [38, 6, 37]),
                          (6, 2, 3, 8, 4, [20, 30, 26]), (8, 2, 4, 15, 12, [29, 6, 29, 6]), (7, 1, 3, 8, 14, [38, 6, 37]),
                          (5, 4, 3, 8, 15, [38, 6, 37]), (7, 8, 2, 4, 9, [42, 42]), (7, 5, 3, 8, 11, [38, 6, 37]),
                          (11, 7, 4, 15, 12, [29, 6, 29, 6]), (6, 2, 3, 8, 12, [38, 6, 37]), (11, 3, 4, 15, 10, [24, 6, 24, 24]),
                          (5, 3, 3, 8, 8, [38, 6, 37]), (17, 4, 4, 15, 15, [29, 6, 29, 6]), (12, 5, 4, 15, 12, [15, 25, 5, 36]),
                          (9, 4, 3, 8, 12, [38, 6, 37]), (7, 5, 3, 8, 8, [38, 6, 37]), (5, 4, 3, 9, 6, [35, 6, 34]),
                          (10, 3, 3, 8, 14, [38, 6, 37]), (8, 6, 3, 8, 9, [38, 6, 37]), (5, 7, 3, 8, 8, [38, 6, 37])]}
    for name, rows in D.items():
        sc = load_scenario(name)
        assert sc.n_signals == len(rows)
        for k, (lanes, outb, G, P, links, durs) in enumerate(rows):
            sm = sc.signal_meta[sc.signal_ids[k]]
            assert len(sm['lanes']) == lanes == int(sc.sig_obs_start[k + 1] - sc.sig_obs_start[k])
            assert len(sm['outbound_lanes']) == outb
            assert (int(sc.tls_ngreen[k]), int(sc.tls_nphase[k]), int(sc.tls_nlinks[k])) == (G, P, links)
            assert sm['green_durations'] == durs
    # Appendix A counts all normal -> normal connections of the net; the compiled scenario keeps those the demand uses:
    # every first-stage link has one or two junction lanes, never none and never more
    for name in D:
        A = load_scenario(name).arrays
        first = A['link_tls_pos'] >= -1
        normal_from = A['lane_internal'][A['link_from_lane']] == 0
        v1, v2 = A['link_via1'][normal_from & first], A['link_via2'][normal_from & first]
        assert (v1 >= 0).all() and ((v2 >= 0) <= (v1 >= 0)).all()


@pytest.mark.parametrize('name', ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21'])
def test_junction_tables_against_an_independent_recomputation(name):
    """tests/golden/junction_pins_<map>.json.gz (tests/golden/make_junction_pins.py, build container, own reading of net.xml):
    every link of the compiled scenario that leaves a normal lane has the traffic light, link index, minor / cont flags and the
    prohibitors (<request response>) of its <connection>; route_cont equals the lane-level recursion over the connections."""
    import gzip
    import json
    import os
    from conftest import ROOT
    with gzip.open(os.path.join(ROOT, 'tests', 'golden', 'junction_pins_%s.json.gz' % name), 'rt') as f:
        pins = json.load(f)
    sc = load_scenario(name)
    A = sc.arrays
    lid = sc.lane_ids

    def key(k):
        return '%s>%s' % (lid[int(A['link_from_lane'][k])], lid[int(A['link_dest_lane'][k])])

    n_first = 0
    for k in range(sc.n_links):
        if A['lane_internal'][A['link_from_lane'][k]]:
            continue
        n_first += 1
        ref = pins['links'][key(k)]
        tls = int(A['link_tls'][k])
        assert (sc.signal_ids[tls] if tls >= 0 else None) == (ref['tl'] if ref['tl'] in sc.signal_ids else None)
        if tls >= 0:
            assert int(A['link_tls_pos'][k]) == ref['idx']
        assert int(A['link_minor'][k]) == ref['minor'] and int(A['link_cont'][k]) == ref['cont']
        assert (int(A['link_via1'][k]) >= 0) + (int(A['link_via2'][k]) >= 0) == min(ref['n_via'], 2)
        mine = sorted(key(int(f)) for f in A['foe_link'][A['link_foe_start'][k]:A['link_foe_start'][k] + A['link_foe_cnt'][k]])
        # the scenario keeps only the connections the demand uses: prohibitors nobody ever drives are dropped
        assert mine == [f for f in ref['foes'] if f in used_keys(sc)], (key(k), mine, ref['foes'])
    assert n_first > 0
    # continuation lengths
    cont = A['route_cont']
    for r in range(sc.n_routes):
        q0 = int(A['route_start'][r])
        for c, row in enumerate(pins['cont'][r]):
            e = int(A['route_edge'][q0 + c])
            for kk in range(int(A['edge_nlanes'][e])):
                want = row[lid[int(A['edge_lane0'][e]) + kk]]
                got = float(cont[q0 + c, kk])
                assert abs(min(got, 1e6) - min(want, 1e6)) <= 1e-3 * max(1.0, min(want, 1e6)), (name, r, c, kk, got, want)


_USED = {}


def used_keys(sc):
    if sc.name not in _USED:
        A = sc.arrays
        _USED[sc.name] = {'%s>%s' % (sc.lane_ids[int(A['link_from_lane'][k])], sc.lane_ids[int(A['link_dest_lane'][k])])
                          for k in range(sc.n_links) if not A['lane_internal'][A['link_from_lane'][k]]}
    return _USED[sc.name]
