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
        assert sc.capacity & (sc.capacity - 1) == 0
