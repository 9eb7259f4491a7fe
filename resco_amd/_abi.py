"""ctypes mirror of the C-ABI structs in include/resco_sim.h (rs_scenario / rs_params)."""
import ctypes as C

import numpy as np

_I32P = C.POINTER(C.c_int32)
_U32P = C.POINTER(C.c_uint32)
_F32P = C.POINTER(C.c_float)

_COUNTS = ['n_lanes', 'n_links', 'n_edges', 'n_routes', 'n_trips', 'n_signals', 'n_obs', 'n_vtypes',
           'n_foes', 'n_route_steps', 'n_tls_states', 'n_tls_dur', 'n_tls_yellow',
           'n_fix_states', 'n_fix_dur', 'n_mv_in', 'n_mv_out', 'n_pr_out',
           'horizon', 'capacity', 'step_length', 'yellow_length', 'kmax']

# (field, ctype) in the exact order of the C struct
_POINTERS = [
    ('lane_len', _F32P), ('lane_vmax', _F32P),
    ('lane_edge', _I32P), ('lane_left', _I32P), ('lane_right', _I32P), ('lane_link_start', _I32P),
    ('lane_link_cnt', _I32P), ('lane_obs', _I32P), ('lane_internal', _I32P),
    ('link_to_lane', _I32P), ('link_dest_lane', _I32P), ('link_to_edge', _I32P), ('link_tls', _I32P),
    ('link_tls_pos', _I32P), ('link_minor', _I32P), ('link_cont', _I32P),
    ('link_foe_start', _I32P), ('link_foe_cnt', _I32P),
    ('link_via_len', _F32P),
    ('link_via1', _I32P), ('link_via2', _I32P), ('link_from_lane', _I32P), ('foe_link', _I32P),
    ('edge_lane0', _I32P), ('edge_nlanes', _I32P),
    ('route_start', _I32P), ('route_edge', _I32P),
    ('route_tlsdist', _F32P),
    ('route_cont', _F32P),
    ('trip_depart', _I32P), ('trip_route', _I32P), ('trip_vtype', _I32P), ('trips_cum', _I32P),
    ('vtype_params', _F32P),
    ('tls_nphase', _I32P), ('tls_ngreen', _I32P), ('tls_nlinks', _I32P), ('tls_state_off', _I32P),
    ('tls_dur_off', _I32P), ('tls_yel_off', _I32P), ('tls_init_phase', _I32P),
    ('tls_states', _I32P), ('tls_dur', _I32P), ('tls_yellow', _I32P),
    ('fix_nphase', _I32P), ('fix_state_off', _I32P), ('fix_dur_off', _I32P), ('fix_init_phase', _I32P),
    ('fix_init_left', _I32P), ('fix_states', _I32P), ('fix_dur', _I32P),
    ('obs_lane', _I32P), ('sig_obs_start', _I32P), ('mv_in_start', _I32P), ('mv_in_idx', _I32P),
    ('mv_out_start', _I32P), ('mv_out_idx', _I32P), ('pr_out_start', _I32P), ('pr_out_idx', _I32P),
]


class ScenarioStruct(C.Structure):
    _fields_ = [(n, C.c_int32) for n in _COUNTS] + _POINTERS


class ParamsStruct(C.Structure):
    _fields_ = [('seed', C.c_uint32), ('max_distance', C.c_float), ('sigma', C.c_float),
                ('speed_dev', C.c_int32), ('fixed_program', C.c_int32), ('trip_log', C.c_int32), ('step_ratio', C.c_int32),
                ('tls_hold', C.c_int32)]


def pack_scenario(sc, step_length=10, yellow_length=None):
    """Returns (ScenarioStruct, keepalive list).  The numpy arrays must outlive the struct."""
    A = sc.arrays
    st = ScenarioStruct()
    keep = []
    counts = dict(
        n_lanes=sc.n_lanes, n_links=sc.n_links, n_edges=sc.n_edges, n_routes=sc.n_routes, n_trips=sc.n_trips,
        n_signals=sc.n_signals, n_obs=sc.n_obs, n_vtypes=A['vtype_params'].shape[0],
        n_foes=len(A['foe_link']), n_route_steps=len(A['route_edge']), n_tls_states=len(A['tls_states']),
        n_tls_dur=len(A['tls_dur']), n_tls_yellow=len(A['tls_yellow']), n_fix_states=len(A['fix_states']),
        n_fix_dur=len(A['fix_dur']), n_mv_in=len(A['mv_in_idx']), n_mv_out=len(A['mv_out_idx']),
        n_pr_out=len(A['pr_out_idx']), horizon=sc.horizon, capacity=sc.capacity, step_length=step_length,
        yellow_length=sc.yellow_length if yellow_length is None else yellow_length, kmax=sc.kmax)
    for k, v in counts.items():
        setattr(st, k, int(v))
    for name, ptype in _POINTERS:
        want = {_I32P: np.int32, _U32P: np.uint32, _F32P: np.float32}[ptype]
        a = A[name]
        if want is np.uint32 and a.dtype == np.int32:
            a = a.view(np.uint32)
        a = np.ascontiguousarray(a, dtype=want)
        keep.append(a)
        setattr(st, name, a.ctypes.data_as(ptype))
    return st, keep
