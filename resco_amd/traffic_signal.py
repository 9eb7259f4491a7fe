"""Signal: the per-intersection object state/reward plugins receive (dict[id -> Signal]).

Mirrors the attribute surface of the reference's resco_benchmark/traffic_signal.py:28-104 (lanes,
lane_sets, lane_sets_outbound, downstream, inbounds_fr_direction, outbound_lanes, out_lane_to_signalid,
phases, yellow_dict, phase, next_phase, signals, full_observation, waiting_times, last_step_vehicles) but is
a *view*: the phase FSM and the detector run inside the HIP step kernel; this object only decodes the
device buffers of one environment on demand.
"""
import numpy as np

from .scenario import build_yellow_program


class Phase:
    """traci.trafficlight.Phase stand-in (duration, state)."""

    def __init__(self, duration, state, minDur=-1, maxDur=-1):
        self.duration, self.state, self.minDur, self.maxDur = duration, state, minDur, maxDur

    def __repr__(self):
        return 'Phase(duration=%s, state=%r)' % (self.duration, self.state)


def create_yellows(phases, yellow_length):
    """Same contract as traffic_signal.py:7-24: (green Phase list) -> (greens + yellows, yellow_dict)."""
    prog, ydict = build_yellow_program([(p.duration, p.state) for p in phases], yellow_length)
    new = list(phases) + [Phase(d, s) for d, s in prog[len(phases):]]
    return new, ydict


class Signal:
    def __init__(self, env, sig_index, sig_id):
        sc = env.scenario
        meta = sc.signal_meta[sig_id]
        self._env = env
        self._index = sig_index
        self.id = sig_id
        self.yellow_time = env.yellow_length
        self.lanes = list(meta['lanes'])
        self.lane_sets = meta['lane_sets']
        self.lane_sets_outbound = meta['lane_sets_outbound']
        self.downstream = meta['downstream']
        self.inbounds_fr_direction = meta['inbounds_fr_direction']
        self.outbound_lanes = list(meta['outbound_lanes'])
        self.out_lane_to_signalid = meta['out_lane_to_signalid']
        self.phases = [Phase(d, s) for d, s in meta['phases']]
        self.yellow_dict = dict(meta['yellow_dict'])
        self.signals = None
        self.last_step_vehicles = None
        self._obs_cache = None
        self._obs_version = -1

    # -- FSM state lives on the device
    @property
    def phase(self):
        return int(self._env._host('phase')[self._env.view_env, self._index])

    @property
    def next_phase(self):
        return int(self._env._host('tls')[self._env.view_env, self._index, 2])

    @property
    def waiting_times(self):
        """vehicle id -> RESCO waiting time (traffic_signal.py:89,198-202) decoded from the device arrays."""
        return {v['id']: v['wait'] for lane in self.lanes for v in self.full_observation[lane]['vehicles']
                if v['wait'] > 0}

    @property
    def full_observation(self):
        env = self._env
        if self._obs_version != env._version:
            self._obs_cache = self._decode()
            self._obs_version = env._version
        return self._obs_cache

    def observe(self, step_length=None, distance=None):
        """The detector already ran on the device during reset()/step(); kept for API compatibility."""
        return None

    def _decode(self):
        env, sc = self._env, self._env.scenario
        e = env.view_env
        agg = env._host('lane_agg')[e]
        o0 = int(sc.sig_obs_start[self._index])
        vl = env._host('veh_lane')[e]
        owner = env._host('veh_owner')[e]
        trip = env._host('veh_trip')[e]
        pos, spd = env._host('veh_pos')[e], env._host('veh_speed')[e]
        acc, rw = env._host('veh_accel')[e], env._host('veh_rwait')[e]
        full = {}
        allv = set()
        mine = np.nonzero((owner == self._index) & (vl < 0xFFFE))[0]
        for j, lane in enumerate(self.lanes):
            row = agg[o0 + j]
            cl = int(sc.obs_lane[o0 + j])
            slots = [s for s in mine if vl[s] == cl]
            slots.sort(key=lambda s: (pos[s], -int(trip[s])))
            vehicles = []
            for s in slots:
                k = int(trip[s])
                vid = sc.trip_ids[k]
                allv.add(vid)
                # TraCI returns waiting times as floats; vehicles outside waiting_times report the int 0
                vehicles.append({'id': vid, 'wait': float(rw[s]) if rw[s] > 0 else 0, 'speed': float(spd[s]),
                                 'acceleration': float(acc[s]), 'position': float(pos[s]),
                                 'type': sc.vtype_ids[int(sc.trip_vtype[k])]})
            full[lane] = {'queue': int(row[0]), 'approach': int(row[1]),
                          'total_wait': float(row[2]) if row[2] > 0 else 0,
                          'max_wait': float(row[3]) if row[3] > 0 else 0, 'vehicles': vehicles}
        full['num_vehicles'] = allv
        if self.last_step_vehicles is None:
            full['arrivals'] = allv
            full['departures'] = set()
        else:
            full['arrivals'] = allv.difference(self.last_step_vehicles)
            full['departures'] = self.last_step_vehicles.difference(allv)
        self.last_step_vehicles = allv
        return full
