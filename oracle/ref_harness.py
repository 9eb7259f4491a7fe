"""Reference harness (BUILD CONTAINER ONLY; test infrastructure).

Imports the reference's own Python -- MultiSignal, Signal, create_yellows, states, rewards, the static
agents -- unmodified from /root/reference, with `traci`, `sumolib` and `gym` replaced by stub modules, and
drives it over a FakeSumo whose simulationStep()/getters are backed by the CPU oracle
(oracle/resco_oracle.c).  That pins everything on the hot path except SUMO's own dynamics:
lane ordering, yellow tables, prep->yellow->set->observe ordering, the RESCO waiting-time rule,
every state / reward formula, metrics, `done`, and the static agents' argmax.

Outputs are written as fixtures (tests/golden/*.npz, *.json) by tests/golden/make_golden.py.
Nothing from /root/reference is copied; it does not exist on the GPU box, and neither this module nor the
fixtures generator run there.
"""
import os
import sys
import types

import numpy as np

REF = os.environ.get('RESCO_REFERENCE', '/root/reference')
BIG = 1.0e29


class _Phase:
    def __init__(self, duration, state, minDur=-1, maxDur=-1, next=(), name=''):
        self.duration, self.state, self.minDur, self.maxDur = duration, state, minDur, maxDur


class _Logic:
    def __init__(self, phases, index):
        self.phases, self.type, self.currentPhaseIndex = list(phases), 0, index

    def getPhases(self):
        return self.phases


class _NS:
    pass


class FakeSumo:
    """Duck-typed TraCI connection (the subset SURVEY.md 8(b) lists) over one OracleEnv."""

    def __init__(self, scenario, orc):
        self.sc, self.orc = scenario, orc
        self.sig_index = {sid: i for i, sid in enumerate(scenario.signal_ids)}
        self.lane_index = {lid: i for i, lid in enumerate(scenario.lane_ids)}
        self._snap_t = None
        self.installed = {}
        s = self
        self.trafficlight, self.lane, self.vehicle, self.simulation = _NS(), _NS(), _NS(), _NS()
        self.trafficlight.getIDList = lambda: tuple(sorted(scenario.signal_ids))
        # (the net's <connection tl= linkIndex=> elements: what Signal.generate_config reads, traffic_signal.py:117)
        self.trafficlight.getControlledLinks = lambda sid: [[tuple(t) for t in lk] for lk in scenario.signal_meta[sid].get('controlled_links', [])]
        self.trafficlight.getAllProgramLogics = s._logics
        self.trafficlight.setProgramLogic = s._set_logic
        self.trafficlight.getPhase = lambda sid: s.orc.get_phase(s.sig_index[sid])
        self.trafficlight.setPhase = lambda sid, idx: s.orc.set_phase(s.sig_index[sid], int(idx))
        self.lane.getLastStepVehicleIDs = s._lane_vehicles
        self.vehicle.getNextTLS = s._next_tls
        self.vehicle.getWaitingTime = lambda v: float(s._veh(v)['sumo_wait'])
        self.vehicle.getSpeed = lambda v: float(s._veh(v)['speed'])
        self.vehicle.getAcceleration = lambda v: float(s._veh(v)['accel'])
        self.vehicle.getLanePosition = lambda v: float(s._veh(v)['pos'])
        self.vehicle.getTypeID = lambda v: s.sc.vtype_ids[int(s.sc.trip_vtype[s._veh(v)['trip']])]
        self.simulation.getTime = lambda: float(s.sc.begin + s.orc.time)

    # -- TLS programs
    def _logics(self, sid):
        meta = self.sc.signal_meta[sid]
        if sid in self.installed:
            return [self.installed[sid]]
        return [_Logic([_Phase(d, st) for d, st in meta['orig_program']], self.orc.get_phase(self.sig_index[sid]))]

    def _set_logic(self, sid, logic):
        # the program Signal.__init__ installs must be exactly the one the scenario compiler built
        want = [(d, st) for d, st in self.sc.signal_meta[sid]['phases']]
        got = [(p.duration, p.state) for p in logic.phases]
        assert got == want, ('create_yellows mismatch', sid, got, want)
        self.installed[sid] = logic
        i = self.sig_index[sid]
        self.orc.set_phase(i, self.orc.get_phase(i))      # [SUMO-K] restart the current index, full duration

    # -- simulation
    def simulationStep(self):
        self.orc.tick()

    def _snapshot(self):
        if self._snap_t != self.orc.time:
            v = self.orc.vehicles()
            self._snap = v
            self._by_lane, self._by_id = {}, {}
            for s_ in range(v['hw']):
                ln = int(v['lane'][s_])
                if ln >= 0xFFFE:
                    continue
                k = int(v['trip'][s_])
                rec = dict(slot=s_, trip=k, lane=ln, pos=v['pos'][s_], speed=v['speed'][s_], accel=v['accel'][s_],
                           sumo_wait=v['sumo_wait'][s_], cursor=int(v['cursor'][s_]))
                self._by_id[self.sc.trip_ids[k]] = rec
                self._by_lane.setdefault(ln, []).append(self.sc.trip_ids[k])
            self._snap_t = self.orc.time
        return self._snap

    def _veh(self, vid):
        self._snapshot()
        return self._by_id[vid]

    def _lane_vehicles(self, lane_id):
        self._snapshot()
        ci = self.lane_index.get(lane_id)
        return tuple(self._by_lane.get(ci, ())) if ci is not None else ()

    def _next_tls(self, vid):
        r = self._veh(vid)
        sc = self.sc
        route = int(sc.trip_route[r['trip']])
        td = sc.route_tlsdist[int(sc.route_start[route]) + r['cursor']]
        if td >= BIG:
            return []
        dist = np.float32(np.float32(sc.lane_len[r['lane']]) - np.float32(r['pos'])) + np.float32(td)
        return [('tls', 0, float(dist), 'G')]


_FACTORY = {'fn': None}


def install_stubs(factory):
    """Put stub traci / sumolib / gym modules in sys.modules (once).  `factory(sumo_cmd)` must return the
    FakeSumo to hand out for the next traci.start / traci.getConnection."""
    _FACTORY['fn'] = factory
    if 'traci' in sys.modules and getattr(sys.modules['traci'], '_resco_stub', False):
        return sys.modules['traci']
    os.environ.setdefault('SUMO_HOME', '/nonexistent')
    traci = types.ModuleType('traci')
    traci._resco_stub = True
    traci.trafficlight = types.SimpleNamespace(Phase=_Phase)
    conns = {}

    def start(cmd, label='default', **kw):
        conns[label] = _FACTORY['fn'](cmd)
        traci._current = conns[label]
        # libsumo mode uses the module itself as the connection
        for ns in ('lane', 'vehicle', 'simulation'):
            setattr(traci, ns, getattr(conns[label], ns))
        tl = conns[label].trafficlight
        tl.Phase = _Phase
        traci.trafficlight = tl
        traci.simulationStep = conns[label].simulationStep

    traci.start = start
    traci.getConnection = lambda label: conns[label]
    traci.switch = lambda label: None
    traci.close = lambda *a, **k: None
    sumolib = types.ModuleType('sumolib')
    sumolib.checkBinary = lambda b: b
    gym = types.ModuleType('gym')
    gym.Env = object
    gym.spaces = types.SimpleNamespace(
        Box=lambda low, high, shape, **kw: types.SimpleNamespace(low=low, high=high, shape=tuple(shape)),
        Discrete=lambda n: types.SimpleNamespace(n=n))
    gym.envs = types.ModuleType('gym.envs')
    gym.envs.registration = types.ModuleType('gym.envs.registration')
    gym.envs.registration.register = lambda *a, **k: None
    gym.register = lambda *a, **k: None
    sys.modules.update({'traci': traci, 'sumolib': sumolib, 'gym': gym, 'gym.envs': gym.envs,
                        'gym.envs.registration': gym.envs.registration})
    if REF not in sys.path:
        sys.path.insert(0, REF)
    return traci


def import_reference():
    import resco_benchmark  # noqa: F401
    from resco_benchmark import rewards, states
    from resco_benchmark.multi_signal import MultiSignal
    from resco_benchmark.traffic_signal import Signal, create_yellows
    from resco_benchmark.agents.maxpressure import MAXPRESSURE
    from resco_benchmark.agents.maxwave import MAXWAVE
    return dict(states=states, rewards=rewards, MultiSignal=MultiSignal, Signal=Signal,
                create_yellows=create_yellows, MAXPRESSURE=MAXPRESSURE, MAXWAVE=MAXWAVE)
