"""ctypes wrapper of the CPU oracle (oracle/resco_oracle.c).  TEST INFRASTRUCTURE ONLY:
importable from tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg."""
import ctypes as C
import os
import subprocess

import numpy as np

from resco_amd._abi import ParamsStruct, pack_scenario

_HERE = os.path.dirname(os.path.abspath(__file__))
_LIB = os.path.join(_HERE, '_build', 'libresco_oracle.so')


def build(force=False):
    if force or not os.path.exists(_LIB) or \
            os.path.getmtime(_LIB) < os.path.getmtime(os.path.join(_HERE, 'resco_oracle.c')):
        subprocess.check_call(['make', '-C', _HERE, '-B' if force else '-s'], stdout=subprocess.DEVNULL)
    return _LIB


class _Vehicles(C.Structure):
    _fields_ = [('hw', C.c_int32), ('next_trip', C.c_int32), ('trip', C.POINTER(C.c_int32)),
                ('lane', C.POINTER(C.c_uint16)),
                ('pos', C.POINTER(C.c_float)), ('speed', C.POINTER(C.c_float)), ('accel', C.POINTER(C.c_float)),
                ('time_loss', C.POINTER(C.c_float)), ('cursor', C.POINTER(C.c_uint16)),
                ('sumo_wait', C.POINTER(C.c_uint16)), ('resco_wait', C.POINTER(C.c_uint16)),
                ('depart', C.POINTER(C.c_uint16)), ('owner', C.POINTER(C.c_uint8))]


_lib = None


def lib():
    global _lib
    if _lib is None:
        L = C.CDLL(build())
        L.orc_create.restype = C.c_void_p
        L.orc_create.argtypes = [C.c_void_p, C.c_void_p, C.c_int32]
        for f in ('orc_destroy', 'orc_reset', 'orc_tick', 'orc_observe'):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = None
        L.orc_step.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_get_phase.argtypes = [C.c_void_p, C.c_int32]
        L.orc_get_phase.restype = C.c_int32
        L.orc_set_phase.argtypes = [C.c_void_p, C.c_int32, C.c_int32]
        L.orc_time.argtypes = [C.c_void_p]
        L.orc_time.restype = C.c_int32
        for f in ('orc_lane_agg', 'orc_drq_norm', 'orc_wait', 'orc_wait_norm'):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.POINTER(C.c_float)
        L.orc_mplight_full.argtypes = [C.c_void_p]
        L.orc_mplight_full.restype = C.POINTER(C.c_float)
        L.orc_reinit_signals.argtypes = [C.c_void_p]
        L.orc_reinit_signals.restype = None
        for f in ('orc_phase', 'orc_mplight', 'orc_wave', 'orc_pressure', 'orc_queue_sum', 'orc_queue_max', 'orc_arrivals', 'orc_departures', 'orc_lane_arrivals'):
            getattr(L, f).argtypes = [C.c_void_p]
            getattr(L, f).restype = C.POINTER(C.c_int32)
        L.orc_get_vehicles.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_stats.argtypes = [C.c_void_p, C.c_void_p]
        L.orc_debug.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_backlog.argtypes = [C.c_void_p, C.c_void_p, C.c_void_p]
        L.orc_trip_log.argtypes = [C.c_void_p]
        L.orc_trip_log.restype = C.POINTER(C.c_int32)
        L.orc_wtot.argtypes = [C.c_void_p]
        L.orc_wtot.restype = C.POINTER(C.c_uint16)
        L.orc_stop_speed.restype = C.c_float
        L.orc_stop_speed.argtypes = [C.c_float] * 3
        L.orc_brake_gap.restype = C.c_float
        L.orc_brake_gap.argtypes = [C.c_float] * 2
        L.orc_follow_speed.restype = C.c_float
        L.orc_follow_speed.argtypes = [C.c_float] * 5
        L.orc_free_speed.restype = C.c_float
        L.orc_free_speed.argtypes = [C.c_float] * 3
        L.orc_hash.restype = C.c_uint32
        L.orc_hash.argtypes = [C.c_uint32] * 5
        _lib = L
    return _lib


class OracleEnv:
    """One environment instance of the CPU oracle."""

    def __init__(self, scenario, env_index=0, seed=0, max_distance=200.0, sigma=0.0, speed_dev=0,
                 fixed_program=0, step_length=10, yellow_length=None, trip_log=0, step_ratio=1, tls_expiry=1):
        self.sc = scenario
        self._st, self._keep = pack_scenario(scenario, step_length, yellow_length)
        self._p = ParamsStruct(seed, max_distance, sigma, speed_dev, fixed_program, trip_log, step_ratio, 0 if tls_expiry else 1)      # rs_params.tls_hold
        self._trip_log = trip_log
        self._h = lib().orc_create(C.byref(self._st), C.byref(self._p), env_index)
        self.S, self.O = scenario.n_signals, scenario.n_obs

    def close(self):
        if self._h:
            lib().orc_destroy(self._h)
            self._h = None

    __del__ = close

    def reset(self):
        lib().orc_reset(self._h)

    def tick(self):
        lib().orc_tick(self._h)

    def reinit_signals(self):
        lib().orc_reinit_signals(self._h)

    def observe(self):
        lib().orc_observe(self._h)

    def step(self, actions):
        a = np.ascontiguousarray(actions, dtype=np.int32)
        assert a.shape == (self.S,)
        lib().orc_step(self._h, a.ctypes.data)

    def get_phase(self, sig):
        return lib().orc_get_phase(self._h, sig)

    def set_phase(self, sig, ph):
        lib().orc_set_phase(self._h, sig, ph)

    @property
    def time(self):
        return lib().orc_time(self._h)

    def _f(self, fn, shape):
        return np.ctypeslib.as_array(getattr(lib(), fn)(self._h), shape=shape).copy()

    def outputs(self):
        S, O = self.S, self.O
        return dict(lane_agg=self._f('orc_lane_agg', (O, 5)), drq_norm=self._f('orc_drq_norm', (O, 5)),
                    phase=self._f('orc_phase', (S,)), mplight=self._f('orc_mplight', (S, 13)),
                    wave=self._f('orc_wave', (S, 12)), wait=self._f('orc_wait', (S,)),
                    wait_norm=self._f('orc_wait_norm', (S,)), pressure=self._f('orc_pressure', (S,)),
                    queue_sum=self._f('orc_queue_sum', (S,)), queue_max=self._f('orc_queue_max', (S,)),
                    arrivals=self._f('orc_arrivals', (S,)), departures=self._f('orc_departures', (S,)),
                    lane_arrivals=self._f('orc_lane_arrivals', (O,)),
                    mplight_full=self._f('orc_mplight_full', (S, 49)))

    def vehicles(self):
        v = _Vehicles()
        lib().orc_get_vehicles(self._h, C.byref(v))
        cap = self.sc.capacity
        out = dict(hw=v.hw, next_trip=v.next_trip)
        for name in ('trip', 'lane', 'pos', 'speed', 'accel', 'time_loss', 'cursor', 'sumo_wait', 'resco_wait',
                     'depart', 'owner'):
            out[name] = np.ctypeslib.as_array(getattr(v, name), shape=(cap,)).copy()
        return out

    def trip_log(self):
        assert self._trip_log
        return np.ctypeslib.as_array(lib().orc_trip_log(self._h), shape=(self.sc.n_trips, 4)).copy()

    def wtot(self):
        return np.ctypeslib.as_array(lib().orc_wtot(self._h), shape=(self.sc.capacity,)).copy()

    def debug(self):
        r, b = C.POINTER(C.c_int32)(), C.POINTER(C.c_int32)()
        lib().orc_debug(self._h, C.byref(r), C.byref(b))
        cap = self.sc.capacity
        return np.ctypeslib.as_array(r, shape=(cap,)).copy(), np.ctypeslib.as_array(b, shape=(cap,)).copy()

    def backlog_delay(self, per_lane=False):
        """(seconds waited so far, count) over the trips that have departed but are not on the network yet"""
        out = (C.c_int64 * 2)()
        pl = np.zeros(self.sc.n_lanes, np.int32)
        lib().orc_backlog(self._h, out, pl.ctypes.data)
        if per_lane:
            return float(out[1]), int(out[0]), pl
        return float(out[1]), int(out[0])

    def stats(self):
        out = (C.c_int64 * 11)()
        lib().orc_stats(self._h, out)
        keys = ['inserted', 'arrived', 'sum_duration', 'sum_depart_delay', 'sum_waiting', 'sum_time_loss_q10',
                'active', 'pending', 'active_ticks', 'ticks', 'cap_blocked']
        d = dict(zip(keys, [int(x) for x in out]))
        d['invariant'] = 0      # (resco_sim.h rs_stats [11]: the kernel's own classification invariants -- the oracle has no such notion)
        return d
