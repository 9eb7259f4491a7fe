"""State-function registry: callables f(signals: dict[id -> Signal]) -> dict[id -> np.ndarray].

Same names, shapes and quirks as the reference's resco_benchmark/states.py (drq :6, drq_norm :34,
mplight :62, mplight_full :83, wave :116).  Every function works on any mapping of Signal-like objects
(slow path, host arithmetic over Signal.full_observation); functions tagged with ``fast_buffer`` are
additionally produced by the HIP step kernel itself and MultiSignal hands those buffers out directly.
"""
import numpy as np


def _lane_speed_sum(signal, lane, scale=1.0):
    total = 0
    for vehicle in signal.full_observation[lane]['vehicles']:
        total += vehicle['speed'] * scale
    return total


def drq(signals):
    out = {}
    for sid, signal in signals.items():
        act = signal.phase
        rows = []
        for i, lane in enumerate(signal.lanes):
            fo = signal.full_observation[lane]
            # quirk kept from the reference: lane POSITION i is compared with the PHASE index
            rows.append([1 if i == act else 0, fo['approach'], fo['total_wait'], fo['queue'],
                         _lane_speed_sum(signal, lane)])
        out[sid] = np.expand_dims(np.asarray(rows), axis=0)
    return out


def drq_norm(signals):
    out = {}
    for sid, signal in signals.items():
        act = signal.phase
        rows = []
        for i, lane in enumerate(signal.lanes):
            fo = signal.full_observation[lane]
            total_speed = 0
            for vehicle in fo['vehicles']:
                total_speed += vehicle['speed'] / 20 / 28
            rows.append([1 if i == act else 0, fo['approach'] / 28, fo['total_wait'] / 28, fo['queue'] / 28,
                         total_speed])
        out[sid] = np.expand_dims(np.asarray(rows), axis=0)
    return out


def _movement_pressure(signal, direction):
    q = 0
    for lane in signal.lane_sets[direction]:
        q += signal.full_observation[lane]['queue']
    for lane in signal.lane_sets_outbound[direction]:
        dwn = signal.out_lane_to_signalid[lane]
        if dwn in signal.signals:
            q -= signal.signals[dwn].full_observation[lane]['queue']
    return q


def mplight(signals):
    out = {}
    for sid, signal in signals.items():
        obs = [signal.phase]
        for direction in signal.lane_sets:
            obs.append(_movement_pressure(signal, direction))
        out[sid] = np.asarray(obs)
    return out


def mplight_full(signals):
    out = {}
    for sid, signal in signals.items():
        obs = [signal.phase]
        for direction in signal.lane_sets:
            total_wait, total_speed, tot_approach = 0, 0, 0
            for lane in signal.lane_sets[direction]:
                fo = signal.full_observation[lane]
                total_wait += fo['total_wait'] / 28
                # quirk kept from the reference: the speed sum restarts for every lane (last lane wins)
                total_speed = _lane_speed_sum(signal, lane)
                tot_approach += fo['approach'] / 28
            obs += [_movement_pressure(signal, direction), total_wait, total_speed, tot_approach]
        out[sid] = np.asarray(obs)
    return out


def wave(signals):
    out = {}
    for sid, signal in signals.items():
        state = []
        for direction in signal.lane_sets:
            s = 0
            for lane in signal.lane_sets[direction]:
                fo = signal.full_observation[lane]
                s += fo['queue'] + fo['approach']
            state.append(s)
        out[sid] = np.asarray(state)
    return out


# buffers the step kernel emits for these registry entries
drq_norm.fast_buffer = 'drq_norm'
mplight.fast_buffer = 'mplight'
wave.fast_buffer = 'wave'
drq.fast_buffer = 'lane_agg'            # assembled on the host from the lane aggregates + phase
mplight_full.fast_buffer = None

REGISTRY = {f.__name__: f for f in (drq, drq_norm, mplight, mplight_full, wave)}
