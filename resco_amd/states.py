"""State-function registry: callables f(signals: dict[id -> Signal]) -> dict[id -> np.ndarray].

Same names, shapes and quirks as the reference's resco_benchmark/states.py (drq :6, drq_norm :34,
mplight :62, mplight_full :83, wave :116).  Every function works on any mapping of Signal-like objects
(slow path, host arithmetic over Signal.full_observation); functions tagged with ``fast_buffer`` are
additionally produced by the HIP step kernel itself and MultiSignal hands those buffers out directly.
"""
import numpy as np

from .config.mdp_config import mdp_configs


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


def _fma2c_regions(signals, cfg):
    """manager -> inbound lanes on the fringe of its region (a downstream direction that leads out of the
    region or nowhere contributes the lanes arriving FROM that direction)"""
    sup = cfg['supervisors']
    fringes = {mgr: [] for mgr in cfg['management']}
    for sid, signal in signals.items():
        for direction, neighbour in signal.downstream.items():
            if neighbour is None or sup[neighbour] != sup[sid]:
                inbound = signal.inbounds_fr_direction.get(direction)
                if inbound is not None:
                    fringes[sup[sid]] += inbound
    return fringes


def _fma2c_states(signals, cfg, full):
    sup = cfg['supervisors']
    lane_wave = {}
    for signal in signals.values():
        for lane in signal.lanes:
            fo = signal.full_observation[lane]
            lane_wave[lane] = fo['queue'] + fo['approach']
    fringes = _fma2c_regions(signals, cfg)
    mgr_obs = {mgr: np.clip(np.asarray([lane_wave[l] for l in lanes]) / cfg['norm_wave'], 0, cfg['clip_wave'])
               for mgr, lanes in fringes.items()}
    managers = {mgr: np.concatenate([mgr_obs[mgr]] + [cfg['alpha'] * mgr_obs[n] for n in cfg['management_neighbors'][mgr]])
                for mgr in mgr_obs}
    own = {}
    for sid, signal in signals.items():
        feats = []
        for lane in signal.lanes:
            feats.append(lane_wave[lane])
            if full:    # fma2c_full interleaves total_wait / 28 and the drq_norm speed term per lane
                fo = signal.full_observation[lane]
                feats.append(fo['total_wait'] / 28)
                total_speed = 0
                for vehicle in fo['vehicles']:
                    total_speed += vehicle['speed'] / 20 / 28
                feats.append(total_speed)
        own[sid] = np.clip(np.asarray(feats) / cfg['norm_wave'], 0, cfg['clip_wave'])
    out = {}
    for sid, signal in signals.items():
        waves = [own[sid]]
        for neighbour in signal.downstream.values():
            if neighbour is not None and sup[neighbour] == sup[sid]:
                waves.append(cfg['alpha'] * own[neighbour])
        waits = np.clip(np.asarray([signal.full_observation[l]['max_wait'] for l in signal.lanes]) / cfg['norm_wait'],
                        0, cfg['clip_wait'])
        out[sid] = np.concatenate([np.concatenate(waves), waits])
    out.update(managers)
    return out


def fma2c(signals):
    """Worker + manager observations of FMA2C (reference states.py:162-229); needs
    config.mdp_config.activate('FMA2C', map) like the reference's main.py:48-72."""
    return _fma2c_states(signals, mdp_configs['FMA2C'], full=False)


def fma2c_full(signals):
    """reference states.py:232-305"""
    return _fma2c_states(signals, mdp_configs['FMA2CFull'], full=True)


# buffers the step kernel emits for these registry entries
drq_norm.fast_buffer = 'drq_norm'
mplight.fast_buffer = 'mplight'
wave.fast_buffer = 'wave'
drq.fast_buffer = 'lane_agg'            # assembled on the host from the lane aggregates + phase
mplight_full.fast_buffer = None

REGISTRY = {f.__name__: f for f in (drq, drq_norm, mplight, mplight_full, wave, fma2c, fma2c_full)}
