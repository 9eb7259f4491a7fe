"""Reward-function registry: callables f(signals: dict[id -> Signal]) -> dict[id -> scalar].

Same names and formulas as the reference's resco_benchmark/rewards.py (wait :6, wait_norm :17,
pressure :28); see states.py for the slow-path / fast-path split.
"""
import numpy as np

from .config.mdp_config import mdp_configs
from .states import _fma2c_regions


def _total_wait(signal):
    total = 0
    for lane in signal.lanes:
        total += signal.full_observation[lane]['total_wait']
    return total


def wait(signals):
    return {sid: -_total_wait(signal) for sid, signal in signals.items()}


def wait_norm(signals):
    return {sid: np.clip(-_total_wait(signal) / 224, -4, 4).astype(np.float32) for sid, signal in signals.items()}


def pressure(signals):
    out = {}
    for sid, signal in signals.items():
        q = 0
        for lane in signal.lanes:
            q += signal.full_observation[lane]['queue']
        for lane in signal.outbound_lanes:
            dwn = signal.out_lane_to_signalid[lane]
            if dwn in signal.signals:
                q -= signal.signals[dwn].full_observation[lane]['queue']
        out[sid] = -q
    return out


def _fma2c_rewards(signals, cfg):
    sup = cfg['supervisors']
    fringes = _fma2c_regions(signals, cfg)
    fringe_arrivals = {mgr: 0 for mgr in cfg['management']}
    liquidity = {mgr: 0 for mgr in cfg['management']}
    for sid, signal in signals.items():
        mgr = sup[sid]
        fo = signal.full_observation
        arrivals = fo['arrivals']
        liquidity[mgr] += len(fo['departures']) - len(arrivals)
        for lane in signal.lanes:
            if lane in fringes[mgr]:
                fringe_arrivals[mgr] += sum(1 for vehicle in fo[lane]['vehicles'] if vehicle['id'] in arrivals)
    managers = {}
    for mgr in cfg['management']:
        r = fringe_arrivals[mgr] + liquidity[mgr]
        for n in cfg['management_neighbors'][mgr]:
            r += cfg['alpha'] * (fringe_arrivals[n] + liquidity[n])
        managers[mgr] = r
    own = {}
    for sid, signal in signals.items():
        r = 0
        for lane in signal.lanes:
            r += signal.full_observation[lane]['queue']
            r += signal.full_observation[lane]['max_wait'] * cfg['coef']
        own[sid] = -r
    out = {}
    for sid, signal in signals.items():
        total = own[sid]
        for neighbour in signal.downstream.values():
            if neighbour is not None and sup[neighbour] == sup[sid]:
                total += cfg['alpha'] * own[neighbour]
        out[sid] = total
    out.update(managers)
    return out


def fma2c(signals):
    """Worker + manager rewards of FMA2C (reference rewards.py:72-136)."""
    return _fma2c_rewards(signals, mdp_configs['FMA2C'])


def fma2c_full(signals):
    """reference rewards.py:139-202"""
    return _fma2c_rewards(signals, mdp_configs['FMA2CFull'])


wait.fast_buffer = 'wait'
wait_norm.fast_buffer = 'wait_norm'
pressure.fast_buffer = 'pressure'

REGISTRY = {f.__name__: f for f in (wait, wait_norm, pressure, fma2c, fma2c_full)}
