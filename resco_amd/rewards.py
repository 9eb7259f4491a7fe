"""Reward-function registry: callables f(signals: dict[id -> Signal]) -> dict[id -> scalar].

Same names and formulas as the reference's resco_benchmark/rewards.py (wait :6, wait_norm :17,
pressure :28); see states.py for the slow-path / fast-path split.
"""
import numpy as np


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


wait.fast_buffer = 'wait'
wait_norm.fast_buffer = 'wait_norm'
pressure.fast_buffer = 'pressure'

REGISTRY = {f.__name__: f for f in (wait, wait_norm, pressure)}
