"""MAXWAVE: greedy on the `wave` state (reference: resco_benchmark/agents/maxwave.py:7-43)."""
import numpy as np

from ..config.signal_config import signal_configs
from .agent import Agent, SharedAgent


class MAXWAVE(SharedAgent):
    def __init__(self, config, obs_act, map_name, thread_number):
        super().__init__(config, obs_act, map_name, thread_number)
        self.valid_acts = signal_configs[map_name]['valid_acts']
        self.agent = WaveAgent(signal_configs[map_name]['phase_pairs'])


class WaveAgent(Agent):
    def __init__(self, phase_pairs):
        self.phase_pairs = phase_pairs

    def act(self, observations, valid_acts=None, reverse_valid=None):
        acts = []
        for i, obs in enumerate(observations):
            if valid_acts is None:
                acts.append(np.argmax([obs[a] + obs[b] for a, b in self.phase_pairs]))
                continue
            best, best_idx = None, None
            for idx in valid_acts[i]:          # dict order; ties keep the first maximum
                a, b = self.phase_pairs[idx]
                press = obs[a] + obs[b]
                if best is None or press > best:
                    best, best_idx = press, idx
            acts.append(valid_acts[i][best_idx])
        return acts

    def observe(self, observation, reward, done, info):
        pass

    def save(self, path):
        pass
