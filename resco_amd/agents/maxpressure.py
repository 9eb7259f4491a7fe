"""MAXPRESSURE: MAXWAVE on the `mplight` state without its phase entry
(reference: resco_benchmark/agents/maxpressure.py:6-18)."""
from ..config.signal_config import signal_configs
from .agent import SharedAgent
from .maxwave import WaveAgent


class MAXPRESSURE(SharedAgent):
    def __init__(self, config, obs_act, map_name, thread_number):
        super().__init__(config, obs_act, map_name, thread_number)
        self.valid_acts = signal_configs[map_name]['valid_acts']
        self.agent = MaxAgent(signal_configs[map_name]['phase_pairs'])


class MaxAgent(WaveAgent):
    def act(self, observation, valid_acts=None, reverse_valid=None):
        return super().act([obs[1:] for obs in observation], valid_acts, reverse_valid)
