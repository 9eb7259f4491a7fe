"""agent_configs for the static controllers (same keys as resco_benchmark/config/agent_config.py:65-82):
which state / reward functions and detector range each one is run with.  Learner hyper-parameters
(IDQN, IPPO, MPLight, FMA2C) are out of scope of the simulator build."""
from .. import rewards, states
from ..agents.static_agents import MAXPRESSURE, MAXWAVE, STOCHASTIC

agent_configs = {
    'STOCHASTIC': {'agent': STOCHASTIC, 'state': states.mplight, 'reward': rewards.wait, 'max_distance': 1},
    'MAXWAVE': {'agent': MAXWAVE, 'state': states.wave, 'reward': rewards.wait, 'max_distance': 50},
    'MAXPRESSURE': {'agent': MAXPRESSURE, 'state': states.mplight, 'reward': rewards.wait, 'max_distance': 200},
    'MAXWAVEVAL': {'agent': MAXWAVE, 'state': states.wave, 'reward': rewards.wait, 'max_distance': 50},
    'MAXPRESSUREVAL': {'agent': MAXPRESSURE, 'state': states.mplight, 'reward': rewards.wait,
                       'max_distance': 9999},
}
