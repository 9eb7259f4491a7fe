"""STOCHASTIC: uniform random green per signal (reference: resco_benchmark/agents/stochastic.py)."""
import random

from .agent import Agent, IndependentAgent


class STOCHASTIC(IndependentAgent):
    def __init__(self, config, obs_act, map_name, thread_number):
        super().__init__(config, obs_act, map_name, thread_number)
        for key in obs_act:
            self.agents[key] = STOCHASTICAgent(obs_act[key][1])


class STOCHASTICAgent(Agent):
    def __init__(self, num_actions):
        self.num_actions = num_actions

    def act(self, observation):
        return random.randint(0, self.num_actions - 1)

    def observe(self, observation, reward, done, info):
        pass
