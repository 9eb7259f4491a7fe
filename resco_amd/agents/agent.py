"""Agent plumbing with the reference's calling convention (resco_benchmark/agents/agent.py:4-79):
``alg(config, obs_act, map_name, thread_number)``, ``act(obs_dict) -> act_dict``,
``observe(obs, rew, done, info)``."""


class Agent(object):
    def act(self, observation):
        raise NotImplementedError

    def observe(self, observation, reward, done, info):
        raise NotImplementedError


class IndependentAgent(Agent):
    """One sub-agent per signal id."""

    def __init__(self, config, obs_act, map_name, thread_number):
        self.config = config
        self.agents = {}

    def act(self, observation):
        return {k: self.agents[k].act(observation[k]) for k in observation.keys()}

    def observe(self, observation, reward, done, info):
        for k in observation.keys():
            self.agents[k].observe(observation[k], reward[k], done, info)
            if done and info['eps'] % self.config['save_freq'] == 0:
                self.agents[k].save(self.config['log_dir'] + 'agent_' + k)


class SharedAgent(Agent):
    """One policy over the batch of signals, with the per-signal valid-action maps of signal_configs."""

    def __init__(self, config, obs_act, map_name, thread_number):
        self.config = config
        self.agent = None
        self.valid_acts = None
        self.reverse_valid = None

    def act(self, observation):
        if self.reverse_valid is None and self.valid_acts is not None:
            self.reverse_valid = {sid: {v: k for k, v in d.items()} for sid, d in self.valid_acts.items()}
        keys = list(observation.keys())
        batch_obs = [observation[k] for k in keys]
        if self.valid_acts is None:
            batch_valid = batch_reverse = None
        else:
            batch_valid = [self.valid_acts.get(k) for k in keys]
            batch_reverse = [self.reverse_valid.get(k) for k in keys]
        batch_acts = self.agent.act(batch_obs, valid_acts=batch_valid, reverse_valid=batch_reverse)
        return {k: batch_acts[i] for i, k in enumerate(keys)}

    def observe(self, observation, reward, done, info):
        keys = list(observation.keys())
        self.agent.observe([observation[k] for k in keys], [reward[k] for k in keys], [done] * len(keys),
                           [False] * len(keys))
        if done and info['eps'] % self.config['save_freq'] == 0:
            self.agent.save(self.config['log_dir'] + 'agent')
