"""Static controllers with the reference's agent calling convention.

``alg(config, obs_act, map_name, thread_number)``; ``act(obs: dict[id -> array]) -> dict[id -> int]``;
``observe(obs, rew, done, info)`` (a no-op: nothing is learned).  Semantics follow the reference's
agents/maxwave.py:18-38 (greedy over the valid phase pairs, first maximum wins, pairs visited in the order
of the map's ``valid_acts`` dict), agents/maxpressure.py:13-18 (the same on ``mplight`` without its phase
entry) and agents/stochastic.py:17-18 (uniform random green).

The decision tables are the very arrays the on-device agents use (``resco_amd.sim.maxwave_tables`` ->
``rs_act_maxwave``), so the host and device policies cannot drift apart.
"""
import random

import numpy as np

from ..config.signal_config import signal_configs


class _PairTables:
    """phase_pairs / valid_acts of one map as dense arrays (same layout as rs_act_maxwave)."""

    def __init__(self, map_name):
        cfg = signal_configs[map_name]
        self.pairs = np.asarray(cfg['phase_pairs'], dtype=np.int64).reshape(-1, 2)
        self.valid_acts = cfg['valid_acts']

    def choose(self, signal_id, movement_values):
        """movement_values: the 12 per-movement numbers of one signal."""
        v = np.asarray(movement_values)
        press = v[self.pairs[:, 0]] + v[self.pairs[:, 1]]
        table = None if self.valid_acts is None else self.valid_acts.get(signal_id)
        if table is None:
            return int(np.argmax(press))                      # first maximum over all pairs; action = pair index
        order = np.fromiter(table.keys(), dtype=np.int64)      # the reference iterates the dict in insertion order
        winner = order[int(np.argmax(press[order]))]           # argmax keeps the first maximum
        return int(table[int(winner)])


class _StaticAgent:
    def __init__(self, config, obs_act, map_name, thread_number):
        self.config, self.map_name = config, map_name

    def observe(self, observation, reward, done, info):
        return None

    def save(self, path):
        return None


class MAXWAVE(_StaticAgent):
    """Greedy on the `wave` state (12 per-movement counts of queued + approaching vehicles)."""
    skip = 0

    def __init__(self, config, obs_act, map_name, thread_number):
        super().__init__(config, obs_act, map_name, thread_number)
        self.tables = _PairTables(map_name)

    def act(self, observation):
        return {sid: self.tables.choose(sid, obs[self.skip:]) for sid, obs in observation.items()}


class MAXPRESSURE(MAXWAVE):
    """The same rule on the `mplight` state; its first entry (the current phase) is not a movement."""
    skip = 1


class STOCHASTIC(_StaticAgent):
    """Uniform random green per signal; obs_act[id] = [obs_shape, n_actions]."""

    def __init__(self, config, obs_act, map_name, thread_number):
        super().__init__(config, obs_act, map_name, thread_number)
        self.n_actions = {sid: spec[1] for sid, spec in obs_act.items()}
        self.rng = random.Random(config.get('seed') if isinstance(config, dict) else None)

    def act(self, observation):
        return {sid: self.rng.randrange(self.n_actions[sid]) for sid in observation}
