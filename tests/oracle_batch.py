"""Tests-only stand-in with BatchedSim's stepping API, backed by the CPU oracle (one OracleEnv per
environment).  Used to exercise the N>1 plumbing of bench.py on CPU (gloo) and as the expected value of the
on-device random policy."""
import numpy as np

from oracle.pyoracle import OracleEnv, lib
from resco_amd.sim import STAT_KEYS


def hashed_random_actions(sc, seed, env_base, n_envs, step_key):
    """What rs_act_random_kernel computes: murmur3(seed ^ 0xA5A5A5A5; env, signal, step_key, 7) % n_green."""
    L = lib()
    out = np.zeros((n_envs, sc.n_signals), np.int32)
    for e in range(n_envs):
        for s in range(sc.n_signals):
            out[e, s] = L.orc_hash((seed ^ 0xA5A5A5A5) & 0xFFFFFFFF, env_base + e, s, step_key & 0xFFFFFFFF, 7) % int(sc.tls_ngreen[s])
    return out


def murmur_hash(seed, a, b, c, d):
    """The counter hash shared by the simulator and the fused policy (oracle's orc_hash = the kernels' d_hash)."""
    return int(lib().orc_hash(seed & 0xFFFFFFFF, a & 0xFFFFFFFF, b & 0xFFFFFFFF, c & 0xFFFFFFFF, d & 0xFFFFFFFF)) & 0xFFFFFFFF


class OracleBatch:
    def __init__(self, sc, n_envs, seed=0, env_base=0, sigma=-1.0, speed_dev=1):
        self.sc, self.n_envs, self.seed, self.env_base = sc, n_envs, seed, env_base
        self.envs = [OracleEnv(sc, env_index=env_base + e, seed=seed, sigma=sigma, speed_dev=speed_dev)
                     for e in range(n_envs)]
        self.actions = np.zeros((n_envs, sc.n_signals), np.int32)
        self._launches = 0
        for e in self.envs:
            e.observe()

    def reset(self):
        for e in self.envs:
            e.reset()
            e.observe()

    def act_random(self, step_key):
        self.actions = hashed_random_actions(self.sc, self.seed, self.env_base, self.n_envs, step_key)

    def step(self, actions=None):
        a = self.actions if actions is None else actions
        for i, e in enumerate(self.envs):
            e.step(a[i])
        self._launches += 1

    def sync(self):
        pass

    def timing(self, enable):
        self._launches = 0

    def timing_read(self):
        return 1.0 * self._launches, self._launches

    def stats(self):
        rows = [e.stats() for e in self.envs]
        return {k: np.array([r[k] for r in rows], np.int64) for k in STAT_KEYS}

    def read(self, name):
        return np.stack([e.outputs()[name] for e in self.envs])
