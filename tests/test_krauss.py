"""CPU: known-answer and property tests of the oracle's car-following arithmetic and counter-based RNG."""
import ctypes as C
import math

import numpy as np

from oracle.pyoracle import lib


def murmur3_32(words, seed):
    """Independent restatement of MurmurHash3_x86_32 over 4 little-endian uint32 words."""
    def rotl(x, r):
        return ((x << r) | (x >> (32 - r))) & 0xFFFFFFFF
    h = seed & 0xFFFFFFFF
    for k in words:
        k = (k * 0xcc9e2d51) & 0xFFFFFFFF
        k = rotl(k, 15)
        k = (k * 0x1b873593) & 0xFFFFFFFF
        h ^= k
        h = rotl(h, 13)
        h = (h * 5 + 0xe6546b64) & 0xFFFFFFFF
    h ^= 16
    h ^= h >> 16
    h = (h * 0x85ebca6b) & 0xFFFFFFFF
    h ^= h >> 13
    h = (h * 0xc2b2ae35) & 0xFFFFFFFF
    h ^= h >> 16
    return h


def test_hash_is_murmur3():
    L = lib()
    rng = np.random.default_rng(0)
    for _ in range(200):
        seed, a, b, c, d = [int(x) for x in rng.integers(0, 2 ** 32, 5)]
        assert L.orc_hash(seed, a, b, c, d) == murmur3_32([a, b, c, d], seed)
    # published MurmurHash3_x86_32 vector: 16 zero bytes, seed 0
    assert murmur3_32([0, 0, 0, 0], 0) == L.orc_hash(0, 0, 0, 0, 0)


def test_brake_gap_known_values():
    L = lib()
    # SUMO brakeGapEuler(v, b, 0): steps = int(v/b); steps*v - b*steps*(steps+1)/2
    assert L.orc_brake_gap(13.89, 4.5) == np.float32(3 * np.float32(13.89) - np.float32(4.5) * 3 * 4 * 0.5)
    assert L.orc_brake_gap(4.4, 4.5) == 0.0
    assert abs(L.orc_brake_gap(27.78, 4.5) - (6 * 27.78 - 4.5 * 21)) < 1e-4


def test_stop_speed_properties():
    L = lib()
    b, tau = 4.5, 1.0
    assert L.orc_stop_speed(0.0, b, tau) == 0.0
    assert L.orc_stop_speed(-3.0, b, tau) == 0.0
    prev = 0.0
    for g in np.linspace(0.01, 300, 400):
        v = L.orc_stop_speed(float(g), b, tau)
        assert v >= prev - 1e-4            # monotone in the gap
        prev = v
        # driving v for tau seconds and then braking with b per step stops within the gap
        dist, vv = v * tau, v
        while vv > 0:
            vv = max(0.0, vv - b)
            dist += vv
        assert dist <= g + 1e-2
    # closed form check against a float64 evaluation of SUMO's maximumSafeStopSpeedEuler
    for g in (1.0, 7.3, 42.0, 150.0):
        gg = g - 0.001
        n = math.floor(0.5 - (tau + math.sqrt(1 + 4 * ((2 * gg / b - tau) + tau * tau)) * -0.5))
        h = 0.5 * n * (n - 1) * b + n * b * tau
        want = n * b + (gg - h) / (n + tau)
        assert abs(L.orc_stop_speed(g, b, tau) - want) < 1e-4


def test_follow_speed_is_stop_speed_plus_leader_brake_gap():
    L = lib()
    v = L.orc_follow_speed(10.0, 8.0, 4.5, 4.5, 1.0)
    assert v == L.orc_stop_speed(np.float32(10.0) + np.float32(L.orc_brake_gap(8.0, 4.5)), 4.5, 1.0)
    assert L.orc_follow_speed(-1.0, 0.0, 4.5, 4.5, 1.0) == 0.0
    # a harder-braking leader is followed with its deceleration
    assert L.orc_follow_speed(10.0, 8.0, 4.5, 9.0, 1.0) <= v
