"""Replay ring + batched DQN learner against an unbatched PyTorch fp32 restatement of the
published DQN update the reference configures (pfrl_dqn.py:49-80, agent_config.py:83-94).  PFRL itself is not
installed, so this is the floating-point reference; tolerance 2e-5 absolute on weights after 4 Adam steps."""
import copy

import numpy as np
import pytest
import torch

from resco_amd.agents.idqn_learn import BatchedDQNLearner, DeviceReplay, linear_epsilon
from resco_amd.agents.idqn_rollout import BatchedIDQN

LANES, ACTS = [3, 5, 4], [2, 4, 3]


def _obs(n, rng):
    o = torch.zeros(n, len(LANES), max(LANES), 5)
    for s, l in enumerate(LANES):
        o[:, s, :l] = torch.as_tensor(rng.random((n, l, 5)), dtype=torch.float32)
    return o


def test_linear_epsilon():
    assert linear_epsilon(0, 1.0, 0.0, 100) == 1.0
    assert abs(linear_epsilon(25, 1.0, 0.0, 100) - 0.75) < 1e-12
    assert linear_epsilon(100, 1.0, 0.0, 100) == 0.0 and linear_epsilon(1000, 1.0, 0.1, 100) == 0.1


def test_replay_ring_links_successors_and_wraps():
    T, N, S, L = 5, 3, 2, 4
    rp = DeviceReplay(T, N, S, L, device='cpu', obs_dtype=torch.float32)
    assert len(rp) == 0
    g = torch.Generator().manual_seed(1)
    for step in range(12):
        obs = torch.full((N, S, L, 5), float(step))
        obs[..., 0] += torch.arange(N).view(N, 1, 1) * 0.01                 # env tag
        obs[..., 1] += torch.arange(S).view(1, S, 1) * 0.001                # signal tag
        rp.push(obs, torch.full((N, S), step % 3), torch.full((N, S), -float(step)), done=(step % 4 == 3))
        if step == 0:
            assert len(rp) == 0
            continue
        assert len(rp) == min(step, T - 1) * N
        o, a, r, o2, d = rp.sample(64, g)
        assert o.shape == (64, S, L, 5) and a.dtype == torch.long and d.shape == (64, S)
        t = torch.round(o[..., 2, 2])                                       # step index of the sampled slot
        assert torch.all(t >= max(0, step - (T - 1))) and torch.all(t <= step - 1)     # newest slot has no successor
        assert torch.equal(torch.round(o2[..., 2, 2]), t + 1)               # successor = next slot, also across the wrap
        assert torch.allclose(o2[..., 0, 0] - o2[..., 0, 2], o[..., 0, 0] - o[..., 0, 2], atol=1e-6)   # same env
        sig = torch.arange(S).float() * 0.001
        assert torch.allclose(o[..., 0, 1] - o[..., 0, 2], sig.expand(64, S), atol=1e-6)        # column s = signal s
        assert torch.equal(a, (t % 3).long()) and torch.equal(r, -t)
        assert torch.equal(d, (t % 4 == 3).float())


def _reference_update(mods, targets, opts, batch, gamma):
    """One DQN update per signal, unbatched: the published rule on the reference architecture."""
    o, a, r, o2, d = batch
    for s, (m, tm, opt) in enumerate(zip(mods, targets, opts)):
        l = LANES[s]
        x, x2 = o[:, s, :l].unsqueeze(1), o2[:, s, :l].unsqueeze(1)         # [B, 1, L, 5] as drq_norm hands it over
        y = m(x).gather(1, a[:, s:s + 1]).squeeze(1)
        with torch.no_grad():
            t = r[:, s] + gamma * (1.0 - d[:, s]) * tm(x2).max(dim=1).values
        loss = torch.nn.functional.smooth_l1_loss(y, t)
        opt.zero_grad()
        loss.backward()
        opt.step()


def test_batched_learner_equals_per_signal_dqn():
    rng = np.random.default_rng(0)
    net = BatchedIDQN(LANES, ACTS)
    mods = net.init_like_reference(seed=3)
    targets = [copy.deepcopy(m) for m in mods]
    opts = [torch.optim.Adam(m.parameters()) for m in mods]
    learner = BatchedDQNLearner(net, gamma=0.99, batch_size=16, target_update=2)
    B = 16
    for it in range(4):
        o, o2 = _obs(B, rng), _obs(B, rng)
        a = torch.stack([torch.as_tensor(rng.integers(0, n, B)) for n in ACTS], dim=1).long()
        r = torch.as_tensor(rng.normal(size=(B, len(LANES))), dtype=torch.float32) * 3.0      # both Huber branches
        d = torch.as_tensor(rng.random((B, len(LANES))) < 0.3, dtype=torch.float32)
        batch = (o, a, r, o2, d)
        _reference_update(mods, targets, opts, batch, 0.99)
        learner.update(batch)
        if it == 1:                         # hard target copy, as every target_update agent steps
            learner.sync_target()
            for m, tm in zip(mods, targets):
                tm.load_state_dict(m.state_dict())
    probe = _obs(7, rng)
    q = net(probe)
    H = max(LANES) - 1
    for s, m in enumerate(mods):
        l, na = LANES[s], ACTS[s]
        ref_q = m(probe[:, s, :l].unsqueeze(1))
        assert torch.allclose(q[:, s, :na], ref_q, atol=2e-5), (s, (q[:, s, :na] - ref_q).abs().max())
        assert torch.all(torch.isinf(q[:, s, na:]))
        assert torch.allclose(net.fc2_w[s], m[5].weight.t(), atol=2e-5)
        assert torch.allclose(net.conv_w[s * 64:(s + 1) * 64], m[0].weight, atol=2e-5)
        w1 = net.fc1_w[s].t().reshape(64, 64, H, 4)
        assert torch.allclose(w1[:, :, :l - 1], m[3].weight.reshape(64, 64, l - 1, 4), atol=2e-5)
        assert torch.count_nonzero(w1[:, :, l - 1:]) == 0                   # padded lanes never leak into fc1
        assert torch.count_nonzero(net.fc3_w[s, :, na:]) == 0 and torch.count_nonzero(net.fc3_b[s, na:]) == 0


def test_observe_step_schedule():
    rng = np.random.default_rng(1)
    net = BatchedIDQN(LANES, ACTS)
    net.init_like_reference(seed=0)
    learner = BatchedDQNLearner(net, batch_size=8, target_update=5)
    rp = DeviceReplay(16, 4, len(LANES), max(LANES), device='cpu', obs_dtype=torch.float32)
    g = torch.Generator().manual_seed(0)
    before = copy.deepcopy(learner.target.state_dict())
    for step in range(1, 6):
        rp.push(_obs(4, rng), torch.zeros(4, 3), torch.ones(4, 3), done=False)
        online_before = copy.deepcopy(net.state_dict())
        out = learner.observe_step(rp, g)
        assert (out is None) == (len(rp) < 8)                              # replay_start_size = minibatch size
        if step < 5:
            assert all(torch.equal(before[k], v) for k, v in learner.target.state_dict().items())
    assert learner.t == 5 and learner.n_updates == 3
    # hard copy at t = 5, BEFORE that step's update (PFRL's DQN.batch_observe_train order)
    assert all(torch.equal(v, learner.target.state_dict()[k]) for k, v in online_before.items())
    assert any(not torch.equal(v, learner.target.state_dict()[k]) for k, v in net.state_dict().items())
