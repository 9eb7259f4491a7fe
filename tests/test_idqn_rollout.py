"""CPU: the batched IDQN forward equals the reference's per-signal architecture (pfrl_dqn.py:24-40)."""
import numpy as np
import torch

from conftest import load_scenario
from resco_amd.agents.idqn_rollout import BatchedIDQN, reference_q_network


def test_batched_forward_equals_per_signal_modules():
    sc = load_scenario('ingolstadt21')
    net = BatchedIDQN.from_scenario(sc)
    mods = net.init_like_reference(seed=3)
    assert net.lmax == 17 and net.amax == 4 and len(mods) == 21
    N = 5
    rng = np.random.default_rng(0)
    obs = np.zeros((N, 21, 17, 5), np.float32)
    for s, L in enumerate(net.lanes):
        obs[:, s, :L] = rng.random((N, L, 5)).astype(np.float32)
    q = net(torch.from_numpy(obs))
    for s, (L, A) in enumerate(zip(net.lanes, net.actions)):
        ref = mods[s](torch.from_numpy(obs[:, s, :L]).unsqueeze(1))          # [N, 1, L, 5] as the reference feeds it
        np.testing.assert_allclose(q[:, s, :A].detach().numpy(), ref.detach().numpy(), rtol=1e-4, atol=1e-5)
        assert torch.isinf(q[:, s, A:]).all()
    a = net.act(torch.from_numpy(obs))
    assert a.dtype == torch.int32 and a.shape == (N, 21)
    assert (a.numpy() < np.asarray(net.actions)[None, :]).all()
    a_eps = net.act(torch.from_numpy(obs), epsilon=1.0)
    assert (a_eps.numpy() < np.asarray(net.actions)[None, :]).all() and (a_eps.numpy() >= 0).all()


def test_reference_architecture_shapes():
    m = reference_q_network(8, 4)          # cologne1: obs (1, 8, 5) -> conv (64, 7, 4)
    assert m[3].in_features == 7 * 4 * 64 and m[7].out_features == 4
    assert sum(p.numel() for p in m.parameters()) == (4 * 64 + 64) + (1792 * 64 + 64) + (64 * 64 + 64) + (64 * 4 + 4)
