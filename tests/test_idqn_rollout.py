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


def test_fused_policy_weight_packing():
    """B fragments of the MFMA 32x32x8 f16 shape: element (kk, nt, lane, j) = w[kk*8 + (lane>>5)*4 + j][nt*32 + (lane&31)]."""
    import numpy as np
    from resco_amd.agents.idqn_fused import _b_fragments, pack_idqn_weights
    rng = np.random.default_rng(0)
    w = rng.normal(size=(2, 20, 40)).astype(np.float32)            # K = 20 rows (3 k-steps, padded), 40 columns (2 tiles, padded)
    p = _b_fragments(w, 3, 2)
    assert p.shape == (2, 3, 2, 64, 4) and p.dtype == np.float16
    for kk, nt, lane, j in [(0, 0, 0, 0), (1, 1, 37, 2), (2, 0, 63, 3), (2, 1, 5, 1), (0, 1, 40, 3)]:
        k, n = kk * 8 + (lane >> 5) * 4 + j, nt * 32 + (lane & 31)
        want = np.float16(w[1, k, n]) if (k < 20 and n < 40) else np.float16(0)
        assert p[1, kk, nt, lane, j] == want
    net = BatchedIDQN([3, 6, 4], [2, 4, 3])
    net.init_like_reference(seed=2)
    pk = pack_idqn_weights(net)
    H, hp = net.lmax - 1, net.lmax // 2
    assert pk['w1'].shape == (3, 64, hp, 2, 64, 4) and pk['w2'].shape == (3, 8, 2, 64, 4) and pk['w3'].shape == (3, 8, 64, 4)
    assert pk['conv_w'].shape == (3, 64, 4) and pk['b3'].shape == (3, 32) and pk['n_actions'].tolist() == [2, 4, 3]
    # fc1 row order (c, h, w): channel 5, position (h=2, w=1) of signal 1 -> k = 9 -> kk 1, lane group 0, j 1
    f = 5 * H * 4 + 2 * 4 + 1
    assert pk['w1'][1, 5, 1, 1, 3, 1] == np.float16(net.fc1_w[1, f, 32 + 3].item())
    assert not pk['w3'][0][:, 2:32].any() and not pk['w3'][0][:, 34:].any()        # columns beyond the 2 actions of signal 0
