"""Batched IDQN rollout policy (BASELINE config 5).

The reference gives every signal its own Q-network (resco_benchmark/agents/pfrl_dqn.py:24-40):

    Conv2d(1, 64, kernel_size=(2, 2)) - ReLU - Flatten - Linear(h*w*64, 64) - ReLU - Linear(64, 64) - ReLU
    - Linear(64, n_actions)                       with (h, w) = (L - 1, 4) for an observation (1, L, 5)

evaluated one observation at a time on the host.  Here the S networks (different L and n_actions per
signal) are evaluated for all N environments at once from the kernel-produced fp16 tensor
``drq_norm_f16 [N, S, Lmax, 5]``: the 2x2 convolution becomes one einsum over unfolded patches, the three
linear layers become batched matmuls over the signal axis, padded lanes / actions are masked.  Weights can be
imported from / exported to the per-signal reference-architecture modules, so a trained IDQN plugs in.
Epsilon-greedy action selection is included; the replay ring and the DQN update live in idqn_learn.py.
"""
import torch
import torch.nn as nn


def reference_q_network(n_lanes, n_actions):
    """The per-signal model of pfrl_dqn.py:30-39 (DiscreteActionValueHead is an output wrapper, omitted)."""
    h, w = n_lanes - 1, 4
    return nn.Sequential(nn.Conv2d(1, 64, kernel_size=(2, 2)), nn.ReLU(), nn.Flatten(), nn.Linear(h * w * 64, 64),
                         nn.ReLU(), nn.Linear(64, 64), nn.ReLU(), nn.Linear(64, n_actions))


class BatchedIDQN(nn.Module):
    def __init__(self, lanes_per_signal, actions_per_signal, dtype=torch.float32, device='cpu'):
        super().__init__()
        self.lanes = [int(x) for x in lanes_per_signal]
        self.actions = [int(x) for x in actions_per_signal]
        S, self.lmax, self.amax = len(self.lanes), max(self.lanes), max(self.actions)
        H = self.lmax - 1
        kw = dict(dtype=dtype, device=device)
        self.conv_w = nn.Parameter(torch.zeros(S * 64, 1, 2, 2, **kw))     # grouped conv2d: one group per signal
        self.conv_b = nn.Parameter(torch.zeros(S * 64, **kw))
        self.fc1_w = nn.Parameter(torch.zeros(S, 64 * H * 4, 64, **kw))    # rows in the reference's Flatten order (c, h, w)
        self.fc1_b = nn.Parameter(torch.zeros(S, 64, **kw))
        self.fc2_w = nn.Parameter(torch.zeros(S, 64, 64, **kw))
        self.fc2_b = nn.Parameter(torch.zeros(S, 64, **kw))
        self.fc3_w = nn.Parameter(torch.zeros(S, 64, self.amax, **kw))
        self.fc3_b = nn.Parameter(torch.zeros(S, self.amax, **kw))
        amask = torch.zeros(S, self.amax, dtype=torch.bool)
        for s, a in enumerate(self.actions):
            amask[s, :a] = True
        self.register_buffer('action_mask', amask.to(device))

    @classmethod
    def from_scenario(cls, sc, **kw):
        lanes = (sc.sig_obs_start[1:] - sc.sig_obs_start[:-1]).tolist()
        return cls(lanes, sc.tls_ngreen.tolist(), **kw)

    # ------------------------------------------------------------------ weight exchange with the reference layout
    @torch.no_grad()
    def load_reference_modules(self, modules):
        """modules[s] = the nn.Sequential of reference_q_network(L_s, A_s) (e.g. a loaded IDQN checkpoint)."""
        H = self.lmax - 1
        for s, m in enumerate(modules):
            conv, fc1, fc2, fc3 = m[0], m[3], m[5], m[7]
            hs = self.lanes[s] - 1
            self.conv_w[s * 64:(s + 1) * 64] = conv.weight.to(self.conv_w)
            self.conv_b[s * 64:(s + 1) * 64] = conv.bias.to(self.conv_b)
            w1 = fc1.weight.reshape(64, 64, hs, 4)                       # out, c, h, w (Flatten order c, h, w)
            full = torch.zeros(64, 64, H, 4, dtype=w1.dtype)
            full[:, :, :hs] = w1                                          # rows of padded lanes stay zero
            self.fc1_w[s] = full.reshape(64, 64 * H * 4).t().to(self.fc1_w)
            self.fc1_b[s] = fc1.bias.to(self.fc1_b)
            self.fc2_w[s] = fc2.weight.t().to(self.fc2_w)
            self.fc2_b[s] = fc2.bias.to(self.fc2_b)
            self.fc3_w[s].zero_()
            self.fc3_b[s].zero_()
            self.fc3_w[s, :, :self.actions[s]] = fc3.weight.t().to(self.fc3_w)
            self.fc3_b[s, :self.actions[s]] = fc3.bias.to(self.fc3_b)
        return self

    @torch.no_grad()
    def init_like_reference(self, seed=0):
        """PyTorch default initialisation of every per-signal network (what an untrained IDQN starts from)."""
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        mods = [reference_q_network(l, a) for l, a in zip(self.lanes, self.actions)]
        torch.random.set_rng_state(g)
        self.load_reference_modules(mods)
        return mods

    # ------------------------------------------------------------------ forward
    def forward(self, obs):
        """obs [N, S, Lmax, 5] (zero padded) -> Q [N, S, Amax] (padded actions = -inf).

        The S 2x2 convolutions are ONE grouped conv2d (signals = groups, MIOpen); the three linear layers are
        signal-major strided-batched GEMMs (rocBLAS): [S] x ([N, F] @ [F, O])."""
        N, S = obs.shape[0], obs.shape[1]
        y = torch.nn.functional.conv2d(obs.to(self.conv_w.dtype), self.conv_w, self.conv_b, groups=S)   # [N, S*64, H, 4]
        y = torch.relu_(y).reshape(N, S, -1).transpose(0, 1)                # [S, N, 64*H*4] in (c, h, w) order
        y = torch.relu_(torch.baddbmm(self.fc1_b.unsqueeze(1), y, self.fc1_w))
        y = torch.relu_(torch.baddbmm(self.fc2_b.unsqueeze(1), y, self.fc2_w))
        q = torch.baddbmm(self.fc3_b.unsqueeze(1), y, self.fc3_w).transpose(0, 1)         # [N, S, Amax]
        return q.masked_fill(~self.action_mask, float('-inf'))

    @torch.no_grad()
    def act(self, obs, epsilon=0.0, generator=None):
        """epsilon-greedy actions, int32 [N, S] on obs.device."""
        q = self.forward(obs)
        greedy = q.argmax(dim=-1)
        if epsilon <= 0.0:
            return greedy.to(torch.int32)
        if not hasattr(self, '_n_act') or self._n_act.device != q.device:
            self._n_act = torch.as_tensor(self.actions, device=q.device, dtype=torch.float32)
        u = torch.rand((2,) + tuple(greedy.shape), device=q.device, generator=generator)
        rnd = torch.minimum((u[1] * self._n_act).long(), (self._n_act - 1).long())
        return torch.where(u[0] < epsilon, rnd, greedy).to(torch.int32)

