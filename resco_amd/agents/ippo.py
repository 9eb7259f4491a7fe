"""Batched IPPO: the S per-signal actor-critic networks of the reference and their PPO update, for N lock-step envs.

The reference wraps one `pfrl.agents.PPO` per signal (resco_benchmark/agents/pfrl_ppo.py:38-75):

    Conv2d(1, 64, (2,2)) - ReLU - Flatten - Linear(h*w*64, 64) - ReLU - Linear(64, 64) - ReLU -
    Branched( Linear(64, A) [LeCun-normal, gain 1e-2] + SoftmaxCategoricalHead ,  Linear(64, 1) )
    Adam(lr 2.5e-4, eps 1e-5); clip_eps 0.1, no value clipping, update_interval 1024, minibatch 256, epochs 4,
    standardize_advantages, entropy_coef 0.001, max_grad_norm 0.5     (PFRL defaults: gamma 0.99, lambda 0.95,
    value_func_coef 1.0)

PFRL is not vendored in the reference tree and not installed here, so - as for DQN (idqn_learn.py) - this module
restates the published PPO update with those settings and is checked against an unbatched PyTorch fp32 restatement
built from `reference_ppo_network` (tests/test_ippo.py), not against PFRL itself.

What is different by design: the S agents are one set of stacked parameters under one Adam (element-wise, disjoint
parameters: equal to S optimisers); the loss is the sum of the per-signal losses; gradient clipping and advantage
standardisation are done per signal as S separate agents would.  A rollout segment holds T env-steps of all N
environments (T*N samples per signal per update instead of 1024 consecutive steps of one environment).
Acting can go through the fused HIP kernel (`FusedIDQN(net).act(obs, sample=True)`: same trunk, softmax sampling).
"""
import math

import torch
import torch.nn as nn


class ReferencePPONet(nn.Module):
    """The per-signal model of pfrl_ppo.py:49-64 (SoftmaxCategoricalHead = softmax over the returned logits)."""

    def __init__(self, n_lanes, n_actions):
        super().__init__()
        h, w = n_lanes - 1, 4
        self.trunk = nn.Sequential(nn.Conv2d(1, 64, kernel_size=(2, 2)), nn.ReLU(), nn.Flatten(), nn.Linear(h * w * 64, 64),
                                   nn.ReLU(), nn.Linear(64, 64), nn.ReLU())
        self.pi = nn.Linear(64, n_actions)
        self.v = nn.Linear(64, 1)
        for layer, gain in ((self.trunk[0], 1.0), (self.trunk[3], 1.0), (self.trunk[5], 1.0), (self.pi, 1e-2), (self.v, 1.0)):
            fan_in = layer.weight[0].numel()                     # init_lecun_normal: N(0, gain / sqrt(fan_in)), zero bias
            nn.init.normal_(layer.weight, 0.0, gain / math.sqrt(fan_in))
            nn.init.zeros_(layer.bias)

    def forward(self, x):
        z = self.trunk(x)
        return self.pi(z), self.v(z).squeeze(-1)


class BatchedIPPO(nn.Module):
    def __init__(self, lanes_per_signal, actions_per_signal, dtype=torch.float32, device='cpu'):
        super().__init__()
        self.lanes = [int(x) for x in lanes_per_signal]
        self.actions = [int(x) for x in actions_per_signal]
        S, self.lmax, self.amax = len(self.lanes), max(self.lanes), max(self.actions)
        H = self.lmax - 1
        kw = dict(dtype=dtype, device=device)
        self.conv_w = nn.Parameter(torch.zeros(S * 64, 1, 2, 2, **kw))
        self.conv_b = nn.Parameter(torch.zeros(S * 64, **kw))
        self.fc1_w = nn.Parameter(torch.zeros(S, 64 * H * 4, 64, **kw))
        self.fc1_b = nn.Parameter(torch.zeros(S, 64, **kw))
        self.fc2_w = nn.Parameter(torch.zeros(S, 64, 64, **kw))
        self.fc2_b = nn.Parameter(torch.zeros(S, 64, **kw))
        self.fc3_w = nn.Parameter(torch.zeros(S, 64, self.amax, **kw))      # policy head (named as FusedIDQN packs it)
        self.fc3_b = nn.Parameter(torch.zeros(S, self.amax, **kw))
        self.v_w = nn.Parameter(torch.zeros(S, 64, 1, **kw))
        self.v_b = nn.Parameter(torch.zeros(S, 1, **kw))
        amask = torch.zeros(S, self.amax, dtype=torch.bool)
        for s, a in enumerate(self.actions):
            amask[s, :a] = True
        self.register_buffer('action_mask', amask.to(device))
        fmask = torch.zeros(S, 64, H, 4, dtype=dtype)
        for s, l in enumerate(self.lanes):
            fmask[s, :, :l - 1] = 1.0
        self.register_buffer('fc1_mask', fmask.reshape(S, 64 * H * 4, 1).to(device))
        # padded lanes feed relu(conv bias) into fc1 and padded actions are masked: their weights stay zero
        self.fc1_w.register_hook(lambda g: g * self.fc1_mask)
        self.fc3_w.register_hook(lambda g: g * self.action_mask.unsqueeze(1))
        self.fc3_b.register_hook(lambda g: g * self.action_mask)

    @classmethod
    def from_scenario(cls, sc, **kw):
        lanes = (sc.sig_obs_start[1:] - sc.sig_obs_start[:-1]).tolist()
        return cls(lanes, sc.tls_ngreen.tolist(), **kw)

    @torch.no_grad()
    def load_reference_modules(self, modules):
        """modules[s]: a ReferencePPONet(L_s, A_s) (or any module with the same .trunk / .pi / .v)."""
        H = self.lmax - 1
        for s, m in enumerate(modules):
            conv, fc1, fc2 = m.trunk[0], m.trunk[3], m.trunk[5]
            hs, A = self.lanes[s] - 1, self.actions[s]
            self.conv_w[s * 64:(s + 1) * 64] = conv.weight.to(self.conv_w)
            self.conv_b[s * 64:(s + 1) * 64] = conv.bias.to(self.conv_b)
            full = torch.zeros(64, 64, H, 4, dtype=fc1.weight.dtype)
            full[:, :, :hs] = fc1.weight.reshape(64, 64, hs, 4)
            self.fc1_w[s] = full.reshape(64, 64 * H * 4).t().to(self.fc1_w)
            self.fc1_b[s] = fc1.bias.to(self.fc1_b)
            self.fc2_w[s] = fc2.weight.t().to(self.fc2_w)
            self.fc2_b[s] = fc2.bias.to(self.fc2_b)
            self.fc3_w[s].zero_()
            self.fc3_b[s].zero_()
            self.fc3_w[s, :, :A] = m.pi.weight.t().to(self.fc3_w)
            self.fc3_b[s, :A] = m.pi.bias.to(self.fc3_b)
            self.v_w[s] = m.v.weight.t().to(self.v_w)
            self.v_b[s] = m.v.bias.to(self.v_b)
        return self

    @torch.no_grad()
    def init_like_reference(self, seed=0):
        g = torch.random.get_rng_state()
        torch.manual_seed(seed)
        mods = [ReferencePPONet(l, a) for l, a in zip(self.lanes, self.actions)]
        torch.random.set_rng_state(g)
        self.load_reference_modules(mods)
        return mods

    def forward(self, obs):
        """obs [B, S, Lmax, 5] -> logits [B, S, Amax] (-inf beyond a signal's actions), value [B, S]."""
        B, S = obs.shape[0], obs.shape[1]
        y = torch.nn.functional.conv2d(obs.to(self.conv_w.dtype), self.conv_w, self.conv_b, groups=S)
        y = torch.relu(y).reshape(B, S, -1).transpose(0, 1)
        y = torch.relu(torch.baddbmm(self.fc1_b.unsqueeze(1), y, self.fc1_w))
        y = torch.relu(torch.baddbmm(self.fc2_b.unsqueeze(1), y, self.fc2_w))
        logits = torch.baddbmm(self.fc3_b.unsqueeze(1), y, self.fc3_w).transpose(0, 1)
        value = torch.baddbmm(self.v_b.unsqueeze(1), y, self.v_w).squeeze(-1).transpose(0, 1)
        return logits.masked_fill(~self.action_mask, float('-inf')), value

    @torch.no_grad()
    def act(self, obs, generator=None):
        logits, _ = self.forward(obs)
        p = torch.softmax(logits.float(), dim=-1)
        return torch.multinomial(p.reshape(-1, p.shape[-1]), 1, generator=generator).reshape(p.shape[:-1]).to(torch.int32)


def gae(rew, val, next_val, done, gamma=0.99, lambd=0.95):
    """Generalised advantage estimates over a segment.  rew, val [T, ...]; next_val [...] = value of the state after
    the last step; done [T] bool: the episode ended after step t (no bootstrap across it)."""
    T = rew.shape[0]
    adv = torch.zeros_like(rew)
    last = torch.zeros_like(rew[0])
    for t in range(T - 1, -1, -1):
        nv = next_val if t == T - 1 else val[t + 1]
        nd = 1.0 - done[t].to(rew.dtype)
        delta = rew[t] + gamma * nd * nv - val[t]
        last = delta + gamma * lambd * nd * last
        adv[t] = last
    return adv, adv + val


class BatchedPPOLearner:
    """PPO update of the S stacked actor-critics of a BatchedIPPO from one rollout segment."""

    def __init__(self, net, lr=2.5e-4, adam_eps=1e-5, gamma=0.99, lambd=0.95, clip_eps=0.1, epochs=4, minibatch=256,
                 entropy_coef=0.001, value_coef=1.0, max_grad_norm=0.5):
        assert isinstance(net, BatchedIPPO)
        self.net = net
        self.opt = torch.optim.Adam(net.parameters(), lr=lr, eps=adam_eps)
        self.gamma, self.lambd, self.clip_eps, self.epochs, self.minibatch = gamma, lambd, clip_eps, int(epochs), int(minibatch)
        self.entropy_coef, self.value_coef, self.max_grad_norm = entropy_coef, value_coef, max_grad_norm
        self.n_updates = 0

    # ---- pieces the test compares one by one with the per-signal restatement
    @torch.no_grad()
    def make_dataset(self, obs, act, rew, done, last_obs, chunk=4096):
        """obs [T, N, S, L, 5], act [T, N, S] int, rew [T, N, S], done [T] bool, last_obs [N, S, L, 5]: old log-probs and
        values from the current parameters, GAE, per-signal standardised advantages; flattened to [T*N, S, ...]."""
        T, N, S = act.shape
        flat = torch.cat([obs.reshape(T * N, *obs.shape[2:]), last_obs], 0)
        lg, vs = [], []
        for i in range(0, flat.shape[0], chunk):
            l_, v_ = self.net(flat[i:i + chunk])
            lg.append(torch.log_softmax(l_.float(), -1))
            vs.append(v_.float())
        logp_all, v_all = torch.cat(lg, 0), torch.cat(vs, 0)
        logp = logp_all[:T * N].gather(-1, act.reshape(T * N, S, 1).long()).squeeze(-1)
        val = v_all[:T * N].reshape(T, N, S)
        adv, ret = gae(rew.float(), val, v_all[T * N:], done, self.gamma, self.lambd)
        adv = adv.reshape(T * N, S)
        adv = (adv - adv.mean(0, keepdim=True)) / (adv.std(0, unbiased=False, keepdim=True) + 1e-8)
        return dict(obs=flat[:T * N], act=act.reshape(T * N, S).long(), logp=logp, adv=adv, ret=ret.reshape(T * N, S))

    def loss(self, mb):
        logits, v = self.net(mb['obs'])
        lp_all = torch.log_softmax(logits.float(), -1)
        lp = lp_all.gather(-1, mb['act'].unsqueeze(-1)).squeeze(-1)
        ratio = torch.exp(lp - mb['logp'])
        a = mb['adv']
        pg = -torch.minimum(ratio * a, torch.clamp(ratio, 1.0 - self.clip_eps, 1.0 + self.clip_eps) * a).mean(0)
        vf = ((v.float() - mb['ret']) ** 2).mean(0)
        # entropy over a signal's own actions: the -inf log-probabilities of padded actions are replaced BEFORE the
        # product (0 * -inf would put NaN into the gradient even behind a torch.where)
        ent = -(torch.exp(lp_all) * lp_all.masked_fill(~self.net.action_mask, 0.0)).sum(-1).mean(0)
        return (pg + self.value_coef * vf - self.entropy_coef * ent).sum()          # sum over signals

    def clip_grad_per_signal(self):
        """torch.nn.utils.clip_grad_norm_(agent parameters, max_grad_norm) for each of the S agents."""
        S = len(self.net.lanes)
        sq = torch.zeros(S, device=self.net.fc1_w.device)
        views = []
        for p_ in self.net.parameters():
            if p_.grad is None:
                continue
            g = p_.grad.reshape(S, -1)
            views.append(g)
            sq += (g.float() ** 2).sum(1)
        scale = torch.clamp(self.max_grad_norm / (sq.sqrt() + 1e-6), max=1.0)
        for g in views:
            g.mul_(scale.unsqueeze(1).to(g.dtype))

    def update(self, obs, act, rew, done, last_obs, generator=None):
        ds = self.make_dataset(obs, act, rew, done, last_obs)
        n = ds['act'].shape[0]
        last = None
        for _ in range(self.epochs):
            perm = torch.randperm(n, device=ds['act'].device, generator=generator)
            for i in range(0, n, self.minibatch):
                idx = perm[i:i + self.minibatch]
                mb = {k: v[idx] for k, v in ds.items()}
                self.opt.zero_grad(set_to_none=True)
                last = self.loss(mb)
                last.backward()
                self.clip_grad_per_signal()
                self.opt.step()
                self.n_updates += 1
        return last.detach()
