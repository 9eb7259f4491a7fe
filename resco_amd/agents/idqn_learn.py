"""Device-resident replay ring and batched DQN learner for the IDQN rollout path ("+ replay write").

The reference wraps one ``pfrl.agents.DQN`` per signal (resco_benchmark/agents/pfrl_dqn.py:49-80):

    Adam() (lr 1e-3) - ReplayBuffer(10000) - LinearDecayEpsilonGreedy(EPS_START, EPS_END, steps)
    - minibatch BATCH_SIZE (32), replay_start_size BATCH_SIZE, target_update_interval TARGET_UPDATE (500),
    gamma GAMMA (0.99)                                       (resco_benchmark/config/agent_config.py:83-94)

and PFRL's DQN update is the published one:  y = Q(s)[a],  t = r + gamma (1 - done) max_a' Q_target(s')[a'],
loss = mean Huber(y - t, delta 1), hard target copy every ``target_update_interval`` agent steps.
PFRL is a third-party dependency that is not vendored in the reference tree and is not installed in this image,
so this module restates that published update and is checked against an unbatched PyTorch fp32 restatement of it
built from ``reference_q_network`` (tests/test_idqn_learn.py) - not against PFRL itself.

What is different by design: the S per-signal agents are ONE set of stacked parameters (BatchedIDQN) trained
with one Adam over the sum of the per-signal losses.  Adam is element-wise and the per-signal parameters are
disjoint, so this equals S independent Adam optimisers step for step.  Transitions of all N lock-step
environments are written to one ring in HBM (no host copy); every signal draws its own minibatch.
"""
import copy

import torch

from .idqn_rollout import BatchedIDQN


class DeviceReplay:
    """Ring over the last ``capacity_steps`` env-steps of all N environments, resident on the GPU.

    Slot i holds (obs_i, act_i, rew_i, done_i): the observation the agents saw, what they did, and the reward /
    episode end that followed.  The successor observation of slot i is the obs of slot i+1, so a transition is
    sampled as (slot i, slot i+1) with i+1 already written; ``done_i`` cuts the bootstrap at episode ends (the
    post-reset observation stored in slot i+1 is then never used as a successor).

    Sizes for ingolstadt21 (S=21, Lmax=17): obs fp16 3 570 B per env-step -> N=4096 is 14.6 MB per slot,
    512 slots = 7.5 GB of the 288 GB HBM."""

    def __init__(self, capacity_steps, n_envs, n_signals, lmax, device='cuda', obs_dtype=torch.float16):
        T, N, S = int(capacity_steps), int(n_envs), int(n_signals)
        assert T >= 2
        self.T, self.N, self.S = T, N, S
        self.obs = torch.zeros(T, N, S, lmax, 5, dtype=obs_dtype, device=device)
        self.act = torch.zeros(T, N, S, dtype=torch.int16, device=device)
        self.rew = torch.zeros(T, N, S, dtype=torch.float32, device=device)
        self.done = torch.zeros(T, dtype=torch.bool, device=device)
        self.head = 0           # next slot to write
        self.count = 0          # slots written so far (saturates at T)
        self._sig = torch.arange(S, device=device)
        self._pos = torch.zeros(2, dtype=torch.int64, device=device)     # [head, count] mirrored on the device (graph replay)

    def __len__(self):
        """Number of complete transitions that can be sampled (per signal)."""
        return max(0, self.count - 1) * self.N

    def stage(self, obs):
        """Copy the observation the agents are about to act on (the simulator's zero-copy output tensor is
        overwritten by the next step, so this must happen before env.step)."""
        self.obs[self.head].copy_(obs)

    def commit(self, act, rew, done):
        """Complete the staged slot with the action taken and the reward / episode end that followed."""
        i = self.head
        self.act[i].copy_(act)
        self.rew[i].copy_(rew)
        self.done[i] = bool(done)
        self.head = (i + 1) % self.T
        self.count = min(self.count + 1, self.T)
        self._pos[0] = self.head
        self._pos[1] = self.count

    def push(self, obs, act, rew, done):
        """obs [N,S,L,5], act [N,S] int, rew [N,S] float, done bool (lock-step: one flag for all envs)."""
        self.stage(obs)
        self.commit(act, rew, done)

    def sample(self, batch_size, generator=None):
        """Independent minibatch per signal: returns o [B,S,L,5], a [B,S] long, r [B,S], o2 [B,S,L,5], d [B,S] float."""
        n_ok = self.count - 1                       # slots with a written successor
        assert n_ok >= 1, 'need two pushes before sampling'
        dev = self.obs.device
        B, S = int(batch_size), self.S
        # the oldest valid slot is head - count; the newest slot (head - 1) has no successor yet
        k = torch.randint(0, n_ok, (B, S), device=dev, generator=generator)
        t = (self.head - self.count + k) % self.T
        e = torch.randint(0, self.N, (B, S), device=dev, generator=generator)
        t2 = (t + 1) % self.T
        sig = self._sig
        return (self.obs[t, e, sig], self.act[t, e, sig].long(), self.rew[t, e, sig], self.obs[t2, e, sig],
                self.done[t].to(torch.float32))


def _sample_from_device_position(rp, batch_size):
    """DeviceReplay.sample with the ring position read from device memory and PyTorch's default generator: every value
    that changes between calls lives on the device, so the call can be captured in a HIP graph and replayed."""
    dev = rp.obs.device
    B, S = int(batch_size), rp.S
    head, count = rp._pos[0], rp._pos[1]
    n_ok = torch.clamp(count - 1, min=1)
    k = (torch.rand(B, S, device=dev) * n_ok).long().clamp_(max=rp.T - 2)
    k = torch.minimum(k, n_ok - 1)
    t = torch.remainder(head - count + k, rp.T)
    e = torch.randint(0, rp.N, (B, S), device=dev)
    t2 = torch.remainder(t + 1, rp.T)
    sig = rp._sig
    return (rp.obs[t, e, sig], rp.act[t, e, sig].long(), rp.rew[t, e, sig], rp.obs[t2, e, sig], rp.done[t].to(torch.float32))


def linear_epsilon(t, start, end, decay_steps):
    """pfrl.explorers.LinearDecayEpsilonGreedy.compute_epsilon (pfrl_dqn.py:62-67 passes config['steps'])."""
    if t >= decay_steps:
        return float(end)
    return float(start + (end - start) * (t / float(decay_steps)))


class BatchedDQNLearner:
    """DQN update for the S stacked per-signal networks of a BatchedIDQN."""

    def __init__(self, qnet, gamma=0.99, lr=1e-3, target_update=500, batch_size=32):
        assert isinstance(qnet, BatchedIDQN)
        self.q = qnet
        self.target = copy.deepcopy(qnet)
        for p in self.target.parameters():
            p.requires_grad_(False)
        self.gamma, self.batch_size, self.target_update = float(gamma), int(batch_size), int(target_update)
        self.opt = torch.optim.Adam(self.q.parameters(), lr=lr)
        self.t = 0              # agent steps seen (PFRL's self.t)
        self.n_updates = 0
        # padded lanes feed relu(conv bias) into fc1: their rows are zero at load and must stay zero
        H = qnet.lmax - 1
        mask = torch.zeros(len(qnet.lanes), 64, H, 4, dtype=qnet.fc1_w.dtype, device=qnet.fc1_w.device)
        for s, l in enumerate(qnet.lanes):
            mask[s, :, :l - 1] = 1.0
        self._fc1_mask = mask.reshape(len(qnet.lanes), 64 * H * 4, 1)
        qnet.fc1_w.register_hook(lambda g: g * self._fc1_mask)

    def loss(self, o, a, r, o2, d):
        """Sum over signals of the mean Huber loss of each signal's minibatch.  Inputs as DeviceReplay.sample()
        returns them, but signal-major batches ([B,S,...]) are read as 'B independent draws per signal'."""
        q = self.q(o)                                               # [B, S, Amax]
        y = q.gather(-1, a.unsqueeze(-1)).squeeze(-1)               # [B, S]
        with torch.no_grad():
            nxt = self.target(o2).max(dim=-1).values                # padded actions are -inf: never the max
            tgt = r + self.gamma * (1.0 - d) * nxt
        per = torch.nn.functional.smooth_l1_loss(y.float(), tgt.float(), reduction='none')   # Huber, delta 1
        return per.mean(dim=0).sum()

    def update(self, batch):
        self.opt.zero_grad(set_to_none=True)
        loss = self.loss(*batch)
        loss.backward()
        self.opt.step()
        self.n_updates += 1
        return loss.detach()

    def capture_update(self, replay, warmup=3):
        """Capture [sample a minibatch per signal -> loss -> backward -> Adam step] as ONE HIP graph.  The ~150 small
        kernels of an update are launch-bound (S = 21 networks x batch 256 is little work each); replaying the graph
        removes the launch gaps.  Needs >= 2 slots in the ring and an Adam created with capturable=True (done here,
        optimiser state is carried over by re-creating it before any update has run)."""
        assert self.n_updates == 0 and len(replay) >= 1
        self.opt = torch.optim.Adam(self.q.parameters(), lr=self.opt.param_groups[0]['lr'], capturable=True)
        self._graph_replay = replay

        def one():
            self.opt.zero_grad(set_to_none=False)
            loss = self.loss(*_sample_from_device_position(replay, self.batch_size))
            loss.backward()
            self.opt.step()
            return loss.detach()

        state = {k: v.detach().clone() for k, v in self.q.state_dict().items()}
        side = torch.cuda.Stream()
        side.wait_stream(torch.cuda.current_stream())
        with torch.cuda.stream(side):
            for p_ in self.q.parameters():
                p_.grad = torch.zeros_like(p_)
            for _ in range(warmup):
                one()
        torch.cuda.current_stream().wait_stream(side)
        self._graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(self._graph):
            self._graph_loss = one()
        # the warm-up / capture iterations were real updates: rewind weights and optimiser to where they started
        self.q.load_state_dict(state)
        for st_ in self.opt.state.values():
            for v in st_.values():
                if torch.is_tensor(v):
                    v.zero_()
        return self

    def sync_target(self):
        self.target.load_state_dict(self.q.state_dict())

    def observe_step(self, replay, generator=None, updates=1):
        """One agent step of PFRL's DQN.observe(): count it, update once the buffer holds a minibatch, and copy
        the target network every ``target_update`` steps."""
        self.t += 1
        out = None
        if self.t % self.target_update == 0:        # PFRL's DQN.batch_observe_train syncs the target BEFORE the replay update
            self.sync_target()
        if len(replay) >= self.batch_size:
            for _ in range(updates):
                if getattr(self, '_graph', None) is not None and replay is self._graph_replay:
                    self._graph.replay()
                    self.n_updates += 1
                    out = self._graph_loss
                else:
                    out = self.update(replay.sample(self.batch_size, generator))
        return out
