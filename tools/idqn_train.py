#!/usr/bin/env python3
"""IDQN training loop entirely on the GPU: HIP simulator -> fp16 observations -> fused HIP policy kernel
(rs_idqn_act) -> device replay ring -> batched DQN update (PyTorch) -> weights re-packed on the device.  Nothing crosses PCIe per step except the launch calls.

    python tools/idqn_train.py [map] [n_envs] [episodes] [batch] [updates_per_step] [graph|nograph] [replay_steps] [eps_end] [seed]

Prints one JSON line per episode (mean episode return of rewards.wait_norm per signal, average trip delay as
utils/readXML.py computes it, epsilon, env-steps/s including learning) and a final line comparing with the
on-device random policy on the same demand.  Random-init weights, synthetic (rou.xml) demand."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.agents.idqn_learn import BatchedDQNLearner, DeviceReplay, linear_epsilon      # noqa: E402
from resco_amd.agents.idqn_fused import FusedIDQN                                           # noqa: E402
from resco_amd.agents.idqn_rollout import BatchedIDQN                                       # noqa: E402
from resco_amd.multi_signal import VecMultiSignal                                           # noqa: E402


def delay(env):
    """Mean over the environments of utils/readXML.py's episode figure (timeLoss + departDelay per trip)."""
    return float(env.sim.trip_delay().mean()), float(env.sim.stats()['arrived'].mean())


def main(map_name='cologne1', n=256, episodes=12, batch=256, updates=1, use_graph=True, replay_steps=0, eps_end=0.0, evaluate=True, quiet=False, seed=0, tls_expiry=True):
    rows = []
    env = VecMultiSignal(map_name, n, states=('drq_norm_f16',), rewards=('wait_norm',), seed=0, tls_expiry=tls_expiry)
    S, steps = env.n_signals, env.horizon_steps
    net = BatchedIDQN.from_scenario(env.scenario, dtype=torch.float32, device='cuda')
    net.init_like_reference(seed=seed)
    learner = BatchedDQNLearner(net, gamma=0.99, lr=1e-3, target_update=500, batch_size=batch)
    policy = FusedIDQN(net, seed=7 + seed)             # acting: one HIP kernel; weights re-packed on the device after each update
    actions = env.tensor('actions')
    # the reference keeps the last 10 000 transitions of its ONE environment = 27.8 episodes of history (pfrl_dqn.py:55); a ring of
    # `replay_steps` env-steps over all N environments (0: the last four episodes -- a small ring forgets exploratory data within
    # four episodes of epsilon reaching 0)
    replay = DeviceReplay(replay_steps if replay_steps > 0 else min(2048, 4 * steps), n, S, net.lmax, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(seed)
    decay = int(0.8 * episodes * steps)                 # the reference decays over config['steps'] agent steps

    env.sim.set_seed(12345)                             # baseline: random policy on an evaluation demand seed
    env.reset()
    for k in range(steps):
        env.act_random(k)
        env.step(None)
    rnd_delay, _ = delay(env)

    for ep in range(episodes):
        env.sim.set_seed(1000 + ep + 7919 * seed)
        obs = env.reset()['drq_norm_f16']
        ret = torch.zeros(n, S, device='cuda')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            eps = linear_epsilon(learner.t, 1.0, eps_end, decay)
            policy.act(obs, epsilon=eps, step_key=learner.t, out=actions)
            replay.stage(obs)
            o, r, done, _ = env.step(None)
            rew = r['wait_norm']
            replay.commit(actions, rew, done)
            ret += rew
            if use_graph and learner.n_updates == 0 and getattr(learner, '_graph', None) is None and len(replay) >= batch:
                learner.capture_update(replay)      # [sample -> loss -> backward -> Adam] as one HIP graph
            if learner.observe_step(replay, gen, updates) is not None:
                policy.refresh_on_device()
            obs = o['drq_norm_f16']
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        d, arrived = delay(env)
        rows.append(dict(episode=ep, epsilon=round(eps, 3), mean_return=float(ret.sum(1).mean()) / S,
                         avg_delay_s=round(d, 2), arrived_per_env=round(arrived, 1), updates=learner.n_updates,
                         env_steps_per_s=round(n * steps / dt), ms_per_step=round(dt / steps * 1e3, 3)))
        if not quiet:
            print(json.dumps(rows[-1]), flush=True)

    # greedy evaluation with the final weights (no further learning): on the baseline's seed and on the seed the last
    # training episode used - DQN keeps adapting within an episode, so a frozen copy can do worse than the last
    # training episodes did
    evals = {12345: None, 1000 + episodes - 1: None}
    for seed in ((12345, 1000 + episodes - 1) if evaluate else ()):
        env.sim.set_seed(seed)
        obs = env.reset()['drq_norm_f16']
        for k in range(steps):
            policy.act(obs, out=actions)
            o, _, _, _ = env.step(None)
            obs = o['drq_norm_f16']
        evals[seed] = round(delay(env)[0], 2)
    final = dict(map=map_name, envs=n, episodes=episodes, batch=batch, updates_per_step=updates, replay_steps=replay.T,
                 best_training_episode_delay_s=min(r['avg_delay_s'] for r in rows),
                 greedy_avg_delay_s=evals[12345], greedy_on_last_training_seed_s=evals[1000 + episodes - 1],
                 random_avg_delay_s=round(rnd_delay, 2))
    if not quiet:
        print(json.dumps(final))
    env.close()
    return rows, final


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0] if len(a) > 0 else 'cologne1', int(a[1]) if len(a) > 1 else 256, int(a[2]) if len(a) > 2 else 12,
         int(a[3]) if len(a) > 3 else 256, int(a[4]) if len(a) > 4 else 1, (a[5] != 'nograph') if len(a) > 5 else True,
         int(a[6]) if len(a) > 6 else 0, float(a[7]) if len(a) > 7 else 0.0, seed=int(a[8]) if len(a) > 8 else 0)
