#!/usr/bin/env python3
"""IPPO training loop on the GPU: HIP simulator -> fp16 observations -> fused HIP policy kernel in sampling mode
(rs_idqn_act, mode 1) -> rollout segment in HBM -> batched PPO update (PyTorch) -> weights re-packed on the device.

    python tools/ippo_train.py [map] [n_envs] [episodes] [segment_steps] [minibatches_per_epoch]

Prints one JSON line per episode (average trip delay as utils/readXML.py computes it, env-steps/s including
learning).  The reference's IPPO learns far more slowly than its IDQN (1400 published episodes); this tool shows the
machinery end to end, not a converged policy.  Random-init weights, synthetic (rou.xml) demand."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.agents.idqn_fused import FusedIDQN                   # noqa: E402
from resco_amd.agents.ippo import BatchedIPPO, BatchedPPOLearner      # noqa: E402
from resco_amd.multi_signal import VecMultiSignal                     # noqa: E402


def main(map_name='cologne1', n=256, episodes=20, seg=30, mbs=4):
    env = VecMultiSignal(map_name, n, states=('drq_norm_f16',), rewards=('wait_norm',), seed=0)
    S, steps = env.n_signals, env.horizon_steps
    net = BatchedIPPO.from_scenario(env.scenario, dtype=torch.float32, device='cuda')
    net.init_like_reference(seed=0)
    learner = BatchedPPOLearner(net, minibatch=max(256, seg * n // mbs))
    policy = FusedIDQN(net, seed=3)
    policy.refresh_on_device()
    actions = env.tensor('actions')
    obs_buf = torch.zeros(seg, n, S, net.lmax, 5, dtype=torch.float16, device='cuda')
    act_buf = torch.zeros(seg, n, S, dtype=torch.int32, device='cuda')
    rew_buf = torch.zeros(seg, n, S, dtype=torch.float32, device='cuda')
    done_buf = torch.zeros(seg, dtype=torch.bool, device='cuda')
    gen = torch.Generator(device='cuda').manual_seed(0)
    t_global, i = 0, 0
    for ep in range(episodes):
        env.sim.set_seed(1000 + ep)
        obs = env.reset()['drq_norm_f16']
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            obs_buf[i].copy_(obs)
            policy.act(obs, step_key=t_global, out=actions, sample=True)
            o, r, done, _ = env.step(None)
            act_buf[i].copy_(actions)
            rew_buf[i].copy_(r['wait_norm'])
            done_buf[i] = bool(done)
            obs = o['drq_norm_f16']
            t_global += 1
            i += 1
            if i == seg:
                learner.update(obs_buf, act_buf, rew_buf, done_buf, obs, generator=gen)
                policy.refresh_on_device()
                i = 0
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        print(json.dumps(dict(episode=ep, avg_delay_s=round(float(env.sim.trip_delay().mean()), 2),
                              arrived_per_env=round(float(env.sim.stats()['arrived'].mean()), 1), adam_steps=learner.n_updates,
                              env_steps_per_s=round(n * steps / dt), ms_per_step=round(dt / steps * 1e3, 3))), flush=True)
    env.close()


if __name__ == '__main__':
    a = sys.argv[1:]
    main(a[0] if len(a) > 0 else 'cologne1', int(a[1]) if len(a) > 1 else 256, int(a[2]) if len(a) > 2 else 20,
         int(a[3]) if len(a) > 3 else 30, int(a[4]) if len(a) > 4 else 4)
