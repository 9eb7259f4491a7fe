#!/usr/bin/env python3
"""BASELINE config 5 (per-GPU share): ingolstadt21 x N envs, full 360-step episode, IDQN epsilon-greedy rollout
on the fp16 observation tensor; reports sim-only, sim + PyTorch policy and sim + fused HIP policy (rs_idqn_act) rates.  Random-init weights (no checkpoints
offline), rewards.wait_norm collected on device."""
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.agents.idqn_fused import FusedIDQN              # noqa: E402
from resco_amd.agents.idqn_rollout import BatchedIDQN          # noqa: E402
from resco_amd.multi_signal import VecMultiSignal              # noqa: E402


def main(n=1024, steps=360, dtype=torch.float16):
    env = VecMultiSignal('ingolstadt21', n, states=('drq_norm_f16',), rewards=('wait_norm',), seed=0)
    net = BatchedIDQN.from_scenario(env.scenario, dtype=dtype, device='cuda')
    net.init_like_reference(seed=0)
    for _ in range(5):          # rocBLAS / allocator warm-up outside the timed regions
        net.act(env.reset()['drq_norm_f16'], epsilon=0.5)
    torch.cuda.synchronize()
    out = {}
    fused = FusedIDQN(net)
    act_buf = env.tensor('actions')
    for mode in ('sim_only', 'sim_plus_policy', 'sim_plus_fused_policy'):
        obs = env.reset()['drq_norm_f16']
        ret = torch.zeros(n, env.n_signals, device='cuda')
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for k in range(steps):
            if mode == 'sim_only':
                env.act_random(k)
                o, r, done, _ = env.step(None)
            elif mode == 'sim_plus_fused_policy':       # one HIP kernel for the 21 networks + epsilon-greedy, writing
                fused.act(obs, epsilon=max(0.0, 1.0 - k / (0.8 * steps)), step_key=k, out=act_buf)     # the simulator's action buffer
                o, r, done, _ = env.step(None)
                ret += r['wait_norm']
            else:
                eps = max(0.0, 1.0 - k / (0.8 * steps))
                a = net.act(obs, epsilon=eps)
                o, r, done, _ = env.step(a)
                ret += r['wait_norm']
            obs = o['drq_norm_f16']
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
        out[mode] = dict(env_steps_per_s=n * steps / dt, ms_per_step=dt / steps * 1e3)
    out['mean_return'] = float(ret.mean())
    out.update(envs=n, steps=steps, dtype=str(dtype), obs_shape=list(obs.shape))
    print(json.dumps(out))
    env.close()


if __name__ == '__main__':
    main(int(sys.argv[1]) if len(sys.argv) > 1 else 1024)
