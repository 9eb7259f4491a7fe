#!/usr/bin/env python3
"""A/B of the launch shape on one MI355X:

  * sim-only env-steps/s for (N environments, K pipes): K handles of N/K environments each (env_base keys the RNG, so the
    union is the same batch), every handle stepping on a HIP stream of its own -- the kernels of different pipes overlap, which
    fills the tail of a launch (4096 workgroups on 768 resident slots = 5.33 rounds) and the launch gaps;
  * config 5 (IDQN rollout): the fused policy kernel of pipe A runs under the step kernel of pipe B.  epsilon = 1 keeps the
    traffic identical to the random policy's (the policy kernel still evaluates all 21 networks), so the rate next to sim-only is
    the cost of the policy in the loop and nothing else; `sched` is the reference's linear epsilon schedule over the episode.

  python tools/pipes_ab.py [--map ingolstadt21] [--steps 40] [--out gpurun_out/r04/pipes_ab.jsonl]
"""
import argparse
import json
import os
import sys
import time

import torch

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario                        # noqa: E402
from resco_amd.sim import BatchedSim                           # noqa: E402


def make_pipes(sc, n, k, outputs, seed=0):
    assert n % k == 0
    per = n // k
    sims = [BatchedSim(sc, per, seed=seed, sigma=-1.0, speed_dev=1, env_base=i * per) for i in range(k)]
    for s in sims:
        s.set_outputs(outputs)
    streams = [torch.cuda.Stream() for _ in range(k)]
    return sims, streams


def sync_all(sims):
    for s in sims:
        s.sync()
    torch.cuda.synchronize()


def sim_only(sc, n, k, steps, start=170):
    sims, streams = make_pipes(sc, n, k, ('drq_norm', 'mplight'))
    ptr = [s.cuda_stream for s in streams]
    for j in range(start):
        for i, s in enumerate(sims):
            s.act_random(j, ptr[i]); s.step(None, ptr[i])
    sync_all(sims)
    st0 = [s.stats() for s in sims]
    t0 = time.perf_counter()
    for j in range(start, start + steps):
        for i, s in enumerate(sims):
            s.act_random(j, ptr[i]); s.step(None, ptr[i])
    sync_all(sims)
    dt = time.perf_counter() - t0
    st1 = [s.stats() for s in sims]
    act = sum(float((b['active_ticks'] - a['active_ticks']).sum()) for a, b in zip(st0, st1)) / (n * steps * 10.0)
    for s in sims:
        s.close()
    return dict(mode='sim_only', envs=n, pipes=k, steps=steps, env_steps_per_s=n * steps / dt, ms_per_step=dt / steps * 1e3, mean_active=act)


def rollout(sc, n, k, steps, eps_mode, start=170, prio=False):
    from resco_amd.agents.idqn_fused import FusedIDQN
    from resco_amd.agents.idqn_rollout import BatchedIDQN
    sims, streams = make_pipes(sc, n, k, ('drq_norm_f16',))
    ptr = [s.cuda_stream for s in streams]
    net = BatchedIDQN.from_scenario(sc, dtype=torch.float16, device='cuda')
    net.init_like_reference(seed=0)
    pol = [FusedIDQN(net, seed=7 + i) for i in range(k)]
    obs = [s.tensor('drq_norm_f16') for s in sims]
    act = [s.tensor('actions') for s in sims]
    sync_all(sims)

    if prio:        # the policy kernels on high-priority streams of their own, ordered with the pipe's step by events
        hi = [torch.cuda.Stream(priority=-1) for _ in range(k)]
        ev_act = [torch.cuda.Event() for _ in range(k)]
        ev_step = [torch.cuda.Event() for _ in range(k)]
        for i in range(k):
            ev_step[i].record(streams[i])

    def one(j, eps):
        for i, s in enumerate(sims):
            if prio:
                hi[i].wait_event(ev_step[i])
                pol[i].act(obs[i], epsilon=eps, step_key=j, stream=hi[i].cuda_stream, out=act[i])
                ev_act[i].record(hi[i])
                streams[i].wait_event(ev_act[i])
                s.step(None, ptr[i])
                ev_step[i].record(streams[i])
            else:
                pol[i].act(obs[i], epsilon=eps, step_key=j, stream=ptr[i], out=act[i])
                s.step(None, ptr[i])

    for j in range(start):
        one(j, 1.0)
    sync_all(sims)
    t0 = time.perf_counter()
    for j in range(start, start + steps):
        one(j, 1.0 if eps_mode == 'eps1' else max(0.0, 1.0 - j / (0.8 * 360)))
    sync_all(sims)
    dt = time.perf_counter() - t0
    for s in sims:
        s.close()
    return dict(mode='sim_plus_fused_policy_' + eps_mode + ('_prio' if prio else ''), envs=n, pipes=k, steps=steps, env_steps_per_s=n * steps / dt, ms_per_step=dt / steps * 1e3)


def group_run(sc, n, k, steps, agent, start=170, per_call=1):
    """the same two loops through rs_group_step: ONE call through ctypes per env-step (or per `per_call` env-steps) for all pipes"""
    from resco_amd.sim import SimGroup
    per = n // k
    idqn = agent == 'idqn'
    sims = [BatchedSim(sc, per, seed=0, sigma=-1.0, speed_dev=1, env_base=i * per) for i in range(k)]
    for s in sims:
        s.set_outputs(('drq_norm_f16',) if idqn else ('drq_norm', 'mplight'))
    grp = SimGroup(sims)
    kw = {}
    if idqn:
        from resco_amd.agents.idqn_fused import FusedIDQN
        from resco_amd.agents.idqn_rollout import BatchedIDQN
        net = BatchedIDQN.from_scenario(sc, dtype=torch.float16, device='cuda')
        net.init_like_reference(seed=0)
        pol = FusedIDQN(net, seed=7)
        kw = dict(policy=pol._h, epsilon=1.0, seed=7)
    grp.step('idqn' if idqn else 'random', step_key=0, n_steps=start, **kw)
    sync_all(sims)
    t0 = time.perf_counter()
    for j in range(start, start + steps, per_call):
        grp.step('idqn' if idqn else 'random', step_key=j, n_steps=min(per_call, start + steps - j), **kw)
    sync_all(sims)
    dt = time.perf_counter() - t0
    for s in sims:
        s.close()
    return dict(mode=('sim_plus_fused_policy_eps1' if idqn else 'sim_only') + '_group', envs=n, pipes=k, steps=steps, steps_per_call=per_call,
                env_steps_per_s=n * steps / dt, ms_per_step=dt / steps * 1e3)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--map', default='ingolstadt21')
    ap.add_argument('--steps', type=int, default=40)
    ap.add_argument('--out', default=None)
    ap.add_argument('--quick', action='store_true')
    ap.add_argument('--rollout-only', action='store_true')
    ap.add_argument('--no-rollout', action='store_true')
    ap.add_argument('--prio', action='store_true', help='rollout: policy kernels on high-priority streams')
    ap.add_argument('--group', action='store_true', help='only the rs_group_step runs (one ctypes call per step for all pipes)')
    a = ap.parse_args()
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', a.map + '.npz'))
    rows = []

    def emit(r):
        rows.append(r)
        print(json.dumps(r), flush=True)
        if a.out:
            with open(a.out, 'a') as f:
                f.write(json.dumps(r) + '\n')

    if a.group:
        for n in (1024, 4096):
            for k in (1, 2, 4, 8) + ((16,) if n == 1024 else ()):
                emit(sim_only(sc, n, k, a.steps))
                emit(group_run(sc, n, k, a.steps, 'random'))
            for k in (2, 4, 8):
                emit(rollout(sc, n, k, a.steps, 'eps1'))
                emit(group_run(sc, n, k, a.steps, 'idqn'))
                emit(group_run(sc, n, k, a.steps, 'idqn', per_call=10))
        return
    shapes = [(4096, 1), (4096, 2), (4096, 4), (3840, 1), (4608, 1), (4608, 2), (3072, 1), (8192, 1), (8192, 2)]
    if a.quick:
        shapes = shapes[:3]
    for n, k in ([] if a.rollout_only else shapes):
        emit(sim_only(sc, n, k, a.steps))
    for n in (() if a.no_rollout else (1024, 4096)):
        for k in (1, 2, 4):
            emit(sim_only(sc, n, k, a.steps))
        for k in (1, 2, 4):
            emit(rollout(sc, n, k, a.steps, 'eps1'))
            if a.prio:
                emit(rollout(sc, n, k, a.steps, 'eps1', prio=True))
        for k in (1, 2):
            emit(rollout(sc, n, k, a.steps, 'sched'))


if __name__ == '__main__':
    main()
