#!/usr/bin/env python3
"""env-steps/s of the BASELINE.json configurations at their per-GPU sizes (one GPU each).

  2: cologne1     x 1024 envs, MaxPressure on device      (reward pressure)
  3: ingolstadt21 x 4096 envs, seeded random on device    (bench.py's workload)
  4: cologne8     x 2048 envs (the per-GPU share of 16 384 / 8), MaxPressure on device
  5: ingolstadt21 x 1024 envs (per-GPU share of 8 192 / 8), full 360-step episode, fp16 observations;
     sim-only rate (the IDQN forward is measured separately by tools/idqn_rollout.py when present)
  1: cologne1 x 1, FIXED programme, host-stepped (plumbing): MultiSignal dict API, steps/s
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
from resco_amd.scenario import Scenario      # noqa: E402
from resco_amd.sim import BatchedSim         # noqa: E402


def run(name, n, policy, steps=360, warm=0, fixed=0, pipes=1):
    """`pipes` handles of n / pipes environments each, every one stepping on its own HIP stream (bench.py --pipes)"""
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    per = n // pipes
    sims = [BatchedSim(sc, per, seed=0, fixed_program=fixed, env_base=i * per, device_envs=n) for i in range(pipes)]

    def one(k):
        for sim in sims:
            if policy == 'maxpressure':
                sim.act_maxwave(1)
            elif policy == 'random':
                sim.act_random(k)
            sim.step(None)

    def sync():
        for sim in sims:
            sim.sync()

    for k in range(warm):
        one(k)
    sync()
    for sim in sims:
        sim.timing(True)
    t0 = time.perf_counter()
    for k in range(warm, warm + steps):
        one(k)
    sync()
    dt = time.perf_counter() - t0
    kms, nl = 0.0, 0
    for sim in sims:
        a, b = sim.timing_read()
        kms += a
        nl += b
    act = sum(float((sim.stats()['active_ticks'] / sim.stats()['ticks']).sum()) for sim in sims) / n
    info = sims[0].info()
    info['n_envs'] = n
    out = dict(map=name, envs=n, pipes=pipes, policy=policy, steps=steps, env_steps_per_s=n * steps / dt, ms_per_step=dt / steps * 1e3,
               kernel_ms=kms / max(1, nl), mean_active=act, **info)
    for sim in sims:
        sim.close()
    return out


def config1():
    from resco_amd import rewards, states
    from resco_amd.multi_signal import MultiSignal
    import tempfile
    env = MultiSignal('FIXED-tr0', 'cologne1', None, states.mplight, rewards.wait, yellow_length=3, end_time=28800,
                      log_dir=tempfile.mkdtemp() + os.sep, seed=0, fixed_program=True)
    env.reset()
    t0 = time.perf_counter()
    done, k = False, 0
    while not done:
        obs, rew, done, info = env.step({ts: 0 for ts in env.all_ts_ids})
        k += 1
    dt = time.perf_counter() - t0
    ts = env.trip_stats()
    env.close()
    return dict(map='cologne1', envs=1, policy='FIXED programme, MultiSignal dict API (host round trip per step)', steps=k,
                env_steps_per_s=k / dt, avg_delay=ts['avg_time_loss'] + ts['avg_depart_delay'])


if __name__ == '__main__':
    print(json.dumps(dict(config=1, **config1())), flush=True)
    for pipes in (1, 2):
        print(json.dumps(dict(config=2, **run('cologne1', 1024, 'maxpressure', pipes=pipes))), flush=True)
        print(json.dumps(dict(config=3, **run('ingolstadt21', 4096, 'random', pipes=pipes))), flush=True)
        print(json.dumps(dict(config=4, **run('cologne8', 2048, 'maxpressure', pipes=pipes))), flush=True)
        print(json.dumps(dict(config=5, **run('ingolstadt21', 1024, 'random', pipes=pipes))), flush=True)
    for n in (16384, 65536):
        print(json.dumps(dict(config='2x', **run('cologne1', n, 'maxpressure', pipes=2))), flush=True)
    print(json.dumps(dict(config='4x', **run('cologne8', 16384, 'maxpressure', pipes=2))), flush=True)
    print(json.dumps(dict(config='3x', **run('ingolstadt21', 16384, 'random', steps=120, pipes=2))), flush=True)
