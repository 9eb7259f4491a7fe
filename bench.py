#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused MultiSignal.step() kernel on MI355X.

Workload (BASELINE.json config 3, the configuration the headline metric is quoted on): ingolstadt21
(21 signals, 163 observed lanes, 4 283 trips / 3600 s) x 4096 lock-step environments PER GPU, bench mode
(Krauss sigma 0.5, per-vehicle speedFactor), on-device seeded random policy (STOCHASTIC analogue), every
step producing lane aggregates + drq_norm + mplight + wave + wait + wait_norm + pressure.
One "step" = one MultiSignal.step() of every environment = 10 one-second simulation ticks, fused in ONE
kernel launch.  Warm-up + timed steps walk through the 360-step episode (defaults: 60 + 300 = one episode);
when an episode ends inside the timed region the reset is part of the timed work.

  python bench.py                               # 1 GPU
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W # N GPUs: env-batch split, no collective on the data path

Rank 0 prints ONE JSON line.  `roofline` prices the step kernel against HBM (this path has no dense
contraction, MFMA is irrelevant); `cpu_baseline` is the C oracle (oracle/, test infrastructure) timed on the
host cores of the same box on a bounded sample.
"""
import argparse
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, ROOT)

HBM_PEAK_GBS = 8000.0            # /opt/skills/guides/MI355X_MICROARCH.md: HBM3E 8.0 TB/s spec (6.29 TB/s measured copy)
EPISODE_STEPS = 360


def shard(rank, world, envs_per_gpu):
    """weak scaling: every rank owns `envs_per_gpu` environments; global env index keys the RNG"""
    return rank * envs_per_gpu, envs_per_gpu


def algorithmic_bytes_per_env_step(sc, mean_active):
    """Compulsory HBM traffic of one env-step (DESIGN.md 'Algorithmic bytes'): the environment slab in and
    out once, actions in, observation / reward buffers out."""
    S, O = sc.n_signals, sc.n_obs
    lmax = int((sc.sig_obs_start[1:] - sc.sig_obs_start[:-1]).max())
    per_vehicle = 26 + 26 + 8          # slab read, slab write, trip_route/trip_vtype gather
    per_signal = 4 + 12 + 12 + 124     # action, TLS state r/w, phase/mplight/wave/wait/wait_norm/pressure/queue_*
    return mean_active * per_vehicle + S * per_signal + O * 40 + S * lmax * 10 + 24 + 80


def pmc_traffic_bytes_per_launch():
    """HBM bytes per launch of rs_step_kernel from the committed rocprofv3 PMC passes (separate --pmc runs of
    this same workload, profiles/r01_final_pmc_summary.json): 2 x FETCH_SIZE (the gfx950 half-count correction of
    MI355X_MICROARCH.md, HBM section) + WRITE_SIZE, both reported in KiB.  None when no summary is committed."""
    path = os.path.join(ROOT, 'profiles', 'r01_final_pmc_summary.json')
    try:
        with open(path) as f:
            c = json.load(f)['counters']
        return (2.0 * c['FETCH_SIZE']['per_launch_avg'] + c['WRITE_SIZE']['per_launch_avg']) * 1024.0
    except Exception:
        return None


def run_timed(sim, steps, warmup, barrier, sync, reduce_max):
    """W untimed steps, then exactly K timed steps bracketed by barrier + device sync; max over ranks."""
    k = 0

    def one():
        nonlocal k
        if k > 0 and k % EPISODE_STEPS == 0:
            sim.reset()
        sim.act_random(k)
        sim.step(None)
        k += 1

    for _ in range(warmup):
        one()
    sync()
    st0 = sim.stats()
    sim.timing(True)
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    sync()
    barrier()
    t1 = time.perf_counter()
    kernel_ms, launches = sim.timing_read()
    sim.timing(False)
    st1 = sim.stats()
    elapsed = reduce_max(t1 - t0)
    return elapsed, kernel_ms, launches, st0, st1


def effective_cores():
    """host cores this process may really use: affinity mask, capped by the cgroup CPU quota"""
    n = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        quota, period = open('/sys/fs/cgroup/cpu.max').read().split()[:2]
        if quota != 'max':
            n = min(n, max(1, int(int(quota) / int(period))))
    except Exception:
        try:
            q = int(open('/sys/fs/cgroup/cpu/cpu.cfs_quota_us').read())
            p = int(open('/sys/fs/cgroup/cpu/cpu.cfs_period_us').read())
            if q > 0:
                n = min(n, max(1, q // p))
        except Exception:
            pass
    return n


def cpu_baseline(sc, seed, budget_s=15.0):
    """C oracle on the host cores: `cores` processes, each running whole 360-step episodes of one
    environment with the same random-policy hash the device uses."""
    import multiprocessing as mp
    cores = effective_cores()
    from oracle.pyoracle import build
    build()
    t0 = time.perf_counter()
    _cpu_worker((0, 1, seed, 24))
    probe = (time.perf_counter() - t0) / 24.0          # seconds per env-step on one core
    per_worker = max(60, min(EPISODE_STEPS * 4, int(budget_s / max(probe, 1e-6))))
    jobs = [(i, 1, seed, per_worker) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context('fork').Pool(cores) as pool:
        done = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    total = sum(done)
    return dict(value=total / wall, unit='env-steps/s', cores=cores, kind='port', single_thread_value=1.0 / probe,
                sample='C oracle (oracle/resco_oracle.c, gcc -O2, scalar): %d processes (affinity %d, cgroup quota applied) x %d '
                       'env-steps of ingolstadt21 from episode start, same hashed random policy, %.1f s wall'
                       % (cores, len(os.sched_getaffinity(0)), per_worker, wall))


def _cpu_worker(job):
    env_index, n, seed, steps = job
    import numpy as np
    from oracle.pyoracle import OracleEnv, lib
    from resco_amd.scenario import Scenario
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', 'ingolstadt21.npz'))
    L = lib()
    env = OracleEnv(sc, env_index=env_index, seed=seed, sigma=-1.0, speed_dev=1)
    G = [int(g) for g in sc.tls_ngreen]
    for k in range(steps):
        if k > 0 and k % EPISODE_STEPS == 0:
            env.reset()
        a = np.array([L.orc_hash(seed ^ 0xA5A5A5A5, env_index, s, k, 7) % G[s] for s in range(sc.n_signals)], np.int32)
        env.step(a)
    return steps


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=60)
    ap.add_argument('--map', default='ingolstadt21')
    ap.add_argument('--envs', type=int, default=4096, help='environments per GPU')
    ap.add_argument('--block', type=int, default=0, help='threads per workgroup (0 = library default)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    args = ap.parse_args()

    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    torch.cuda.set_device(local)
    dist = None
    if world > 1 or os.environ.get('RESCO_BENCH_FORCE_DIST') == '1':      # the env var exercises the RCCL path at N=1
        import torch.distributed as dist
        dist.init_process_group('nccl', rank=rank, world_size=world)      # RCCL: barrier + one MAX only

    from resco_amd.scenario import Scenario
    from resco_amd.sim import BatchedSim
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', args.map + '.npz'))
    env_base, n_local = shard(rank, world, args.envs)
    sim = BatchedSim(sc, n_local, device=local, seed=args.seed, sigma=-1.0, speed_dev=1, env_base=env_base,
                     block_threads=args.block)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        sim.sync()
        torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device='cuda')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed, kernel_ms, launches, st0, st1 = run_timed(sim, args.steps, args.warmup, barrier, sync, reduce_max)
    ticks = (st1['ticks'] - st0['ticks']).astype('float64')
    mean_active = float(((st1['active_ticks'] - st0['active_ticks']) / ticks.clip(min=1)).mean()) if ticks.min() > 0 \
        else float(st1['active'].mean())
    info = sim.info()
    env_steps = world * n_local * args.steps
    value = env_steps / elapsed
    b_alg = algorithmic_bytes_per_env_step(sc, mean_active)
    k_avg_s = (kernel_ms / max(1, launches)) * 1e-3
    achieved = b_alg * n_local / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
    out = {
        'metric': 'env-steps/sec', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s x %d lock-step envs per GPU (BASELINE config 3), fixed demand from the map\'s '
                               'rou.xml, on-device seeded random policy, Krauss sigma 0.5 + speedFactor dev 0.1'
                               % (args.map, n_local),
                   'map': args.map, 'envs_per_gpu': n_local, 'ticks_per_env_step': 10,
                   'block_threads': info['block_threads'], 'lds_bytes_per_env': info['lds_bytes'],
                   'parallelism': 'env-batch split x%d, no collective on the data path' % world},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS,
                     'traffic': pmc_traffic_bytes_per_launch() if (args.map == 'ingolstadt21' and n_local == 4096) else None,
                     'traffic_note': 'bytes per launch, rocprofv3 --pmc FETCH_SIZE / WRITE_SIZE passes of this workload '
                                     '(profiles/r01_final_pmc_summary.json, steps 60..160 of the episode); algorithmic bytes per '
                                     'launch = algorithmic_bytes_per_env_step x env_steps_per_launch',
                     'kernel': 'rs_step_kernel', 'kernel_avg_ms': k_avg_s * 1e3, 'launches': launches,
                     'algorithmic_bytes_per_env_step': b_alg, 'env_steps_per_launch': n_local,
                     'note': 'state is Infinity-Cache resident and the kernel is issue/latency bound: the HBM '
                             'fraction is small by construction (SURVEY.md 8d)'},
        'mean_active_vehicles_per_env': mean_active,
        'sim_ticks_per_s': value * 10, 'vehicle_ticks_per_s': value * 10 * mean_active,
    }
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(sc, args.seed)
                import shutil
                have_sumo = shutil.which('sumo') is not None
                try:
                    import libsumo  # noqa: F401
                    have_sumo = True
                except Exception:
                    pass
                # a SUMO / libsumo timing is reported only when SUMO exists on the box
                out['sumo_baseline'] = 'not measured (SUMO found but no runner shipped)' if have_sumo else \
                    'SUMO unavailable on this host'
            except Exception as e:          # the baseline must never take the GPU number down with it
                out['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}
        print(json.dumps(out), flush=True)
    sim.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()


if __name__ == '__main__':
    main()
