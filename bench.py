#!/usr/bin/env python3
"""bench.py -- env-steps/s of the fused MultiSignal.step() kernel on MI355X.

Workload (BASELINE.json config 3, the configuration the headline metric is quoted on): ingolstadt21
(21 signals, 163 observed lanes, 4 283 trips / 3600 s) x 4096 lock-step environments PER GPU, bench mode
(Krauss sigma 0.5, per-vehicle speedFactor), on-device seeded random policy (STOCHASTIC analogue), every
step producing the per-lane drq_norm rows + mplight + wait + wait_norm + pressure (config 3's state and reward functions;
the other derived buffers are switched off with rs_set_outputs).
One "step" = one MultiSignal.step() of every environment = 10 one-second simulation ticks, fused in ONE
kernel launch per PIPE: the batch of a GPU is split into --pipes (default 2) handles of envs / pipes environments, each
stepping on a HIP stream of its own (the global environment index keys the RNG, so the union is the same batch whatever the
split).  The launches of different pipes overlap, which fills the tail of a launch -- 4096 workgroups on 1024 resident slots
are 4 rounds that do not end together -- and the gaps between dependent launches: +10 % on one MI355X (profiles/r06_pipes_ab.jsonl).  The timed window is placed in the BULK of the 360-step episode whatever --steps / --warmup are: an
untimed fast-forward first rolls the batch to step 180 - K/2 - W (the demand ramps up over the hour, so the first steps
of an episode are a nearly empty network); then W untimed warm-up steps, then exactly K timed steps.  The defaults
(W = 60, K = 300) time steps 60..360; the driver's short run (W = 5, K = 20) times steps 170..190, whose load is within
a few per cent of the episode mean.  When an episode ends inside the timed region the reset is part of the timed work.

  python bench.py                               # 1 GPU
  python bench.py --gpus N                      # N GPUs of this node: starts N ranks itself (one per device, RCCL rendezvous on 127.0.0.1)
  python -m torch.distributed.run --nnodes=1 --nproc-per-node N --master-addr 127.0.0.1 --master-port P \
         bench.py --gpus N --steps K --warmup W # the same under a launcher; a launcher whose world differs from --gpus is refused
                                                # (env-batch split, no collective on the data path; `rccl_ranks` = the world RCCL reports)

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
OUTPUTS = ('drq_norm', 'mplight')
DEFAULT_PIPES = 2
PER_PIPE_CALLS = os.environ.get('RESCO_BENCH_PER_PIPE_CALLS') == '1'


def shard(rank, world, envs_per_gpu):
    """weak scaling: every rank owns `envs_per_gpu` environments; global env index keys the RNG"""
    return rank * envs_per_gpu, envs_per_gpu


def algorithmic_bytes_per_env_step(sc, mean_active):
    """compulsory HBM traffic of one env-step = state in + state out once (the ticks in between need no
    HBM) + actions in + outputs out:  B = V*(24 r + 24 w + 12 static r) + S*(4 action + 4 FSM r + 4 FSM w) + SL*5*s_out
    + S*4*n_rew, with s_out = 4 (fp32 lane rows) and n_rew = 2 reward vectors.  Derived state vectors (mplight, wave,
    drq_norm, the fp16 tensor ...) are recomputable from the SL*5 lane aggregates and are not counted."""
    S, O = sc.n_signals, sc.n_obs
    return mean_active * 60.0 + S * 12.0 + O * 5 * 4.0 + S * 4 * 2.0


def designed_bytes_per_env_step(sc, mean_active):
    """what the kernel moves per env-step BY DESIGN: the slab fields it loads / stores once, the
    HBM-resident private field it touches every tick (tloss; L2 hits after the first), every output buffer"""
    S, O = sc.n_signals, sc.n_obs
    lmax = int((sc.sig_obs_start[1:] - sc.sig_obs_start[:-1]).max())
    # slab fields in and out once (pos, speed, lane, trip, cursor, waiting time); every tick: time loss (read + write).  (Round 6: the
    # cooperation mailboxes are read only by the ~5 % of the plans whose record is flagged; the speed factor is recomputed at the load.)
    per_vehicle = (4 + 4 + 2 + 2 + 2 + 2) * 2 + 10 * (4 + 4)
    per_signal = 4 + 16 + 16 + 4 * (1 + 13 + 1 + 1 + 1 + 1 + 1 + 2)          # action, FSM in / out, mplight + the per-signal scalars
    return mean_active * per_vehicle + S * per_signal + O * 20 + 24 + 80     # + the drq_norm rows


KERNEL_SOURCES = ('resco_amd/csrc/resco_step.h', 'resco_amd/csrc/resco_sim.hip', 'resco_amd/csrc/resco_tables.h',
                  'include/resco_model.h', 'include/resco_sim.h')


def kernel_source_hash():
    """identifies the build a PMC summary was measured on (tools/pmc_passes.sh stores the same hash)"""
    import hashlib
    h = hashlib.sha256()
    for rel in KERNEL_SOURCES:
        with open(os.path.join(ROOT, rel), 'rb') as f:
            h.update(f.read())
    return h.hexdigest()[:16]


def pmc_traffic(args, n_local, world):
    """HBM bytes per STEP (= the `pipes` overlapping launches of one step) from the PMC counters.  They are collected in SEPARATE rocprofv3 --pmc passes of this very command
    (tools/pmc_passes.sh <tag> K W -> profiles/r03_pmc_s<K>_w<W>.json), so a figure is reported only when a committed
    summary exists for this run's --steps / --warmup on the default workload AND was measured on the kernel sources that are
    running now (source hash); otherwise None, with the reason."""
    path = os.path.join(ROOT, 'profiles', 'r06_pmc_s%d_w%d.json' % (args.steps, args.warmup))
    default_workload = world == 1 and args.map == 'ingolstadt21' and n_local == 4096 and args.block == 0 and args.pipes == DEFAULT_PIPES and args.tls_expiry == 1
    if not default_workload or not os.path.exists(path):
        return None, ('HBM bytes per launch come from separate rocprofv3 --pmc passes of this command (tools/pmc_passes.sh); '
                      'there is no committed summary for this workload / window (%s)' % os.path.basename(path))
    try:
        with open(path) as f:
            pm = json.load(f)
        if pm.get('source_hash') != kernel_source_hash():
            return None, ('stale: %s was measured on kernel sources %s, this build is %s -- re-run tools/pmc_passes.sh'
                          % (os.path.basename(path), pm.get('source_hash'), kernel_source_hash()))
        c = pm['counters']
        fetch_kib, write_kib = c['FETCH_SIZE']['per_launch_avg'], c['WRITE_SIZE']['per_launch_avg']
        # MI355X_MICROARCH.md (HBM / rocprofv3): both counters are in KiB; gfx950's FETCH_SIZE counts half of the bytes
        k = args.pipes
        return k * (2.0 * fetch_kib + write_kib) * 1024.0, (
            '%d x (2 x FETCH_SIZE + WRITE_SIZE) KiB per launch, averaged over the %d launches of the PMC passes of this same command '
            '(profiles/%s, same kernel sources %s): %d x (%.0f + %.0f) MB' % (k, c['FETCH_SIZE']['launches'], os.path.basename(path),
                                                                             pm['source_hash'], k, 2.0 * fetch_kib * 1024 / 1e6, write_kib * 1024 / 1e6))
    except Exception as e:
        return None, 'PMC summary %s unreadable: %r' % (os.path.basename(path), e)


EPISODE_MID = 180


def window_start(steps, warmup):
    """first timed step inside the episode: centred on the middle of the episode, never before `warmup`"""
    if steps >= EPISODE_STEPS:
        return warmup
    return max(warmup, min(EPISODE_MID - steps // 2, EPISODE_STEPS - steps))


def run_timed(sims, steps, warmup, barrier, sync, reduce_max, all_outputs_steps=0):
    """W untimed steps, then exactly K timed steps bracketed by barrier + device sync; max over ranks.  `sims`: the pipes of
    this rank -- every step launches [agent, step] on each pipe's own stream, one host thread issuing them in turn."""
    from resco_amd.sim import SimGroup
    # (tests/test_distributed_cpu.py drives this function with CPU stand-ins that have no C handle: per-pipe calls there)
    group = SimGroup(sims) if all(getattr(s, '_h', None) is not None for s in sims) else None
    k = 0

    def one():
        # ONE call through the C ABI per env-step for all pipes (rs_group_step): per pipe the random policy's kernel and the step
        # kernel on the pipe's own stream
        nonlocal k
        if k > 0 and k % EPISODE_STEPS == 0:
            for sim in sims:
                sim.reset()
        if PER_PIPE_CALLS or group is None:     # round 4's driver: two calls through ctypes per pipe and step (tools/host_ceiling.sh)
            for sim in sims:
                sim.act_random(k)
                sim.step(None)
        else:
            group.step('random', step_key=k)
        k += 1

    def stats():
        st = [sim.stats() for sim in sims]
        return {key: __import__('numpy').concatenate([x[key] for x in st]) for key in st[0]}

    # The LAST steps before the warm-up -- the end of the untimed fast-forward into the bulk of the episode, or, when there is
    # none, of the warm-up itself -- run with EVERY derived output buffer switched on (lane_agg, wave, mplight_full, the fp16
    # tensor, lane_arrivals next to drq_norm + mplight) and are timed on their own: what the output mask saves, reported next to
    # the headline figure.  Untimed as far as the contract's K steps go; `all_outputs_steps` of them whatever --warmup is.
    ff = window_start(steps, warmup) - warmup
    n_all = min(all_outputs_steps, ff + warmup)
    n_all_ff = min(n_all, ff)                   # ... of which inside the fast-forward
    all_outputs_rate = None
    # What the HOST needs to issue one step (all pipes): the first (up to) 48 untimed steps are handed to the runtime back to back and
    # timed until the last call returns -- before the device has worked them off (the queues take them all) -- and only then waited
    # for.  Under N ranks this is one rank's issue cost with the others competing for the cores: 1 / it is the step rate the host
    # side can sustain.  (The network is still empty then: these steps say nothing about the kernel.)
    n_plain = (ff - n_all_ff) + (0 if n_all_ff else warmup - n_all)
    n_probe = min(48, n_plain)
    issue_s = None
    if n_probe > 0:
        sync()
        barrier()
        t3 = time.perf_counter()
        for _ in range(n_probe):
            one()
        issue_s = (time.perf_counter() - t3) / n_probe
        sync()
    for _ in range(n_plain - n_probe):
        one()
    if n_all > 0:
        for sim in sims:
            sim.set_outputs(None)
        sync()
        t2 = time.perf_counter()
        for _ in range(n_all):
            one()
        sync()
        all_outputs_rate = n_all / (time.perf_counter() - t2)       # steps per second of this rank
        for sim in sims:
            sim.set_outputs(OUTPUTS)
    for _ in range(warmup if n_all_ff else 0):
        one()
    sync()
    st0 = stats()
    for sim in sims:
        sim.timing(True)
    barrier()
    sync()
    t0 = time.perf_counter()
    for _ in range(steps):
        one()
    sync()
    barrier()
    t1 = time.perf_counter()
    kernel_ms, launches = 0.0, 0
    for sim in sims:
        ms, n = sim.timing_read()
        kernel_ms += ms
        launches += n
        sim.timing(False)
    st1 = stats()
    elapsed = reduce_max(t1 - t0)
    return elapsed, kernel_ms, launches, st0, st1, all_outputs_rate, n_all, t1 - t0, issue_s


def state_digest(sims, dist, rank, world):
    """sha1 over the per-environment state (vehicles, signals, counters) of every environment of every rank in GLOBAL environment
    order: equal for any split of the same batch over ranks and pipes (the global index keys the RNG)."""
    import hashlib
    import numpy as np
    per_env = []
    for sim in sims:
        bufs = [np.ascontiguousarray(sim.read(b)) for b in ('veh_lane', 'veh_trip', 'veh_pos', 'veh_speed', 'veh_swait', 'veh_rwait', 'tls', 'stats',
                                                            'mplight', 'drq_norm')]
        for e in range(sim.n_envs):
            h = hashlib.sha1()
            for b in bufs:
                h.update(b[e].tobytes())
            per_env.append(h.hexdigest())
    if dist is not None and world > 1:
        parts = [None] * world
        dist.all_gather_object(parts, per_env)
        per_env = [x for p in parts for x in p]
    return hashlib.sha1(''.join(per_env).encode()).hexdigest()


def bind_to_gpu_numa_node(local_rank):
    """N > 1: every rank is one host thread issuing ~2 launches per 0.1-2 ms; pin it to the cores of the NUMA node its GPU hangs
    off (PCI address from the HIP runtime -> sysfs), so that 8 ranks do not migrate across sockets.  Best effort and
    conservative: without a PCI address, a NUMA node or usable cores nothing is changed.  Returns the node or None."""
    try:
        import torch
        prop = torch.cuda.get_device_properties(local_rank)
        if not all(hasattr(prop, a) for a in ('pci_domain_id', 'pci_bus_id', 'pci_device_id')):
            return None
        dev = '/sys/bus/pci/devices/%04x:%02x:%02x.0' % (prop.pci_domain_id, prop.pci_bus_id, prop.pci_device_id)
        node = int(open(os.path.join(dev, 'numa_node')).read())
        if node < 0:
            return None
        cpus = []
        for part in open('/sys/devices/system/node/node%d/cpulist' % node).read().strip().split(','):
            a, _, b = part.partition('-')
            cpus += list(range(int(a), int(b or a) + 1))
        cpus = sorted(set(cpus) & os.sched_getaffinity(0))
        if len(cpus) < 2:
            return None
        os.sched_setaffinity(0, cpus)
        return node
    except Exception:
        return None


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
    _cpu_worker((0, 1, seed, EPISODE_STEPS))
    probe = (time.perf_counter() - t0) / EPISODE_STEPS     # seconds per env-step on one core, over a WHOLE episode
    per_worker = max(EPISODE_STEPS, min(EPISODE_STEPS * 8, int(budget_s / max(probe, 1e-6)) // EPISODE_STEPS * EPISODE_STEPS))
    jobs = [(i, 1, seed, per_worker) for i in range(cores)]
    t0 = time.perf_counter()
    with mp.get_context('fork').Pool(cores) as pool:
        done = pool.map(_cpu_worker, jobs)
    wall = time.perf_counter() - t0
    total = sum(done)
    return dict(value=total / wall, unit='env-steps/s', cores=cores, kind='port', single_thread_value=1.0 / probe,
                sample='C oracle (oracle/resco_oracle.c, gcc -O3 -march=x86-64-v3, one environment per process): %d processes (affinity %d, '
                       'cgroup quota applied) x %d env-steps (whole 360-step episodes of ingolstadt21), same hashed random policy, %.1f s wall'
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


def launch_ranks(n):
    """re-exec as `python -m torch.distributed.run --nnodes=1 --nproc-per-node n --master-addr 127.0.0.1 --master-port <free> bench.py <same flags>`"""
    import socket
    with socket.socket() as s:
        s.bind(('127.0.0.1', 0))
        port = s.getsockname()[1]
    env = dict(os.environ)
    env.setdefault('HSA_ENABLE_IPC_MODE_LEGACY', '0')      # RCCL on this pool: dmabuf IPC only
    env.setdefault('OMP_NUM_THREADS', '1')
    cmd = [sys.executable, '-m', 'torch.distributed.run', '--nnodes=1', '--nproc-per-node', str(n), '--master-addr', '127.0.0.1',
           '--master-port', str(port), os.path.abspath(__file__)] + sys.argv[1:]
    sys.stdout.flush()
    os.execve(sys.executable, cmd, env)


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument('--gpus', type=int, default=1)
    ap.add_argument('--steps', type=int, default=300)
    ap.add_argument('--warmup', type=int, default=60)
    ap.add_argument('--map', default='ingolstadt21')
    ap.add_argument('--envs', type=int, default=4096, help='environments per GPU')
    ap.add_argument('--pipes', type=int, default=DEFAULT_PIPES, help='handles (HIP streams) the batch of a GPU is split into')
    ap.add_argument('--block', type=int, default=0, help='threads per workgroup (0 = library default)')
    ap.add_argument('--seed', type=int, default=0)
    ap.add_argument('--no-cpu-baseline', action='store_true')
    ap.add_argument('--tls-expiry', type=int, default=1, choices=(0, 1),
                    help='1 (default, rs_params.tls_hold = 0): a phase set through setPhase expires after its programme duration (SUMO\'s '
                         'documented setPhase); 0: it stays until the next action (round 5\'s default: a heavier network, 469 instead of 421 '
                         'vehicles per environment in the driver\'s window); the figure is reported for both (profiles/r06_bench_both_modes.txt)')
    ap.add_argument('--digest', action='store_true', help='add a digest of the final per-environment state (all ranks, global env order)')
    args = ap.parse_args()

    if args.gpus < 1:
        raise SystemExit('--gpus must be >= 1')
    if 'WORLD_SIZE' not in os.environ and args.gpus > 1:
        # plain `python bench.py --gpus N`: become the contract's launch -- N ranks, one per GPU, through torch.distributed.run
        # (the reference's only parallelism is a process per trial, main.py:40-44; here a process per GPU)
        launch_ranks(args.gpus)         # does not return
    import torch
    rank = int(os.environ.get('RANK', '0'))
    world = int(os.environ.get('WORLD_SIZE', '1'))
    local = int(os.environ.get('LOCAL_RANK', '0'))
    if world != args.gpus:
        # a launcher / flag mismatch must never print a line that claims N GPUs while timing another number of them
        raise SystemExit('bench.py: --gpus %d but the launcher started WORLD_SIZE=%d ranks: start exactly --gpus ranks '
                         '(python -m torch.distributed.run --nproc-per-node %d ... bench.py --gpus %d, or plain python bench.py --gpus %d)'
                         % (args.gpus, world, args.gpus, args.gpus, args.gpus))
    if not torch.cuda.is_available():
        raise SystemExit('bench.py needs an MI355X: the HIP path has no CPU fallback')
    if 'RESCO_BENCH_DEVICE' not in os.environ and torch.cuda.device_count() < world:
        raise SystemExit('bench.py: --gpus %d but only %d device(s) visible (RESCO_BENCH_DEVICE=<id> puts every rank on one device: a '
                         'functional check, not a measurement)' % (world, torch.cuda.device_count()))
    # RESCO_BENCH_DEVICE / RESCO_BENCH_BACKEND: several ranks on ONE GPU with a gloo rendezvous -- how the N > 1 path is run
    # end to end where no multi-GPU node exists (tests/test_gpu_parity.py::test_two_ranks_through_bench_on_one_gpu)
    local = int(os.environ.get('RESCO_BENCH_DEVICE', local))
    backend = os.environ.get('RESCO_BENCH_BACKEND', 'nccl')
    torch.cuda.set_device(local)
    numa = bind_to_gpu_numa_node(local) if world > 1 and os.environ.get('RESCO_BENCH_NO_NUMA') != '1' else None
    dist = None
    if world > 1 or os.environ.get('RESCO_BENCH_FORCE_DIST') == '1':      # the env var exercises the RCCL path at N=1
        import torch.distributed as dist
        dist.init_process_group(backend, rank=rank, world_size=world)     # RCCL: barrier + one MAX only
    # the world size the process group itself reports (None: single process, no group)
    rccl_ranks = dist.get_world_size() if dist is not None else None
    if rccl_ranks is not None and rccl_ranks != args.gpus:
        raise SystemExit('bench.py: the process group has %d ranks, --gpus says %d' % (rccl_ranks, args.gpus))

    from resco_amd.scenario import Scenario
    from resco_amd.sim import BatchedSim
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', args.map + '.npz'))
    env_base, n_local = shard(rank, world, args.envs)
    if args.pipes < 1 or n_local % args.pipes:
        raise SystemExit('--envs must be a multiple of --pipes')
    per = n_local // args.pipes
    sims = [BatchedSim(sc, per, device=local, seed=args.seed, sigma=-1.0, speed_dev=1, env_base=env_base + i * per,
                       block_threads=args.block, device_envs=n_local, tls_expiry=args.tls_expiry) for i in range(args.pipes)]
    sim = sims[0]
    # BASELINE config 3 / SURVEY 8(d): "state fns computed every step: lane aggregates -> drq_norm + mplight; rewards wait +
    # pressure" -- only what those consume is written (the per-signal rewards and metrics always are)
    for x in sims:
        x.set_outputs(OUTPUTS)

    def barrier():
        if dist is not None:
            dist.barrier()

    def sync():
        for x in sims:
            x.sync()
        torch.cuda.synchronize()

    def reduce_max(x):
        if dist is None:
            return x
        t = torch.tensor([x], dtype=torch.float64, device='cuda' if backend == 'nccl' else 'cpu')
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed, kernel_ms, launches, st0, st1, all_out_rate, n_all, elapsed_local, issue_s = run_timed(sims, args.steps, args.warmup, barrier, sync, reduce_max,
                                                                                             all_outputs_steps=20)
    ticks = (st1['ticks'] - st0['ticks']).astype('float64')
    mean_active = float(((st1['active_ticks'] - st0['active_ticks']) / ticks.clip(min=1)).mean()) if ticks.min() > 0 \
        else float(st1['active'].mean())
    info = sim.info()
    env_steps = world * n_local * args.steps
    value = env_steps / elapsed
    b_alg = algorithmic_bytes_per_env_step(sc, mean_active)
    b_wide = designed_bytes_per_env_step(sc, mean_active)
    k_avg_s = (kernel_ms / max(1, launches)) * 1e-3
    # one step = `pipes` launches of n_local / pipes environments that overlap on their streams: the bytes of all of them over
    # the average duration of one of them (= pipes x the per-launch figure)
    achieved_per_launch = b_alg * per / k_avg_s / 1e9 if k_avg_s > 0 else 0.0
    # `achieved`: the algorithmic bytes of everything this rank stepped over ITS wall clock of the timed region -- no assumption
    # about how well the launches of the pipes overlap (pipes x the per-launch figure is the upper bound, reported next to it)
    achieved = b_alg * n_local * args.steps / elapsed_local / 1e9
    w0 = window_start(args.steps, args.warmup)
    traffic, traffic_note = pmc_traffic(args, n_local, world)
    out = {
        'metric': 'env-steps/sec', 'value': value, 'unit': 'env-steps/s', 'n_gpus': world, 'rccl_ranks': rccl_ranks, 'steps': args.steps,
        'warmup': args.warmup, 'ms_per_step': elapsed / args.steps * 1e3, 'higher_is_better': True,
        'scaling': 'weak', 'vs_baseline': None, 'dtype': 'f32', 'data': 'synthetic',
        'config': {'workload': '%s x %d lock-step envs per GPU (BASELINE config 3), fixed demand from the map\'s '
                               'rou.xml, on-device seeded random policy, Krauss sigma 0.5 + speedFactor dev 0.1'
                               % (args.map, n_local),
                   'tls_expiry': args.tls_expiry, 'map': args.map, 'envs_per_gpu': n_local, 'pipes': args.pipes, 'ticks_per_env_step': 10,
                   'episode_window': [w0, w0 + args.steps], 'untimed_fast_forward_steps': w0 - args.warmup,
                   'block_threads': info['block_threads'], 'lds_bytes_per_env': info['lds_bytes'],
                   'outputs_per_step': list(OUTPUTS) + ['wait', 'wait_norm', 'pressure', 'phase', 'queue_sum', 'queue_max', 'arrivals', 'departures'],
                   'parallelism': 'env-batch split x%d, no collective on the data path' % world,
                   'rank0_numa_node': numa},
        'roofline': {'bound': 'hbm', 'achieved': achieved, 'peak': HBM_PEAK_GBS, 'unit': 'GB/s',
                     'frac': achieved / HBM_PEAK_GBS,
                     'traffic': traffic, 'traffic_note': traffic_note,
                     'kernel': 'rs_step_kernel', 'kernel_avg_ms': k_avg_s * 1e3, 'launches': launches,
                     'concurrent_launches': args.pipes, 'achieved_per_launch': achieved_per_launch,
                     'achieved_upper_bound': args.pipes * achieved_per_launch,
                     'achieved_note': 'a step is %d launches of %d environments each on %d HIP streams.  `achieved` = algorithmic bytes of the '
                                      '%d env-steps of the timed region / its wall clock on this rank; `achieved_per_launch` = algorithmic bytes of '
                                      'one launch / its average duration by HIP events on its own stream, `achieved_upper_bound` = %d x that (what '
                                      'perfectly overlapping launches would give); `traffic` = %d x the per-launch counter figure'
                                      % (args.pipes, per, args.pipes, n_local * args.steps, args.pipes, args.pipes),
                     'algorithmic_bytes_per_env_step': b_alg, 'env_steps_per_launch': per,
                     'formula': 'SURVEY 8(d): 60*V + 12*S + 20*SL + 8*S', 'designed_bytes_per_env_step': b_wide,
                     'achieved_designed_bytes': b_wide * n_local * args.steps / elapsed_local / 1e9,
                     'note': 'state is Infinity-Cache resident and the kernel is issue/latency bound: the HBM '
                             'fraction is small by construction (SURVEY.md 8d)'},
        'mean_active_vehicles_per_env': mean_active,
        'host': {'issue_us_per_step': issue_s * 1e6 if issue_s else None, 'calls_per_step': 2 * args.pipes if PER_PIPE_CALLS else 1,
                 'note': 'rank 0: wall time of the calls that hand ONE step (agent + step kernel of every pipe) to the runtime, measured on the first (up to) 48 untimed steps, '
                         'issued back to back and not waited for until all are issued; %s'
                         % ('two calls through ctypes per pipe (rs_act_random, rs_step)' if PER_PIPE_CALLS else 'one call through ctypes (rs_group_step)')},
        'sim_ticks_per_s': value * 10, 'vehicle_ticks_per_s': value * 10 * mean_active,
        'all_outputs': {'value': world * n_local * all_out_rate if all_out_rate else None, 'unit': 'env-steps/s', 'steps': n_all,
                        'episode_window': [max(0, w0 - args.warmup - n_all), max(0, w0 - args.warmup - n_all) + n_all] if w0 - args.warmup >= n_all else [w0 - n_all, w0],
                        'note': 'rank 0, the last untimed steps before the warm-up (the network is nearly as loaded as in the timed window) '
                                'with EVERY derived buffer written (lane_agg, drq_norm, wave, mplight, mplight_full, '
                                'drq_norm_f16, lane_arrivals): rounds 1-2 measured this workload, rounds 3-4 write what config 3 consumes'},
    }
    if args.digest:
        out['state_digest'] = state_digest(sims, dist, rank, world)
    if rank == 0:
        if world == 1 and not args.no_cpu_baseline:
            try:
                out['cpu_baseline'] = cpu_baseline(sc, args.seed)
                # a SUMO / libsumo timing is reported only when SUMO exists on the box
                from tools.sumo_runner import sumo_baseline
                out['sumo_baseline'] = sumo_baseline(args.map, budget_s=20.0)
            except Exception as e:          # the baseline must never take the GPU number down with it
                out['cpu_baseline'] = {'value': None, 'unit': 'env-steps/s', 'cores': os.cpu_count(), 'kind': 'port',
                                       'sample': 'failed: %r' % (e,)}
    for x in sims:
        x.close()
    if dist is not None:
        dist.barrier()
        dist.destroy_process_group()
    if rank == 0:
        # the ONE JSON line goes out last: RCCL prints its version banner through C stdio, which a pipe only sees when the
        # C buffers are flushed
        import ctypes
        try:
            ctypes.CDLL(None).fflush(None)
        except Exception:
            pass
        print(json.dumps(out), flush=True)


if __name__ == '__main__':
    main()
