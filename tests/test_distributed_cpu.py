"""CPU, world_size 2, gloo: the multi-GPU path of bench.py is an env-batch split with no data-path
collective -- ranks own disjoint environment ranges keyed by their GLOBAL index, meet at a barrier, and
report the MAX of their times; inside a rank the batch is split once more into pipes (handles on their own HIP
streams).  The union of the shards and pipes must equal the single-process, single-handle batch."""
import os
import pickle
import sys
import tempfile

import numpy as np
import pytest
import torch.multiprocessing as mp

from conftest import ROOT, load_scenario

ENVS_PER_RANK, PIPES, STEPS, WARMUP, SEED = 4, 2, 4, 2, 21


def _rank_main(rank, world, port, outdir):
    import torch
    import torch.distributed as dist
    sys.path.insert(0, ROOT)
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    os.environ.update(MASTER_ADDR='127.0.0.1', MASTER_PORT=str(port))
    dist.init_process_group('gloo', rank=rank, world_size=world)
    import bench
    from oracle_batch import OracleBatch
    sc = load_scenario('cologne1')
    base, n = bench.shard(rank, world, ENVS_PER_RANK)
    per = n // PIPES            # the batch of a rank is stepped as PIPES handles (bench.py --pipes), each on its own stream on the GPU
    sims = [OracleBatch(sc, per, seed=SEED, env_base=base + i * per) for i in range(PIPES)]

    def reduce_max(x):
        t = torch.tensor([x], dtype=torch.float64)
        dist.all_reduce(t, op=dist.ReduceOp.MAX)
        return float(t.item())

    elapsed, kms, launches, st0, st1 = bench.run_timed(sims, STEPS, WARMUP, dist.barrier, lambda: None, reduce_max)[:5]
    with open(os.path.join(outdir, 'rank%d.pkl' % rank), 'wb') as f:
        pickle.dump(dict(base=base, n=n, elapsed=elapsed, launches=launches, mplight=np.concatenate([x.read('mplight') for x in sims]),
                         lane_agg=np.concatenate([x.read('lane_agg') for x in sims]), ticks=(st1['ticks'] - st0['ticks'])), f)
    dist.barrier()
    dist.destroy_process_group()


def test_env_batch_split_over_two_ranks():
    outdir = tempfile.mkdtemp()
    port = 29500 + (os.getpid() % 2000)
    mp.spawn(_rank_main, args=(2, port, outdir), nprocs=2, join=True)
    parts = [pickle.load(open(os.path.join(outdir, 'rank%d.pkl' % r), 'rb')) for r in range(2)]
    assert [p['base'] for p in parts] == [0, ENVS_PER_RANK] and all(p['n'] == ENVS_PER_RANK for p in parts)
    assert parts[0]['elapsed'] == parts[1]['elapsed'] > 0          # MAX over ranks, identical everywhere
    assert all(p['launches'] == STEPS * PIPES for p in parts)            # one launch per pipe and step
    assert all((p['ticks'] == STEPS * 10).all() for p in parts)
    # single-process reference batch of 2 x ENVS_PER_RANK environments
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    import bench
    from oracle_batch import OracleBatch
    whole = OracleBatch(load_scenario('cologne1'), 2 * ENVS_PER_RANK, seed=SEED, env_base=0)
    bench.run_timed([whole], STEPS, WARMUP, lambda: None, whole.sync, lambda x: x)
    np.testing.assert_array_equal(whole.read('mplight'), np.concatenate([p['mplight'] for p in parts]))
    np.testing.assert_array_equal(whole.read('lane_agg'), np.concatenate([p['lane_agg'] for p in parts]))


def test_algorithmic_bytes_formula():
    import bench
    sc = load_scenario('ingolstadt21')
    b = bench.algorithmic_bytes_per_env_step(sc, 300.0)
    # 60 B per active vehicle + 12 B per signal (action, FSM in / out) + 20 B per observed lane + 8 B of rewards per signal
    assert b == 300 * 60 + 21 * 12 + 163 * 20 + 21 * 8
    assert bench.designed_bytes_per_env_step(sc, 300.0) > b


def test_bench_gpus_flag_means_that_many_ranks(monkeypatch):
    """`python bench.py --gpus N` (no launcher) re-executes itself as the contract's N-rank launch; a launcher whose world size differs
    from --gpus is refused with a non-zero exit BEFORE anything is timed (round-5 review: `--gpus 8` under one process silently timed
    one GPU and printed n_gpus 1)."""
    import subprocess
    import bench
    seen = {}

    def fake_execve(exe, cmd, env):
        seen['cmd'], seen['env'] = cmd, env
        raise SystemExit(0)

    monkeypatch.setattr(os, 'execve', fake_execve)
    monkeypatch.setattr(sys, 'argv', ['bench.py', '--gpus', '4', '--steps', '3', '--warmup', '1'])
    monkeypatch.delenv('WORLD_SIZE', raising=False)
    with pytest.raises(SystemExit):
        bench.main()
    cmd = seen['cmd']
    assert cmd[1:4] == ['-m', 'torch.distributed.run', '--nnodes=1'] and cmd[cmd.index('--nproc-per-node') + 1] == '4'
    assert cmd[cmd.index('--master-addr') + 1] == '127.0.0.1' and cmd[-6:] == ['--gpus', '4', '--steps', '3', '--warmup', '1']
    assert seen['env']['HSA_ENABLE_IPC_MODE_LEGACY'] == '0'
    # launcher / flag mismatch: a world of 1 with --gpus 2 (and the reverse) must fail loudly, on any box
    root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
    for world, gpus in ((1, 2), (2, 1)):
        r = subprocess.run([sys.executable, os.path.join(root, 'bench.py'), '--gpus', str(gpus), '--steps', '2', '--warmup', '1'],
                           env=dict(os.environ, WORLD_SIZE=str(world), RANK='0', LOCAL_RANK='0'), capture_output=True, text=True, timeout=170)
        assert r.returncode != 0 and 'started WORLD_SIZE=%d' % world in r.stderr and '{"metric"' not in r.stdout
