#!/usr/bin/env python3
"""Host launch cost at small batches: [static agent -> rs_step] issued launch by launch vs replayed from a captured HIP graph.

Config 2 (cologne1 x 1024, MaxPressure on the device) runs ~0.1-0.2 ms per env-step: two kernel launches through ctypes per
step are a visible share of that, and at 8 GPUs (8 processes on one host) it is what decides the per-GPU retention (SURVEY
8e).  The library's launches are plain asynchronous kernel launches on the caller's stream (no sync, no host copies when
actions stay on the device), so torch.cuda.graph captures them: K steps per graph, replayed.

  python tools/graph_ab.py            -> one JSON line per (config, mode)
"""
import json
import os
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import torch                                  # noqa: E402
from resco_amd.scenario import Scenario      # noqa: E402
from resco_amd.sim import BatchedSim         # noqa: E402


def run(name, n, policy, steps=360, per_graph=0):
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    sim = BatchedSim(sc, n, seed=0)
    stream = torch.cuda.Stream()
    sp = stream.cuda_stream

    def one(k):
        if policy == 'maxpressure':
            sim.act_maxwave(1, stream=sp)
        else:
            sim.act_random(k, stream=sp)
        sim.step(None, stream=sp)

    one(0)                                    # uploads the agent tables, warms the kernels
    stream.synchronize()
    graph = None
    if per_graph:
        graph = torch.cuda.CUDAGraph()
        with torch.cuda.graph(graph, stream=stream):
            for k in range(per_graph):
                one(k)
    torch.cuda.synchronize()
    t0 = time.perf_counter()
    if graph is None:
        for k in range(steps):
            one(k)
    else:
        with torch.cuda.stream(stream):
            for _ in range(steps // per_graph):
                graph.replay()
    stream.synchronize()
    dt = time.perf_counter() - t0
    done = steps if graph is None else steps // per_graph * per_graph
    st = sim.stats()
    out = dict(map=name, envs=n, policy=policy, mode='launches' if graph is None else 'hip graph, %d steps per replay' % per_graph,
               steps=done, us_per_step=dt / done * 1e6, env_steps_per_s=n * done / dt, ticks=int(st['ticks'][0]))
    sim.close()
    return out


if __name__ == '__main__':
    for cfg, name, n, pol in ((2, 'cologne1', 1024, 'maxpressure'), (4, 'cologne8', 2048, 'maxpressure'), (3, 'ingolstadt21', 4096, 'maxpressure')):
        for pg in (0, 1, 10):
            print(json.dumps(dict(config=cfg, **run(name, n, pol, per_graph=pg))), flush=True)
