#!/usr/bin/env python3
"""SUMO side of the comparison -- active only where SUMO exists.

The path this repository replaces is `self.sumo.simulationStep()` behind RESCO's MultiSignal
(resco_benchmark/multi_signal.py:102-105, started with `--random --time-to-teleport -1`, :127-131) plus the TraCI calls of
Signal (traffic_signal.py:96-100 program install, :174 getPhase, :184-187 setPhase, :240 getLastStepVehicleIDs).  SUMO is
not part of this repository and is not installed in the build container; when a box does have it (`libsumo` or `traci`
importable, or a `sumo` binary on PATH) and the scenario XML files of the reference are reachable, this module provides

  sumo_baseline(map, budget_s)     bench.py's `sumo_baseline`: env-steps/s of the SAME step loop (prep -> yellow ticks ->
                                   set -> green ticks -> per-lane observe) on libsumo / TraCI, one process per host core
  diff_vs_sumo(map, steps, seed)   the north-star check: the same net / route files, sigma = 0, speedDev = 0, a fixed action
                                   script -> per-step phase indices and per-lane vehicle counts of SUMO next to rs_step's
                                   (python tools/sumo_runner.py diff --map cologne1 --steps 60; needs a GPU as well)

and otherwise says exactly what is missing -- never a fabricated number.  Nothing here is imported by resco_amd/.

Scenario files are looked for in $RESCO_ENVIRONMENTS/<map>/<map>.sumocfg, then in an installed `resco_benchmark` package,
then in /root/reference/resco_benchmark/environments (the build container's read-only reference checkout).
"""
import argparse
import json
import multiprocessing as mp
import os
import shutil
import sys
import time

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
if ROOT not in sys.path:
    sys.path.insert(0, ROOT)


def find_sumo():
    """-> ('libsumo' | 'traci' | None, module or None, binary or None)"""
    try:
        import libsumo
        return 'libsumo', libsumo, shutil.which('sumo')
    except Exception:
        pass
    binary = shutil.which('sumo')
    home = os.environ.get('SUMO_HOME')
    if binary is None and home and os.path.exists(os.path.join(home, 'bin', 'sumo')):
        binary = os.path.join(home, 'bin', 'sumo')
    if home and os.path.join(home, 'tools') not in sys.path:
        sys.path.append(os.path.join(home, 'tools'))
    if binary is not None:
        try:
            import traci
            return 'traci', traci, binary
        except Exception:
            return None, None, binary
    return None, None, None


def find_sumocfg(map_name):
    cands = []
    if os.environ.get('RESCO_ENVIRONMENTS'):
        cands.append(os.environ['RESCO_ENVIRONMENTS'])
    try:
        import resco_benchmark
        cands.append(os.path.join(os.path.dirname(resco_benchmark.__file__), 'environments'))
    except Exception:
        pass
    cands.append('/root/reference/resco_benchmark/environments')
    for d in cands:
        p = os.path.join(d, map_name, map_name + '.sumocfg')
        if os.path.exists(p):
            return p
    return None


class SumoLoop:
    """The reference's step loop on a real SUMO, restated minimally: program install as Signal.__init__ does it, the
    prep / yellow / set FSM, step_length ticks, per-lane counts of the configured lanes."""

    def __init__(self, map_name, seed=0, deterministic=False, label=None):
        from resco_amd.config.map_config import map_configs
        from resco_amd.scenario import Scenario, build_yellow_program, green_phases
        kind, api, binary = find_sumo()
        if api is None:
            raise RuntimeError('SUMO unavailable')
        cfg = find_sumocfg(map_name)
        if cfg is None:
            raise RuntimeError('scenario files not found')
        self.api, self.kind = api, kind
        self.mc = map_configs[map_name]
        self.sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', map_name + '.npz'))
        cmd = [binary or 'sumo', '-c', cfg, '--no-warnings', 'True', '--time-to-teleport', '-1', '--seed', str(int(seed))]
        if deterministic:
            cmd += ['--default.speeddev', '0']
        if kind == 'libsumo':
            api.start(cmd)
            self.conn = api
        else:
            api.start(cmd, label=label or ('resco_amd_%d' % os.getpid()))
            self.conn = api.getConnection(label or ('resco_amd_%d' % os.getpid()))
        c = self.conn
        if deterministic:       # parity mode of the comparison: no dawdling, no speed-factor spread
            for vt in c.vehicletype.getIDList():
                c.vehicletype.setImperfection(vt, 0.0)
                c.vehicletype.setSpeedDeviation(vt, 0.0)
        self.ids = list(self.sc.signal_ids)
        self.yellow, self.n_green = {}, {}
        Y = self.mc['yellow_length']
        for sid in self.ids:
            logic = c.trafficlight.getAllProgramLogics(sid)[0]
            prog = [(int(p.duration), p.state) for p in logic.getPhases()]
            greens = green_phases(prog)                                  # multi_signal.py:52-59
            phases, ydict = build_yellow_program(greens, Y)              # traffic_signal.py:7-24
            logic.type = 0                                               # traffic_signal.py:96-100
            logic.phases = [api.trafficlight.Phase(d, st) for d, st in phases]
            c.trafficlight.setProgramLogic(sid, logic)
            self.yellow[sid], self.n_green[sid] = ydict, len(greens)
        self.next_phase = {sid: 0 for sid in self.ids}
        self.lanes = [l for l in self.sc.obs_lane_ids]

    def step(self, actions):
        c, T, Y = self.conn, self.mc['step_length'], self.mc['yellow_length']
        for sid, a in zip(self.ids, actions):                            # Signal.prep_phase, traffic_signal.py:176-184
            cur = c.trafficlight.getPhase(sid)
            self.next_phase[sid] = int(a)
            if cur != a:
                key = '%d_%d' % (cur, a)
                if key in self.yellow[sid]:
                    c.trafficlight.setPhase(sid, self.yellow[sid][key])
        for _ in range(Y):
            c.simulationStep()
        for sid in self.ids:                                             # Signal.set_phase, :186-187
            c.trafficlight.setPhase(sid, self.next_phase[sid])
        for _ in range(T - Y):
            c.simulationStep()

    def observe(self):
        c = self.conn
        phases = [c.trafficlight.getPhase(sid) for sid in self.ids]
        counts = []
        for lane in self.lanes:
            try:
                counts.append(c.lane.getLastStepVehicleNumber(lane))
            except Exception:
                counts.append(-1)                                        # configured lane that is not in the net (cologne8 has two)
        return phases, counts

    def close(self):
        try:
            self.conn.close()
        except Exception:
            pass


def _timing_worker(job):
    map_name, idx, budget_s = job
    import numpy as np
    loop = SumoLoop(map_name, seed=1000 + idx, label='bench_%d' % idx)
    rng = np.random.default_rng(idx)
    n, t0 = 0, time.perf_counter()
    while time.perf_counter() - t0 < budget_s and n < 360:
        loop.step([int(rng.integers(0, loop.n_green[sid])) for sid in loop.ids])
        loop.observe()
        n += 1
    dt = time.perf_counter() - t0
    loop.close()
    return n, dt


def sumo_baseline(map_name, budget_s=20.0):
    """-> dict for bench.py's JSON line, or a string saying what is missing"""
    kind, api, binary = find_sumo()
    if api is None:
        return 'SUMO unavailable on this host' if binary is None else 'sumo binary found, but neither libsumo nor traci can be imported'
    if find_sumocfg(map_name) is None:
        return 'SUMO (%s) found, but the scenario XML files of %s are not on this box (set RESCO_ENVIRONMENTS)' % (kind, map_name)
    cores = len(os.sched_getaffinity(0)) if hasattr(os, 'sched_getaffinity') else (os.cpu_count() or 1)
    try:
        with mp.get_context('spawn').Pool(cores) as pool:          # libsumo holds one simulation per process
            res = pool.map(_timing_worker, [(map_name, i, budget_s) for i in range(cores)])
    except Exception as e:
        return 'SUMO (%s) found but the run failed: %r' % (kind, e)
    steps, wall = sum(r[0] for r in res), max(r[1] for r in res)
    return dict(value=steps / wall, unit='env-steps/s', cores=cores, kind=kind,
                sample='%d processes x up to %.0f s of the prep/yellow/set/observe loop of %s on %s with a random policy'
                       % (cores, budget_s, map_name, kind))


def diff_vs_sumo(map_name, steps=60, seed=0, sim_cls=None):
    """phase indices and per-lane vehicle counts: SUMO (sigma 0, speedDev 0) vs rs_step (parity mode), same action script.
    sim_cls: the simulator class to put next to SUMO (default BatchedSim = the HIP library; the CPU test of this runner passes
    the host emulation of the kernel and a stand-in for libsumo, tests/test_sumo_runner.py)"""
    import numpy as np
    if sim_cls is None:
        from resco_amd.sim import BatchedSim as sim_cls
    loop = SumoLoop(map_name, seed=seed, deterministic=True)
    sc = loop.sc
    sim = sim_cls(sc, 1, seed=seed, sigma=0.0, speed_dev=0, max_distance=1.0e9)
    obs_lane = np.asarray(sc.obs_lane)
    rng = np.random.default_rng(seed)
    phase_equal, count_equal, count_abs, n_cmp = 0, 0, 0.0, 0
    rows = []
    for k in range(steps):
        acts = [int(rng.integers(0, g)) for g in sc.tls_ngreen]
        loop.step(acts)
        sim.step(np.asarray([acts], np.int32))
        ph_s, cnt_s = loop.observe()
        ph_g = sim.read('phase')[0].tolist()
        lanes_now = sim.read('veh_lane')[0]
        per_lane = np.bincount(lanes_now[lanes_now != 0xFFFF].astype(np.int64), minlength=sc.n_lanes)
        cnt_g = [int(per_lane[l]) if l >= 0 else -1 for l in obs_lane]   # lane.getLastStepVehicleNumber of every configured lane
        phase_equal += int(ph_s == ph_g)
        valid = [i for i, c_ in enumerate(cnt_s) if c_ >= 0]
        count_equal += sum(1 for i in valid if cnt_s[i] == cnt_g[i])
        count_abs += sum(abs(cnt_s[i] - cnt_g[i]) for i in valid)
        n_cmp += len(valid)
        rows.append(dict(step=k, phases_sumo=ph_s, phases_hip=ph_g, lanes_differing=sum(1 for i in valid if cnt_s[i] != cnt_g[i])))
    loop.close()
    sim.close()
    return dict(map=map_name, steps=steps, phase_vectors_equal=phase_equal, lane_counts_equal=count_equal, lane_counts_compared=n_cmp,
                mean_abs_count_diff=count_abs / max(1, n_cmp), per_step=rows)


if __name__ == '__main__':
    ap = argparse.ArgumentParser()
    ap.add_argument('cmd', choices=['probe', 'time', 'diff'])
    ap.add_argument('--map', default='cologne1')
    ap.add_argument('--steps', type=int, default=60)
    ap.add_argument('--budget', type=float, default=20.0)
    a = ap.parse_args()
    if a.cmd == 'probe':
        kind, api, binary = find_sumo()
        print(json.dumps(dict(api=kind, binary=binary, sumocfg=find_sumocfg(a.map))))
    elif a.cmd == 'time':
        print(json.dumps(sumo_baseline(a.map, a.budget)))
    else:
        kind, api, binary = find_sumo()
        if api is None or find_sumocfg(a.map) is None:
            print(json.dumps(dict(error=sumo_baseline(a.map, 0.0))))
        else:
            print(json.dumps(diff_vs_sumo(a.map, a.steps)))
