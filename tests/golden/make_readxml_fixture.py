#!/usr/bin/env python3
"""Close the trip-metric path against its consumer (BUILD CONTAINER ONLY): the reference's own utils/readXML.py, unmodified,
reads a tripinfo_<run>.xml that the PRODUCT's writer (resco_amd.multi_signal.tripinfo_records / write_tripinfo) produced from
an oracle episode, and the per-episode averages it prints become tests/golden/readxml_<map>.json.  The GPU test
(test_gpu_parity.py::test_trip_metrics_equal_what_the_reference_reads) replays the same episode through the C ABI and asserts
that BatchedSim.trip_metrics() gives those numbers.

readXML.py is a script: it walks <cwd>/../../results/*/tripinfo_<i>.xml, reads the demand from <cwd>/../environments/<map>/
and appends "'<run name>': [per-episode averages]" to avg_<metric>.py in the cwd.  It is run here (runpy) inside a scratch tree
that mirrors that layout, with matplotlib stubbed (it selects the TkAgg backend and calls plt.show()).

  python tests/golden/make_readxml_fixture.py
"""
import json
import os
import runpy
import sys
import tempfile
import types

import numpy as np

TLS_EXPIRY = 0      # the committed fixtures were generated under round 5's default (the phase stays); the test passes the recorded value

HERE = os.path.dirname(os.path.abspath(__file__))
ROOT = os.path.dirname(os.path.dirname(HERE))
sys.path.insert(0, ROOT)
REF = '/root/reference'
CASES = [('cologne1', 'STOCHASTIC', 200, 5), ('cologne3', 'STOCHASTIC', 200, 5)]      # (map, policy, max_distance, seed)


def episode(name, policy, max_distance, seed):
    """one oracle episode with the hashed random policy of rs_act_random; returns what the writer needs"""
    from oracle.pyoracle import OracleEnv
    from resco_amd.scenario import Scenario
    sys.path.insert(0, os.path.join(ROOT, 'tests'))
    from conftest import preroll_actions
    sc = Scenario.load(os.path.join(ROOT, 'resco_amd', 'scenarios', name + '.npz'))
    env = OracleEnv(sc, env_index=0, seed=seed, sigma=-1.0, speed_dev=1, max_distance=max_distance, trip_log=1, tls_expiry=TLS_EXPIRY)
    env.observe()
    for k in range(360):
        env.step(preroll_actions(sc, seed, 0, k))
    return sc, env


def main():
    from oracle import ref_harness                      # installs the traci / sumolib / gym stubs and SUMO_HOME
    ref_harness.install_stubs(lambda cmd: None)       # readXML only imports resco_benchmark.config.map_config
    from resco_amd.multi_signal import tripinfo_records, write_tripinfo
    plt = types.ModuleType('matplotlib.pyplot')
    plt.title = plt.plot = plt.show = lambda *a, **k: None
    mpl = types.ModuleType('matplotlib')
    mpl.use = lambda *a, **k: None
    mpl.pyplot = plt
    sys.modules['matplotlib'], sys.modules['matplotlib.pyplot'] = mpl, plt
    if REF not in sys.path:
        sys.path.insert(0, REF)
    for name, policy, md, seed in CASES:
        sc, env = episode(name, policy, md, seed)
        v = env.vehicles()
        recs = tripinfo_records(sc, env.trip_log(), env.time, v['lane'], v['trip'], v['depart'], v['time_loss'], env.wtot(),
                                seed, 0, 1)
        tmp = tempfile.mkdtemp()
        run_dir = 'fixture-tr0-%s-0-drq-wait' % name
        write_tripinfo(os.path.join(tmp, 'results', run_dir, 'tripinfo_1.xml'), recs)
        cwd = os.path.join(tmp, 'resco_benchmark', 'utils')
        os.makedirs(cwd)
        os.symlink(os.path.join(REF, 'resco_benchmark', 'environments'), os.path.join(tmp, 'resco_benchmark', 'environments'))
        old = os.getcwd()
        os.chdir(cwd)
        try:
            runpy.run_path(os.path.join(REF, 'resco_benchmark', 'utils', 'readXML.py'), run_name='__main__')
        finally:
            os.chdir(old)
        out = {}
        for metric in ('timeLoss', 'duration', 'waitingTime'):
            with open(os.path.join(cwd, 'avg_%s.py' % metric)) as f:
                line = f.readline()
            key, val = line.split(':', 1)
            out[metric] = json.loads(val.strip().rstrip(','))[0]
        st = env.stats()
        waited, n_wait = env.backlog_delay()
        fx = dict(map=name, seed=seed, max_distance=md, tls_expiry=TLS_EXPIRY, policy='rs_act_random / conftest.preroll_actions', entries=len(recs),
                  arrived=st['arrived'], queued_never_departed=n_wait, readXML=out)
        with open(os.path.join(HERE, 'readxml_%s.json' % name), 'w') as f:
            json.dump(fx, f, indent=1)
        print(json.dumps(fx))


if __name__ == '__main__':
    main()
