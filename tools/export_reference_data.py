#!/usr/bin/env python3
"""Build-container tool: turn the reference's scenario DATA into the files this package ships.

Runs only where /root/reference exists (never on the GPU box).  Produces
  resco_amd/config/signal_configs.json   the per-map signal_configs dict (data; hot maps)
  resco_amd/config/mdp_configs.json      the FMA2C normalisation constants / manager hierarchy (data)
  resco_amd/scenarios/<map>.npz          compiled flat tables (resco_amd.scenario.compile_scenario)

No reference source text is copied: signal_config.py is *executed* as data and its dict is
re-serialised; net.xml / rou.xml are parsed and compiled to index tables.
"""
import importlib.util
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
REF = os.environ.get('RESCO_REFERENCE', '/root/reference')
MAPS = ['cologne1', 'cologne3', 'cologne8', 'ingolstadt1', 'ingolstadt7', 'ingolstadt21']
# vehicle slots per environment where the compiler's rule (next power of two above 4 x the free-flow concurrency) is overridden.
# ingolstadt21: the rule gives 1024; 896 is the largest capacity whose working memory (40 768 B) lets FOUR workgroups share a CU
# of the MI355X (+14 % env-steps/s, profiles/r06_ab_occupancy.txt).  Peak number of vehicles on the network over whole episodes
# (oracle, four environments each): FIXED 590, STOCHASTIC 610-740, MAXWAVE / MAXPRESSURE with the repaired valid_acts 340 / 570;
# the as-configured MAXWAVE / MAXPRESSURE cells (known gaps: one approach never gets green) reach 870 / 1024 -- they ran into the
# limit of 1024 as well.  Trips that find the network full wait in their backlog and are counted (stats: `cap_blocked`).
CAPACITY = {'ingolstadt21': 896}


def load_ref_module(rel, name):
    spec = importlib.util.spec_from_file_location(name, os.path.join(REF, rel))
    m = importlib.util.module_from_spec(spec)
    spec.loader.exec_module(m)
    return m


def main():
    from resco_amd.scenario import compile_from_sumocfg
    sc_mod = load_ref_module('resco_benchmark/config/signal_config.py', '_ref_signal_config')
    mc_mod = load_ref_module('resco_benchmark/config/map_config.py', '_ref_map_config')
    out = {}
    for m in MAPS:
        cfg = sc_mod.signal_configs[m]
        enc = {}
        for k, v in cfg.items():
            if k == 'valid_acts':
                enc[k] = None if v is None else {sid: [[int(a), int(b)] for a, b in d.items()] for sid, d in v.items()}
            else:
                enc[k] = v
        out[m] = enc
    with open(os.path.join(ROOT, 'resco_amd', 'config', 'signal_configs.json'), 'w') as f:
        json.dump(out, f, separators=(',', ':'))
    md_mod = load_ref_module('resco_benchmark/config/mdp_config.py', '_ref_mdp_config')
    mdp = {}
    for agent, per_map in md_mod.mdp_configs.items():
        mdp[agent] = {m: per_map[m] for m in MAPS if m in per_map}
    with open(os.path.join(ROOT, 'resco_amd', 'config', 'mdp_configs.json'), 'w') as f:
        json.dump(mdp, f, separators=(',', ':'))
    for m in MAPS:
        mc = mc_mod.map_configs[m]
        cfgpath = os.path.join(REF, 'resco_benchmark', mc['net'])
        sc = compile_from_sumocfg(m, cfgpath, sc_mod.signal_configs[m], lights=mc['lights'],
                                  yellow_length=mc['yellow_length'], capacity=CAPACITY.get(m))
        sc.save(os.path.join(ROOT, 'resco_amd', 'scenarios', m + '.npz'))
        print(m, 'lanes', sc.n_lanes, 'links', sc.n_links, 'edges', sc.n_edges, 'routes', sc.n_routes,
              'trips', sc.n_trips, 'dropped', sc.dropped_trips, 'signals', sc.n_signals, 'obs', sc.n_obs,
              'capacity', sc.capacity, 'unmapped obs lanes', int((sc.obs_lane < 0).sum()))


if __name__ == '__main__':
    main()
