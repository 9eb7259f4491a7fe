"""mdp_configs[agent][map]: normalisation constants and the manager hierarchy the FMA2C state / reward
functions read (data of the reference's resco_benchmark/config/mdp_config.py, re-serialised to
mdp_configs.json by tools/export_reference_data.py).

The reference's driver rewrites the module-level dict before building the environment
(main.py:48-72): ``mdp_configs[agent]`` is replaced by the entry of the chosen map and gets a derived
``supervisors`` reverse map.  ``activate`` does the same, so states.fma2c / rewards.fma2c find
``mdp_configs['FMA2C']`` in the shape they expect.
"""
import copy
import json
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'mdp_configs.json')
with open(_PATH) as _f:
    _ALL = json.load(_f)

mdp_configs = copy.deepcopy(_ALL)


def activate(agent, map_name):
    """Select the per-map entry of `agent` ('FMA2C', 'FMA2CFull', 'FMA2CVAL') and derive `supervisors`."""
    cfg = copy.deepcopy(_ALL[agent][map_name])
    management = cfg.get('management')
    if management is not None:
        cfg['supervisors'] = {worker: manager for manager, workers in management.items() for worker in workers}
    mdp_configs[agent] = cfg
    return cfg
