"""signal_configs[map] = {phase_pairs, valid_acts, <signal id>: {lane_sets, downstream}}.

The data is the reference's per-map signal configuration (resco_benchmark/config/signal_config.py),
re-serialised to signal_configs.json by tools/export_reference_data.py and consumed as-is, quirks
included (e.g. ingolstadt21 '89173763' lists itself as its own S downstream).
"""
import json
import os

_PATH = os.path.join(os.path.dirname(os.path.abspath(__file__)), 'signal_configs.json')


def _load():
    with open(_PATH) as f:
        raw = json.load(f)
    out = {}
    for m, cfg in raw.items():
        dec = {}
        for k, v in cfg.items():
            if k == 'valid_acts':
                dec[k] = None if v is None else {sid: {int(a): int(b) for a, b in pairs} for sid, pairs in v.items()}
            else:
                dec[k] = v
        out[m] = dec
    return out


signal_configs = _load()
