"""map_configs: the per-map run parameters MultiSignal is constructed from.

Same keys as the reference's resco_benchmark/config/map_config.py (lights, net, route, step_length,
yellow_length, step_ratio, start_time, end_time, warmup) for the maps whose demand files ship with the
reference.  `net` keeps the reference's relative path (resolved against --pwd when the RESCO environment
directory is present); when it is not, MultiSignal falls back to the pre-compiled tables in
resco_amd/scenarios/<map>.npz.
"""


def _entry(name, start, lights=()):
    return {
        'lights': list(lights),
        'net': 'environments/%s/%s.sumocfg' % (name, name),
        'route': None,
        'step_length': 10,
        'yellow_length': 3,
        'step_ratio': 1,
        'start_time': start,
        'end_time': start + 3600,
        'warmup': 0,
    }


_INGOLSTADT7 = ['cluster_1757124350_1757124352', 'gneJ143', 'gneJ207',
                'cluster_306484187_cluster_1200363791_1200363826_1200363834_1200363898_1200363927_1200363938_'
                '1200363947_1200364074_1200364103_1507566554_1507566556_255882157_306484190',
                '32564122', 'gneJ260', 'gneJ210']

_INGOLSTADT21 = ['1863241632', '2330725114', '243351999', '243641585', '243749571', '30503246', '30624898',
                 '32564122', '89127267', '89173763', '89173808', 'cluster_1427494838_273472399',
                 'cluster_1757124350_1757124352', 'cluster_1863241547_1863241548_1976170214',
                 'cluster_306484187_cluster_1200363791_1200363826_1200363834_1200363898_1200363927_1200363938_'
                 '1200363947_1200364074_1200364103_1507566554_1507566556_255882157_306484190',
                 'gneJ143', 'gneJ207', 'gneJ208', 'gneJ210', 'gneJ255', 'gneJ257']

map_configs = {
    'ingolstadt1': _entry('ingolstadt1', 57600),
    'ingolstadt7': _entry('ingolstadt7', 57600, _INGOLSTADT7),
    'ingolstadt21': _entry('ingolstadt21', 57600, _INGOLSTADT21),
    'cologne1': _entry('cologne1', 25200),
    'cologne3': _entry('cologne3', 25200),
    'cologne8': _entry('cologne8', 25200),
}
