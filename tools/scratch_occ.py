import sys, json, os
sys.path.insert(0, '.')
from tools.gpu_check import timing, parity
os.environ['RS_CAPACITY'] = os.environ.get('CAP', '896')
for b in [int(x) for x in os.environ.get('BLOCKS', '512,-10512').split(',')]:
    for (st, wm) in ((20, 170), (120, 100)):
        r = timing('ingolstadt21', 4096, b, steps=st, warm=wm)
        print(os.environ.get('RESCO_SIM_LIB', 'default'), 'cap', os.environ['RS_CAPACITY'], 'block', b, 'steps %d+%d' % (wm, st), '%.3fM env-steps/s kernel %.3f ms lds %d V %.0f' % (r['env_steps_per_s'] / 1e6, r['kernel_ms'], r['lds'], r['mean_active']), flush=True)
