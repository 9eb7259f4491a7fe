#!/usr/bin/env python3
"""A/B builds of the HIP library: compile one variant per -D switch set HERE (hipcc cross-compiles), then time them all
back to back in ONE gpurun call on the same box.
  python tools/ab.py build name1:"-DX=1 -DY" name2:""      -> variants/name.so
  python tools/ab.py run [--envs N] [--block B] [--steps K]  (on the GPU box: every variants/*.so + the default library)"""
import glob, json, os, subprocess, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
VAR = os.path.join(ROOT, 'variants')


def build(specs):
    from resco_amd.build import HIPCC, FLAGS, SRC
    os.makedirs(VAR, exist_ok=True)
    procs = []
    for spec in specs:
        name, _, flags = spec.partition(':')
        out = os.path.join(VAR, name + '.so')
        procs.append((name, subprocess.Popen([HIPCC] + FLAGS + flags.split() + [SRC, '-o', out])))
    for name, p in procs:
        print(name, 'rc', p.wait())


def run(envs, blocks, steps, warm, mapname):
    libs = sorted(glob.glob(os.path.join(VAR, '*.so')))
    for lib in libs:
        for block in blocks:
            code = ("import sys, json; sys.path.insert(0, %r); from tools.gpu_check import timing; "
                    "print(json.dumps(timing(%r, %d, %d, steps=%d, warm=%d)))" % (ROOT, mapname, envs, block, steps, warm))
            env = dict(os.environ, RESCO_SIM_LIB=lib)
            out = subprocess.run([sys.executable, '-c', code], env=env, capture_output=True, text=True)
            try:
                r = json.loads(out.stdout.strip().splitlines()[-1])
                print('%-28s block %5d  %8.0f env-steps/s  kernel %.3f ms  lds %d  V %.0f' % (os.path.basename(lib), block, r['env_steps_per_s'], r['kernel_ms'], r['lds'], r['mean_active']), flush=True)
            except Exception:
                print(os.path.basename(lib), block, 'FAILED', out.stderr[-300:], flush=True)


if __name__ == '__main__':
    if sys.argv[1] == 'build':
        build(sys.argv[2:])
    else:
        import argparse
        ap = argparse.ArgumentParser()
        ap.add_argument('cmd'); ap.add_argument('--envs', type=int, default=4096); ap.add_argument('--blocks', default='-512')
        ap.add_argument('--steps', type=int, default=120); ap.add_argument('--warm', type=int, default=100); ap.add_argument('--map', default='ingolstadt21')
        a = ap.parse_args()
        run(a.envs, [int(b) for b in a.blocks.split(',')], a.steps, a.warm, a.map)
