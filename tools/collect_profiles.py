#!/usr/bin/env python3
"""Copy the evidence tools/refresh_profiles.sh left under gpurun_out/<tag>/ into profiles/ (tracked), named per round:
  python tools/collect_profiles.py r03"""
import csv
import glob
import json
import os
import shutil
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
tag = sys.argv[1] if len(sys.argv) > 1 else 'r06'
PIPES = 2          # bench.py's default: launches per step
src = os.path.join(ROOT, 'gpurun_out', tag)
dst = os.path.join(ROOT, 'profiles')


def cp(a, b):
    if os.path.exists(os.path.join(src, a)):
        shutil.copy(os.path.join(src, a), os.path.join(dst, '%s_%s' % (tag, b)))
        print('profiles/%s_%s' % (tag, b))


for a, b in (('bench_default.json', 'bench_default.json'), ('bench_driver_window.json', 'bench_driver_window.json'),
             ('bench_under_rocprof.json', 'bench_under_rocprof.json'), ('phase_profile.txt', 'phase_profile.txt'),
             ('phase_profile_one_workgroup_per_cu.txt', 'phase_profile_one_workgroup_per_cu.txt'),
             ('pipes_ab.jsonl', 'pipes_ab.jsonl'), ('heldout_idqn.txt', 'heldout_idqn.txt'), ('bench_configs.jsonl', 'bench_configs.jsonl'),
             ('reference_bands.txt', 'reference_bands.txt'), ('tls_expiry_bands.txt', 'tls_expiry_bands.txt'), ('reference_bands_both_modes.txt', 'reference_bands_both_modes.txt'),
             ('heldout_both_modes.txt', 'heldout_both_modes.txt'), ('bench_default_hold.json', 'bench_default_hold.json'), ('bench_driver_window_hold.json', 'bench_driver_window_hold.json'), ('idqn_rollout.jsonl', 'idqn_rollout.jsonl'), ('prof/%s_kernel_stats.csv' % tag, 'kernel_stats.csv'),
             ('prof_dw/%s_dw_kernel_stats.csv' % tag, 'driver_window_kernel_stats.csv'),
             ('pmc_s300_w60/pmc_summary.json', 'pmc_s300_w60.json'), ('pmc_s20_w5/pmc_summary.json', 'pmc_s20_w5.json'),
             ('pmcdiag/diag_summary.json', 'pmc_diag.json')):
    cp(a, b)
# kernel-trace summary of the step kernel over the timed launches of the default command
for name, out in (('prof/%s_kernel_trace.csv' % tag, 'kernel_trace_summary.txt'), ('prof_dw/%s_dw_kernel_trace.csv' % tag, 'driver_window_kernel_trace_summary.txt')):
    p = os.path.join(src, name)
    if not os.path.exists(p):
        continue
    rows = [r for r in csv.DictReader(open(p)) if 'rs_step_kernel' in r.get('Kernel_Name', '')]
    rows.sort(key=lambda r: int(r['Start_Timestamp']))
    dur = [(int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e6 for r in rows]
    steps = 20 if '_dw' in name else 300
    timed = rows[-PIPES * steps:]           # the timed window is the end of the run
    d = dur[-PIPES * steps:]
    span = (max(int(r['End_Timestamp']) for r in timed) - min(int(r['Start_Timestamp']) for r in timed)) / 1e6
    with open(os.path.join(dst, '%s_%s' % (tag, out)), 'w') as f:
        f.write('rocprofv3 --kernel-trace of `%s`\n' % ('python bench.py --no-cpu-baseline' + (' --steps 20 --warmup 5' if '_dw' in name else '')))
        f.write('rs_step_kernel launches: %d (%d per step, one per pipe: fast-forward + warm-up + timed; the first %d are the reset observes)\n' % (len(dur), PIPES, PIPES))
        f.write('the %d timed launches (%d steps x %d pipes of 2048 environments): mean %.4f ms, min %.4f, max %.4f\n' % (len(d), steps, PIPES, sum(d) / len(d), min(d), max(d)))
        f.write('first start to last end of the timed launches: %.3f ms = %.4f ms per step (the launches of the two pipes overlap)\n' % (span, span / steps))
        f.write('all launches: mean %.4f ms\n' % (sum(dur) / len(dur)))
    print('profiles/%s_%s' % (tag, out))
res = subprocess.run(['bash', os.path.join(ROOT, 'tools', 'kernel_resources.sh')], capture_output=True, text=True).stdout
with open(os.path.join(dst, '%s_kernel_resources.txt' % tag), 'w') as f:
    f.write('hipcc -Rpass-analysis=kernel-resource-usage (tools/kernel_resources.sh), gfx950, build flags of resco_amd/build.py\n' + res)
print('profiles/%s_kernel_resources.txt' % tag)
