#!/bin/bash
R=$GRAFT_REPO_ROOT
cd /tmp && export TMPDIR=/tmp
RESCO_SIM_LIB=$R/variants/r_two.so python -m pytest $R/tests/test_gpu_parity.py -q -x -k "idqn or fused or config5 or group_step" 2>&1 | tail -2
for l in q_r16 r_two; do
for nk in "1024 1" "4096 1"; do
  RESCO_SIM_LIB=$R/variants/$l.so rocprofv3 --kernel-trace --stats --output-format csv -d /tmp/pp_$l -o t -- python $R/tools/scratch/c5.py $nk idqn > /tmp/pp.log 2>&1
  echo -n "$l $nk"; python - <<PY
import csv, glob
for f in glob.glob('/tmp/pp_$l/*kernel_stats.csv') + glob.glob('/tmp/pp_$l/*/*kernel_stats.csv'):
    for r in csv.DictReader(open(f)):
        if 'idqn' in r['Name']:
            print('   ', r['Name'][:30], 'calls', r['Calls'], 'avg us %.1f' % (float(r['AverageNs']) / 1e3), 'min %.1f max %.1f' % (float(r['MinNs']) / 1e3, float(r['MaxNs']) / 1e3))
PY
  rm -rf /tmp/pp_$l
done
done
for rep in 1 2; do
for nk in "1024 4" "1024 8" "4096 2"; do
  python $R/tools/scratch/c5.py $nk random 2>/dev/null | tail -1
  for l in q_r16 r_two; do
  echo -n "$l  "; RESCO_SIM_LIB=$R/variants/$l.so python $R/tools/scratch/c5.py $nk idqn 2>/dev/null | tail -1
  done
done
done
