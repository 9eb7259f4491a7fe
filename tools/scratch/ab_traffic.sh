#!/bin/bash
# A/B of library variants: bench line (driver window) twice each, interleaved, then FETCH_SIZE / WRITE_SIZE of the timed launches
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/abt; mkdir -p $OUT
cd /tmp && export TMPDIR=/tmp
LIBS="$@"
for rep in 1 2; do
  for l in $LIBS; do
    RESCO_SIM_LIB=$R/variants/$l.so python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$l rep $rep dw %.0f  kernel %.4f ms  alg %.0f B  all_outputs %.0f' % (d['value'], d['roofline']['kernel_avg_ms'], d['roofline']['algorithmic_bytes_per_env_step'], d['all_outputs']['value']))"
  done
done
for l in ; do
  RESCO_SIM_LIB=$R/variants/$l.so python $R/bench.py --no-cpu-baseline 2>/dev/null | tail -1 | python -c "
import sys, json; d = json.loads(sys.stdin.read()); print('$l default %.0f  kernel %.4f ms' % (d['value'], d['roofline']['kernel_avg_ms']))"
done
for l in ; do
  for c in FETCH_SIZE WRITE_SIZE "TCC_HIT_sum TCC_MISS_sum"; do
    n=$(echo $c | tr ' ' '_')
    RESCO_SIM_LIB=$R/variants/$l.so timeout 240 rocprofv3 --pmc $c --kernel-trace --output-format csv -d $OUT/${l}_$n -o p -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/$l.$n.log 2>&1
    python - <<PY
import csv, glob, collections
for f in sorted(glob.glob('$OUT/${l}_$n/*counter_collection.csv')) + sorted(glob.glob('$OUT/${l}_$n/*/*counter_collection.csv')):
    per = collections.defaultdict(dict)
    for r in csv.DictReader(open(f)):
        if 'rs_step_kernel' not in r.get('Kernel_Name', ''): continue
        d = int(r['Dispatch_Id']); per[r['Counter_Name']][d] = per[r['Counter_Name']].get(d, 0.0) + float(r['Counter_Value'])
    for k, by in sorted(per.items()):
        vals = [by[d] for d in sorted(by)][-40:]
        print('$l', k, 'per launch over the 40 timed launches: %.1f' % (sum(vals) / len(vals)))
PY
  done
done
