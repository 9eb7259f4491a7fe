#!/bin/bash
# Arbitrary PMC counter sets for the step kernel, one rocprofv3 run per set (counters only with --kernel-trace):
#   bash tools/pmc_sets.sh <outdir under gpurun_out> <steps> <warmup> "CTR_A CTR_B ..." "CTR_C ..." ...
# Prints and stores (<outdir>/sets_summary.json) the per-launch average of every counter over the rs_step_kernel launches.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/$1; STEPS=$2; WARM=$3; shift 3
CMD="python $GRAFT_REPO_ROOT/bench.py --steps $STEPS --warmup $WARM --no-cpu-baseline ${BENCH_ARGS:-}"
mkdir -p $OUT
i=0
for set in "$@"; do
  i=$((i+1))
  timeout 300 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/s$i -o s$i -- $CMD > $OUT/s$i.log 2>&1
  echo "set $i rc=$? : $set"
done
python - <<PY
import csv, glob, collections, json
summary = {}
for f in sorted(glob.glob('$OUT/s*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'rs_step_kernel' not in r.get('Kernel_Name', ''): continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print('%-40s per-launch avg %.5g  (%d launches)' % (k, v / max(1, n), n))
        summary[k] = v / max(1, n)
json.dump(dict(command='$CMD', counters=summary), open('$OUT/sets_summary.json', 'w'), indent=1)
PY
