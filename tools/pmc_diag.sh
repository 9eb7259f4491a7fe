#!/bin/bash
# Diagnostic PMC passes (issue mix, divergence, instruction fetch, queue depths) for the step kernel.
# Counters only with --kernel-trace, one set per run (see tools/pmc_passes.sh).
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmcdiag}
CMD="python $GRAFT_REPO_ROOT/bench.py --steps ${2:-300} --warmup ${3:-60} --no-cpu-baseline"
mkdir -p $OUT
i=0
for set in "SQ_ACTIVE_INST_VALU SQ_ACTIVE_INST_SCA SQ_ACTIVE_INST_LDS SQ_ACTIVE_INST_VMEM SQ_ACTIVE_INST_MISC SQ_THREAD_CYCLES_VALU SQ_BUSY_CU_CYCLES SQ_CYCLES" \
           "SQ_IFETCH SQ_IFETCH_LEVEL SQC_ICACHE_REQ SQC_ICACHE_HITS SQC_ICACHE_MISSES SQ_INSTS_BRANCH SQ_INST_CYCLES_SALU SQ_INSTS_VALU_TRANS_F32" \
           "SQ_INST_LEVEL_VMEM SQ_INST_LEVEL_LDS SQ_INSTS_LDS_ATOMIC SQ_LDS_ADDR_CONFLICT SQ_LDS_ATOMIC_RETURN SQ_INSTS_VALU_INT32 SQ_INSTS_VALU_FMA_F32 SQ_INSTS_VALU_ADD_F32"; do
  i=$((i+1))
  timeout 200 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
python - <<PY
import csv, glob, collections, json
summary = {}
for f in sorted(glob.glob('$OUT/p*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'rs_step_kernel' not in r.get('Kernel_Name', ''): continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(k, 'per-launch avg %.4g' % (v / max(1, n)), 'launches', n)
        summary[k] = v / max(1, n)
json.dump(summary, open('$OUT/diag_summary.json', 'w'), indent=1)
PY
