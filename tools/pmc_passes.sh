#!/bin/bash
# rocprofv3 PMC passes for the step kernel (run on the MI355X box via gpurun).  Counters are collected in
# their own runs with --kernel-trace only (never with sys/hip/hsa tracing).
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-pmc}
CMD="python $GRAFT_REPO_ROOT/bench.py --steps ${2:-30} --warmup ${3:-100} --no-cpu-baseline"
mkdir -p $OUT
i=0
for set in "SQ_WAVE_CYCLES SQ_WAIT_ANY SQ_WAIT_INST_ANY SQ_ACTIVE_INST_ANY SQ_INSTS_VALU SQ_INSTS_LDS SQ_INSTS_SALU SQ_WAVES" \
           "SQ_WAIT_INST_LDS SQ_LDS_BANK_CONFLICT SQ_LDS_IDX_ACTIVE SQ_INSTS_SMEM SQ_BUSY_CYCLES SQ_INSTS_VMEM_RD SQ_INSTS_VMEM_WR GRBM_GUI_ACTIVE" \
           "FETCH_SIZE" "WRITE_SIZE" "TCC_HIT_sum TCC_MISS_sum"; do
  i=$((i+1))
  timeout 240 rocprofv3 --pmc $set --kernel-trace --output-format csv -d $OUT/p$i -o p$i -- $CMD > $OUT/p$i.log 2>&1
  echo "pass $i rc=$? : $set"
done
python - <<PY
import csv, glob, collections
summary = {}
for f in sorted(glob.glob('$OUT/p*/*counter_collection.csv')):
    acc = collections.defaultdict(lambda: [0.0, 0])
    for r in csv.DictReader(open(f)):
        if 'rs_step_kernel' not in r.get('Kernel_Name', ''): continue
        a = acc[r['Counter_Name']]; a[0] += float(r['Counter_Value']); a[1] += 1
    for k, (v, n) in sorted(acc.items()):
        print(f.split('/')[-2], k, 'per-launch avg', v / max(1, n), 'launches', n)
        summary[k] = dict(per_launch_avg=v / max(1, n), launches=n)
import json, sys
sys.path.insert(0, '$GRAFT_REPO_ROOT')
from bench import kernel_source_hash
json.dump(dict(command='$CMD', kernel='rs_step_kernel', source_hash=kernel_source_hash(), counters=summary,
               note='FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)'),
          open('$OUT/pmc_summary.json', 'w'), indent=1)
PY
