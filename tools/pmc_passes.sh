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
import csv, glob, collections, json, sys
sys.path.insert(0, '$GRAFT_REPO_ROOT')
from bench import kernel_source_hash, DEFAULT_PIPES
STEPS, PIPES = ${2:-30}, DEFAULT_PIPES
summary = {}
for f in sorted(glob.glob('$OUT/p*/*counter_collection.csv')):
    per = collections.defaultdict(list)                 # counter -> [(dispatch id, value)]
    for r in csv.DictReader(open(f)):
        if 'rs_step_kernel' not in r.get('Kernel_Name', ''): continue
        per[r['Counter_Name']].append((int(r['Dispatch_Id']), float(r['Counter_Value'])))
    for k, rows in sorted(per.items()):
        # a counter may be reported in several rows per dispatch (one per XCD / instance): add them up per dispatch first
        by = collections.OrderedDict()
        for d, v in sorted(rows):
            by[d] = by.get(d, 0.0) + v
        vals = list(by.values())
        timed = vals[-PIPES * STEPS:]         # the timed window is the end of the run
        print(f.split('/')[-2], k, 'per-launch avg over the %d timed launches' % len(timed), sum(timed) / max(1, len(timed)), '(all %d launches: %s)' % (len(vals), sum(vals) / max(1, len(vals))))
        summary[k] = dict(per_launch_avg=sum(timed) / max(1, len(timed)), launches=len(timed), all_launches=len(vals), all_launches_avg=sum(vals) / max(1, len(vals)))
json.dump(dict(command='$CMD', kernel='rs_step_kernel', source_hash=kernel_source_hash(), pipes=PIPES, counters=summary,
               note='per_launch_avg: over the timed launches of the command (steps x pipes launches of 4096 / pipes environments each; the fast-forward and warm-up launches are left out).  FETCH_SIZE/WRITE_SIZE are in KiB; on gfx950 FETCH_SIZE reports half of the bytes of wide coalesced reads (MI355X_MICROARCH.md, HBM)'),
          open('$OUT/pmc_summary.json', 'w'), indent=1)
PY
