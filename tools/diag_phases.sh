#!/bin/bash
# Per-phase instruction accounting: a -DRS_DIAG build (variants/diag.so) skips parts of the tick loop on the
# measured launches only (after the warm-up built the real traffic state); the difference of the PMC counters
# between masks is what that part costs.  masks: 1 = approach registration (A), 2 = look-ahead hop loop (C),
# 4 = lane-change evaluation (E), 8 = foe check, 16 = rearmost-vehicle search on the next lane, 32 = hops after the first.
set -u
cd /tmp && export TMPDIR=/tmp
OUT=$GRAFT_REPO_ROOT/gpurun_out/${1:-diag}
mkdir -p $OUT
cp $GRAFT_REPO_ROOT/variants/diag.so $GRAFT_REPO_ROOT/resco_amd/csrc/libresco_sim.so
CMD="python $GRAFT_REPO_ROOT/bench.py --steps 30 --warmup 100 --no-cpu-baseline"
for mask in ${MASKS:-0 1 2 4 7}; do
  RS_DIAG_SKIP=$mask RS_DIAG_AFTER=${AFTER:-101} timeout 200 rocprofv3 --pmc SQ_INSTS_VALU SQ_THREAD_CYCLES_VALU SQ_INSTS_SALU SQ_WAVE_CYCLES SQ_INSTS_LDS SQ_INSTS_VMEM_RD SQ_WAIT_ANY SQ_INSTS_BRANCH \
     --kernel-trace --output-format csv -d $OUT/m$mask -o m -- $CMD > $OUT/m$mask.log 2>&1
  echo "mask $mask rc=$?"
done
python - <<PY
import csv, glob, collections
for mask in [int(x) for x in '${MASKS:-0 1 2 4 7}'.split()]:
    rows = collections.defaultdict(dict)
    for f in glob.glob('$OUT/m%d/*counter_collection.csv' % mask):
        for r in csv.DictReader(open(f)):
            if 'rs_step_kernel' in r['Kernel_Name']:
                rows[int(r['Dispatch_Id'])][r['Counter_Name']] = float(r['Counter_Value'])
    dur = {}
    for f in glob.glob('$OUT/m%d/*kernel_trace.csv' % mask):
        for r in csv.DictReader(open(f)):
            if 'rs_step_kernel' in r['Kernel_Name']:
                dur[int(r['Dispatch_Id'])] = (int(r['End_Timestamp']) - int(r['Start_Timestamp'])) / 1e3
    ids = sorted(rows)[-${LAST:-30}:]
    avg = lambda k: sum(rows[i].get(k, 0.0) for i in ids) / len(ids)
    v, t = avg('SQ_INSTS_VALU'), avg('SQ_THREAD_CYCLES_VALU')
    print('mask %d  us %.0f  VALU %.1fM  lanes/instr %.1f  SALU %.1fM  LDS %.1fM  VMEM_RD %.1fM  BRANCH %.1fM  wave_cyc %.2fG  wait %.2fG' % (
        mask, sum(dur.get(i, 0) for i in ids) / len(ids), v / 1e6, t / max(v, 1), avg('SQ_INSTS_SALU') / 1e6, avg('SQ_INSTS_LDS') / 1e6,
        avg('SQ_INSTS_VMEM_RD') / 1e6, avg('SQ_INSTS_BRANCH') / 1e6, avg('SQ_WAVE_CYCLES') / 1e9, avg('SQ_WAIT_ANY') / 1e9))
PY
