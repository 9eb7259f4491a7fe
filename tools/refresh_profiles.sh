#!/bin/bash
# Regenerate the measured evidence of the current build on the MI355X box (run through gpurun); outputs land
# in gpurun_out/final/ and are copied into profiles/ by hand afterwards.
set -u
R=$GRAFT_REPO_ROOT
OUT=$R/gpurun_out/final
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_default.log 2>/dev/null
tail -1 $OUT/bench_default.log > $OUT/bench_default.json
python tools/bench_configs.py > $OUT/bench_configs.jsonl 2>/dev/null
python tools/phase_profile.py ingolstadt21 4096 512 > $OUT/phase_profile.txt 2>/dev/null
python tools/idqn_rollout.py 1024 2>/dev/null | tail -1 > $OUT/idqn_rollout.jsonl
python tools/idqn_rollout.py 4096 2>/dev/null | tail -1 >> $OUT/idqn_rollout.jsonl
python tools/policy_eval.py > $OUT/policy_eval.jsonl 2>/dev/null
cd /tmp && export TMPDIR=/tmp
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o final -- python $R/bench.py --steps 360 --warmup 0 --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
cd $R && bash tools/pmc_passes.sh final/pmc 100 60 > $OUT/pmc_passes.log 2>&1
cat $OUT/prof/final_kernel_stats.csv
tail -3 $OUT/pmc_passes.log
cut -c1-200 $OUT/bench_default.json
