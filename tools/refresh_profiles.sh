#!/bin/bash
# Regenerate the measured evidence of the current build on the MI355X box (run through gpurun); outputs land in
# gpurun_out/$TAG/ and are copied into profiles/ (named per round) by hand afterwards.
#   gpurun --timeout 1200 -- 'TAG=r02 bash tools/refresh_profiles.sh'
set -u
R=$GRAFT_REPO_ROOT
TAG=${TAG:-final}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_default.log 2>/dev/null
tail -1 $OUT/bench_default.log > $OUT/bench_default.json
python tools/phase_profile.py ingolstadt21 4096 0 > $OUT/phase_profile.txt 2>/dev/null
python tools/phase_profile.py ingolstadt21 256 0 > $OUT/phase_profile_one_workgroup_per_cu.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# the SAME command as the contract line (default --steps / --warmup), CPU baseline off: per-kernel time by rocprofv3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
cd $R && bash tools/pmc_passes.sh $TAG/pmc 300 60 > $OUT/pmc_passes.log 2>&1
[ "${DIAG:-1}" = 1 ] && bash tools/pmc_diag.sh $TAG/pmcdiag > $OUT/pmc_diag.log 2>&1
head -5 $OUT/prof/${TAG}_kernel_stats.csv
tail -3 $OUT/pmc_passes.log
cut -c1-300 $OUT/bench_default.json
