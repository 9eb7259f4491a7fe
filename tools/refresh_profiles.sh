#!/bin/bash
# Regenerate the measured evidence of the current build on the MI355X box (run through gpurun); outputs land in
# gpurun_out/$TAG/ and are copied into profiles/ (named per round) afterwards (tools/collect_profiles.py).
#   gpurun --timeout 2400 -- 'TAG=r05 bash tools/refresh_profiles.sh'      (HELDOUT=0 skips the four-minute IDQN trainings)
set -u
R=$GRAFT_REPO_ROOT
TAG=${TAG:-final}
OUT=$R/gpurun_out/$TAG
mkdir -p $OUT
cd $R
python bench.py > $OUT/bench_default.log 2>/dev/null
tail -1 $OUT/bench_default.log > $OUT/bench_default.json
python bench.py --steps 20 --warmup 5 > $OUT/bench_driver_window.log 2>/dev/null
tail -1 $OUT/bench_driver_window.log > $OUT/bench_driver_window.json
# the other answer to what setPhase leaves behind (round 5's default): both windows
python bench.py --tls-expiry 0 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_default_hold.json
python bench.py --tls-expiry 0 --steps 20 --warmup 5 --no-cpu-baseline 2>/dev/null | tail -1 > $OUT/bench_driver_window_hold.json
python tools/phase_profile.py ingolstadt21 4096 0 > $OUT/phase_profile.txt 2>/dev/null
python tools/phase_profile.py ingolstadt21 256 0 > $OUT/phase_profile_one_workgroup_per_cu.txt 2>/dev/null
python tools/bench_configs.py > $OUT/bench_configs.jsonl 2>/dev/null
rm -f $OUT/pipes_ab.jsonl $OUT/idqn_rollout.jsonl
python tools/pipes_ab.py --no-rollout --out $OUT/pipes_ab.jsonl > /dev/null 2>&1
python tools/pipes_ab.py --rollout-only --out $OUT/idqn_rollout.jsonl > /dev/null 2>&1
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_reference_result_bands or test_reference_result_known_gaps" -s 2>&1 | grep -o "band .*\|[0-9]* passed.*\|[0-9]* failed.*" > $OUT/reference_bands.txt
python -m pytest tests/test_gpu_parity.py -m gpu -q -k "test_tls_expiry_evidence" -s 2>&1 | grep -o "expiry .*\|[0-9]* passed.*\|[0-9]* failed.*" > $OUT/tls_expiry_bands.txt
[ "${HELDOUT:-1}" = 1 ] && python -m pytest tests/test_gpu_heldout.py -m gpu -q -s 2>&1 | grep -o "heldout .*\|[0-9]* passed.*\|[0-9]* failed.*" > $OUT/heldout_idqn.txt
python tools/both_modes.py > $OUT/reference_bands_both_modes.txt 2>/dev/null
[ "${HELDOUT:-1}" = 1 ] && python tools/heldout_both_modes.py > $OUT/heldout_both_modes.txt 2>/dev/null
cd /tmp && export TMPDIR=/tmp
# the SAME command as the contract line (default --steps / --warmup), CPU baseline off: per-kernel time by rocprofv3
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof -o $TAG -- python $R/bench.py --no-cpu-baseline > $OUT/bench_under_rocprof.log 2>&1
grep '^{"metric"' $OUT/bench_under_rocprof.log > $OUT/bench_under_rocprof.json
rocprofv3 --kernel-trace --stats --output-format csv -d $OUT/prof_dw -o ${TAG}_dw -- python $R/bench.py --steps 20 --warmup 5 --no-cpu-baseline > $OUT/bench_dw_under_rocprof.log 2>&1
cd $R && bash tools/pmc_passes.sh $TAG/pmc_s300_w60 300 60 > $OUT/pmc_passes_s300_w60.log 2>&1
bash tools/pmc_passes.sh $TAG/pmc_s20_w5 20 5 > $OUT/pmc_passes_s20_w5.log 2>&1
[ "${DIAG:-1}" = 1 ] && bash tools/pmc_diag.sh $TAG/pmcdiag > $OUT/pmc_diag.log 2>&1
head -5 $OUT/prof/${TAG}_kernel_stats.csv
tail -3 $OUT/pmc_passes_s300_w60.log
cut -c1-300 $OUT/bench_default.json
