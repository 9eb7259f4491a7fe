#!/bin/bash
# The host ceiling of EIGHT ranks without an 8-GPU node: `torch.distributed.run --nproc-per-node 8 bench.py --gpus 8`
# with all ranks on device 0 (RESCO_BENCH_DEVICE=0), a gloo rendezvous, and so few environments per rank that the device is not the
# limit.  The eight ranks still SHARE one device, so their measured step rate is the device's; what the run isolates is bench.py's
# `host.issue_us_per_step`: the wall time one rank's host thread needs to hand a step to the runtime while seven others do the same on
# the box's cores (cgroup quota included) -- 1 / it is the step rate the host side can sustain per GPU.  Compared with what each BASELINE config needs per
# rank at the single-GPU rates of this build.
#   bash tools/host_ceiling.sh > profiles/r05_host_ceiling.txt        (on the GPU box)
cd "$(dirname "$0")/.." || exit 1
export RESCO_BENCH_DEVICE=0 RESCO_BENCH_BACKEND=gloo
run() {  # label, ranks, map, envs per rank, pipes, extra env
  local label="$1" n="$2" map="$3" envs="$4" pipes="$5"; shift 5
  local line
  line=$(env "$@" python -m torch.distributed.run --nnodes=1 --nproc-per-node "$n" --master-addr 127.0.0.1 --master-port $((29500 + RANDOM % 1000)) \
         bench.py --gpus "$n" --steps 200 --warmup 20 --map "$map" --envs "$envs" --pipes "$pipes" --no-cpu-baseline 2>/dev/null | grep '^{"metric"' | tail -1)
  python - "$label" "$n" "$envs" "$pipes" "$line" <<'PY'
import json, sys
label, n, envs, pipes, line = sys.argv[1], int(sys.argv[2]), int(sys.argv[3]), int(sys.argv[4]), sys.argv[5]
r = json.loads(line)
h = r['host']['issue_us_per_step']
print('%-62s ranks %d x %4d envs x %d pipes: host issues a step in %6.1f us = %6.0f steps/s per rank; measured %5.0f steps/s (device shared by the ranks); numa %s' % (
    label, n, envs, pipes, h, 1e6 / h, 1e3 / r['ms_per_step'], r['config']['rank0_numa_node']))
PY
}
echo "# host: $(nproc) cores visible, cgroup cpu.max: $(cat /sys/fs/cgroup/cpu.max 2>/dev/null)"
echo "# needed per rank at this build's single-GPU rates: config 3 (ingolstadt21 x 4096, 2 pipes) ~690 steps/s; config 4 (cologne8 x 2048) ~3 700; config 5 (ingolstadt21 x 1024 + policy, 8 pipes) ~2 200"
for map in ingolstadt21 cologne8; do
  run "$map: 1 rank, rs_group_step" 1 $map 64 2
  run "$map: 8 ranks, rs_group_step, NUMA binding" 8 $map 64 2
  run "$map: 8 ranks, rs_group_step, no binding" 8 $map 64 2 RESCO_BENCH_NO_NUMA=1
  run "$map: 8 ranks, two calls per pipe and step (round 4)" 8 $map 64 2 RESCO_BENCH_PER_PIPE_CALLS=1
  run "$map: 8 ranks, rs_group_step, 8 pipes of 8" 8 $map 64 8
  run "$map: 8 ranks, two calls per pipe and step, 8 pipes of 8" 8 $map 64 8 RESCO_BENCH_PER_PIPE_CALLS=1
done
