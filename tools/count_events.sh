#!/bin/bash
# Dynamic event counts of the tick loop (list walks, CAS rounds, hops, ...) from a -DRS_COUNT build
# (variants/count.so); printed by rs_destroy at the end of a bench run.  Normalise by "vehicle-ticks (plan)".
cp $GRAFT_REPO_ROOT/resco_amd/csrc/libresco_sim.so /tmp/keep.so
cp $GRAFT_REPO_ROOT/variants/${VARIANT:-count}.so $GRAFT_REPO_ROOT/resco_amd/csrc/libresco_sim.so
timeout 300 python $GRAFT_REPO_ROOT/bench.py --steps ${1:-60} --warmup ${2:-100} --no-cpu-baseline 2>&1 | grep "RS_COUNT\|RS_BARWAIT"
cp /tmp/keep.so $GRAFT_REPO_ROOT/resco_amd/csrc/libresco_sim.so
