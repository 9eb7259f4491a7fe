#!/bin/bash
# usage: ab.sh a.so b.so ...  : alternate the variants, 2 bench runs each
for r in 1 2; do for v in "$@"; do cp variants/$v resco_amd/csrc/libresco_sim.so; echo -n "$v "; python bench.py --no-cpu-baseline 2>&1 | tail -1 | cut -c27-45; done; done
