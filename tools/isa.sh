#!/bin/bash
# study build of ONE capacity (64- and 80-VGPR step kernels) with the ISA kept: tools/isa.sh <cap> [extra hipcc flags] -> /tmp/isa/
cap=$1; shift
mkdir -p /tmp/isa && cd /tmp/isa && rm -f resco_sim-hip-*
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mllvm -disable-machine-licm -fPIC -shared -DRS_ONE_CAP=$cap "$@" \
  -I/root/repo/include -I/root/repo/resco_amd/csrc /root/repo/resco_amd/csrc/resco_sim.hip -o /tmp/isa/one.so -save-temps -Rpass-analysis=kernel-resource-usage 2>&1 |
grep -A10 "Name: _Z1.rs_step" | sed 's/.*remark: [^ ]* *//; s/\[-Rpass.*//' | grep "Name\|SGPRs\|VGPRs\|Scratch" | tr '\n' ' ' | sed 's/Function Name/\nkernel/g'; echo
