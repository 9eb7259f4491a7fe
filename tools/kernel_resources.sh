#!/bin/bash
# register / scratch usage of every kernel, as hipcc reports it (no GPU needed).  Extra arguments go to hipcc (-DCELL_LEN=16.0f ...)
cd "$(dirname "$0")/.." || exit 1
/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -fno-honor-nans -mllvm -disable-machine-licm -fPIC -shared -Iinclude -Iresco_amd/csrc "$@" \
    resco_amd/csrc/resco_sim.hip -o /tmp/rs_resources.so -Rpass-analysis=kernel-resource-usage 2>&1 |
awk '/remark: Function Name:/ {name=$5}
     /remark:     VGPRs:/ {v=$4} /remark:     TotalSGPRs:/ {s=$4} /ScratchSize/ {sc=$5} /Occupancy/ {o=$5}
     /SGPRs Spill/ {ss=$5} /VGPRs Spill/ {vs=$5}
     /LDS Size/ {print name, "VGPR", v, "SGPR", s, "scratch_B_per_lane", sc, "waves_per_SIMD", o, "sgpr_spill", ss, "vgpr_spill", vs}' | c++filt
