mkdir -p gpurun_out/r06h
(
echo "== head (cap 896), profiling kernel"; RESCO_SIM_LIB=variants/head.so python tools/phase_profile.py ingolstadt21 4096 0
echo "== dense (cap 832), profiling kernel"; RS_CAPACITY=832 python tools/phase_profile.py ingolstadt21 4096 0
echo "== head, one WG per CU (256 envs)"; RESCO_SIM_LIB=variants/head.so python tools/phase_profile.py ingolstadt21 256 0
echo "== dense, one WG per CU (256 envs)"; RS_CAPACITY=832 python tools/phase_profile.py ingolstadt21 256 0
) 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r06h/phase_profiles.txt
