cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05b
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "resblock" 2>&1 | tail -5 | tee gpurun_out/r05b/test_resblock.txt
DEMFI_HIP_LIB=$PWD/demfi_amd/csrc/libdemfi_hip_trace.so PROBE_B=21 timeout 300 python tools/rb_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05b/rb_trace_b21.txt
PROBE_B=21 PROBE_DATA=relu timeout 300 python tools/conv_probe.py resblock 20 2>&1 | tail -2 | tee gpurun_out/r05b/probe_b21.txt
DEMFI_AB_ROUNDS=1 bash tools/ab_ops.sh gpurun_out/r05b/ab "DEMFI_RESBLOCK=0" "" 2>&1 | tee gpurun_out/r05b/ab.txt
