cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05a
timeout 900 python -m pytest tests/test_gpu_kernels.py -m gpu -q -k "resblock" 2>&1 | tail -15 > gpurun_out/r05a/test_resblock.txt
cat gpurun_out/r05a/test_resblock.txt
DEMFI_HIP_LIB=$PWD/demfi_amd/csrc/libdemfi_hip_trace.so PROBE_B=21 timeout 300 python tools/rb_trace.py 2>&1 | grep -v amdgpu.ids | tee gpurun_out/r05a/rb_trace_b21.txt
timeout 1200 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q 2>&1 | tail -5 | tee gpurun_out/r05a/test_e2e.txt
DEMFI_AB_ROUNDS=2 bash tools/ab_ops.sh gpurun_out/r05a/ab "DEMFI_RESBLOCK=0" "" 2>&1 | tee gpurun_out/r05a/ab.txt
