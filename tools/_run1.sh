cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05c
timeout 1200 python -m pytest tests/test_gpu_e2e.py tests/test_gpu_kernels.py -m gpu -x -q 2>&1 | tail -4 | tee gpurun_out/r05c/tests.txt
timeout 900 python -m pytest tests/test_gpu_configs.py -m gpu -x -q -k "batched_equals_module or benched_path" 2>&1 | tail -4 | tee -a gpurun_out/r05c/tests.txt
DEMFI_AB_ROUNDS=2 bash tools/ab_ops.sh gpurun_out/r05c/ab "DEMFI_WARP_TB=0" "DEMFI_WARP_TB=1" "DEMFI_WARP_TB=2" 2>&1 | tee gpurun_out/r05c/ab.txt
