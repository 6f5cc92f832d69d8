#!/bin/bash
cd /root/repo
export DEMFI_HIP_LIB=/root/repo/demfi_amd/csrc/libdemfi_hip_trace.so PROBE_DATA=relu PROBE_KERNEL=dacc
for k in 0 8 16 24; do
  echo "--- KNOB=$k"
  for c in c3x3 c3x3res; do DEMFI_KNOB=$k timeout 200 python tools/phase_trace.py $c 3 2>&1 | grep -v Warn | grep -v amdgpu.ids | grep -v "wave [123]"; done
done | tee gpurun_out/tr_dacc.txt
unset DEMFI_HIP_LIB
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -3
