#!/bin/bash
cd /root/repo
timeout 400 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "conv" 2>&1 | tail -2
for v in "DEMFI_C64_STG=1" "DEMFI_DACC_HELP=0" "DEMFI_DACC_HELP=1"; do
  env $v timeout 300 python tools/op_time.py Decoder_res.0.conv1 Dec_first Decoder_res_2.0.conv1 2>&1 | grep -v Warning | grep -v amdgpu.ids
done | tee gpurun_out/knob.txt
