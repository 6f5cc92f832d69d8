#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5
OPS="Decoder_res.0.conv1 Decoder_res.0.conv2 Dec_first Decoder_res_2.0.conv1 Decoder_res_2.0.conv2"
for r in 1 2; do
  for v in 1 0; do
    echo "== DEMFI_C64_STG=$v"
    DEMFI_C64_STG=$v timeout 300 python tools/op_time.py $OPS 2>&1 | grep -v Warning | grep -v amdgpu.ids
    DEMFI_C64_STG=$v timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d.get('verified',{}).get('mismatching_bytes'))"
  done
done 2>&1 | tee gpurun_out/dacc.txt
