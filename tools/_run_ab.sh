#!/bin/bash
cd /root/repo
OPS="Booster_Module.GB.convzr1 Booster_Module.GB.convq1 Booster_Module.GB.convzr2 Booster_Module.GB.convq2"
unset DEMFI_HIP_LIB
timeout 300 python -m pytest tests/test_gpu_kernels.py -x -q -m gpu -k "sep_gru" 2>&1 | tail -2
for r in 1 2; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export DEMFI_HIP_LIB=/root/repo/demfi_amd/csrc/libdemfi_hip_prev.so; else unset DEMFI_HIP_LIB; fi
    echo "== $lib"
    timeout 300 python tools/op_time.py $OPS 2>&1 | grep -v Warning | grep -v amdgpu.ids
    timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d.get('verified',{}).get('mismatching_bytes'))"
  done
done 2>&1 | tee gpurun_out/ab.txt
