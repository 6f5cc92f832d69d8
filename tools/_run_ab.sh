#!/bin/bash
cd /root/repo
OPS="Refine_Module.enc1#t Refine_Module.enc2 Refine_Module.dec1 Refine_Module.dec2 FF_RDB_Module.RDBs.0.convs.0.conv.0 FF_RDB_Module.RDBs.0.convs.3.conv.0 FF_RDB_Module.RDBs.0.LFF FF_RDB_Module.GFF.0 FF_RDB_Module.UPNet.2 FAC_FB_Module.shared_FGAC.w_gen Refine_Module.enc1#aF"
unset DEMFI_HIP_LIB
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -2
for r in 1 2; do
  for lib in prev cur; do
    if [ $lib = prev ]; then export DEMFI_HIP_LIB=/root/repo/demfi_amd/csrc/libdemfi_hip_prev.so; else unset DEMFI_HIP_LIB; fi
    echo "== $lib"
    [ $r = 1 ] && timeout 300 python tools/op_time.py $OPS 2>&1 | grep -v Warning | grep -v amdgpu.ids
    timeout 300 python bench.py --no-cpu-baseline 2>&1 | tail -1 | python -c "import json,sys; d=json.loads(sys.stdin.read()); print('bench', d['value'], d['ms_per_step'], d.get('verified',{}).get('mismatching_bytes'), d['breakdown_ms']['trunk_once_per_window'])"
  done
done 2>&1 | tee gpurun_out/ab.txt
