#!/bin/bash
cd /root/repo
export DEMFI_HIP_LIB=/root/repo/demfi_amd/csrc/libdemfi_hip_abl.so
OPS="Booster_Module.GB.convzr1 Booster_Module.GB.convq1 Booster_Module.GB.convzr2 Booster_Module.GB.convq2"
for v in 0 1 2 3 4; do
  DEMFI_SEP_VARIANT=$v timeout 300 python tools/op_time.py $OPS 2>&1 | grep -v Warning
done > gpurun_out/gru_abl.txt
cat gpurun_out/gru_abl.txt
