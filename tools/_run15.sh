mkdir -p gpurun_out/r3v
export DEMFI_HIP_LIB=$PWD/demfi_amd/csrc/libdemfi_hip_trace.so
for op in Booster_Module.GB.convzr1 Booster_Module.GB.convq1 Booster_Module.GB.convzr2 Booster_Module.GB.convq2; do
  python tools/phase_trace.py "op:$op" 2>gpurun_out/r3v/trace.err >> gpurun_out/r3v/trace.txt
done
cat gpurun_out/r3v/trace.txt; grep -v amdgpu gpurun_out/r3v/trace.err | tail -3
