mkdir -p gpurun_out/r3c
T=gpurun_out/r3c/trace.txt
export DEMFI_HIP_LIB=$PWD/demfi_amd/csrc/libdemfi_hip_trace.so
for data in zero relu; do
  for pair in 0 1 2; do
    DEMFI_PAIR=$pair PROBE_DATA=$data python tools/phase_trace.py c3x3 3 2>>gpurun_out/r3c/trace.err >> $T
    DEMFI_PAIR=$pair PROBE_DATA=$data python tools/phase_trace.py c3x3res 3 2>>gpurun_out/r3c/trace.err >> $T
  done
done
DEMFI_PAIR=0 PROBE_DATA=relu python tools/phase_trace.py c3x3 21 2>>gpurun_out/r3c/trace.err >> $T
cat $T; grep -v amdgpu.ids gpurun_out/r3c/trace.err | tail -5
