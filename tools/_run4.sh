mkdir -p gpurun_out/r3d
DEMFI_PAIR=3 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > gpurun_out/r3d/tests_stg.txt 2>&1
tail -3 gpurun_out/r3d/tests_stg.txt
P=gpurun_out/r3d/probe.txt
for data in relu zero; do for pair in 0 3; do
  echo "== PAIR=$pair DATA=$data" >> $P
  DEMFI_PAIR=$pair PROBE_DATA=$data python tools/conv_probe.py c3x3 40 2>/dev/null >> $P
  DEMFI_PAIR=$pair PROBE_DATA=$data python tools/conv_probe.py c3x3res 40 2>/dev/null >> $P
done; done
for pair in 0 3; do
  echo "== PAIR=$pair batch 21 relu" >> $P
  DEMFI_PAIR=$pair PROBE_B=21 PROBE_DATA=relu python tools/conv_probe.py c3x3 10 2>/dev/null >> $P
  DEMFI_PAIR=$pair PROBE_B=21 PROBE_DATA=relu python tools/conv_probe.py c3x3res 10 2>/dev/null >> $P
done
cat $P
T=gpurun_out/r3d/trace.txt
export DEMFI_HIP_LIB=$PWD/demfi_amd/csrc/libdemfi_hip_trace.so
for data in relu; do
  DEMFI_PAIR=3 PROBE_DATA=$data python tools/phase_trace.py c3x3 3 2>>gpurun_out/r3d/trace.err >> $T
  DEMFI_PAIR=3 PROBE_DATA=$data python tools/phase_trace.py c3x3res 3 2>>gpurun_out/r3d/trace.err >> $T
done
unset DEMFI_HIP_LIB
cat $T
DEMFI_PAIR=3 python bench.py --steps 10 --warmup 3 > gpurun_out/r3d/bench_stg.json 2> gpurun_out/r3d/bench_stg.err
head -c 400 gpurun_out/r3d/bench_stg.json; tail -2 gpurun_out/r3d/bench_stg.err
