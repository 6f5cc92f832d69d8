mkdir -p gpurun_out/r3e
C=$PWD/demfi_amd/csrc
DEMFI_PAIR=3 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > gpurun_out/r3e/tests_stg.txt 2>&1
tail -2 gpurun_out/r3e/tests_stg.txt
P=gpurun_out/r3e/probe.txt
for lib in slp noslp; do
  if [ $lib = slp ]; then export DEMFI_HIP_LIB=$C/libdemfi_hip_slp.so; else export DEMFI_HIP_LIB=$C/libdemfi_hip.so; fi
  for data in relu zero; do for pair in 0 3; do
    echo "== lib=$lib PAIR=$pair DATA=$data" >> $P
    DEMFI_PAIR=$pair PROBE_DATA=$data python tools/conv_probe.py c3x3 40 2>/dev/null >> $P
    DEMFI_PAIR=$pair PROBE_DATA=$data python tools/conv_probe.py c3x3res 40 2>/dev/null >> $P
  done; done
  echo "== lib=$lib gru narrow" >> $P
  PROBE_DATA=relu python tools/conv_probe.py gru 20 2>/dev/null >> $P
  python tools/conv_probe.py narrow 20 2>/dev/null >> $P
  PROBE_DATA=relu python tools/conv_probe.py c7x7 10 2>/dev/null >> $P
done
cat $P
T=gpurun_out/r3e/trace.txt
export DEMFI_HIP_LIB=$C/libdemfi_hip_trace.so
DEMFI_PAIR=3 PROBE_DATA=relu python tools/phase_trace.py c3x3 3 2>>gpurun_out/r3e/trace.err >> $T
DEMFI_PAIR=3 PROBE_DATA=relu python tools/phase_trace.py c3x3res 3 2>>gpurun_out/r3e/trace.err >> $T
DEMFI_PAIR=0 PROBE_DATA=relu python tools/phase_trace.py c3x3 3 2>>gpurun_out/r3e/trace.err >> $T
cat $T
for cfg in "slp 0" "noslp 0" "noslp 3" "slp 3"; do set -- $cfg
  if [ $1 = slp ]; then export DEMFI_HIP_LIB=$C/libdemfi_hip_slp.so; else export DEMFI_HIP_LIB=$C/libdemfi_hip.so; fi
  DEMFI_PAIR=$2 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3e/bench_$1_$2.json 2> gpurun_out/r3e/bench_$1_$2.err
  echo "bench $1 PAIR=$2: $(head -c 230 gpurun_out/r3e/bench_$1_$2.json | cut -c60-230)"
done
