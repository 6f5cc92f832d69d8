mkdir -p gpurun_out/r3h
C=$PWD/demfi_amd/csrc
DEMFI_PAIR=3 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -1
T=gpurun_out/r3h/trace.txt
export DEMFI_HIP_LIB=$C/libdemfi_hip_trace.so
DEMFI_PAIR=3 PROBE_DATA=relu python tools/phase_trace.py c3x3 3 2>>gpurun_out/r3h/trace.err >> $T
DEMFI_PAIR=3 PROBE_DATA=relu python tools/phase_trace.py c3x3res 3 2>>gpurun_out/r3h/trace.err >> $T
cat $T; tail -3 gpurun_out/r3h/trace.err
unset DEMFI_HIP_LIB
P=gpurun_out/r3h/probe.txt
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
for pair in 0 3 0 3; do
  DEMFI_PAIR=$pair python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3h/bench_$pair.json 2> gpurun_out/r3h/bench_$pair.err
  echo "bench PAIR=$pair: $(head -c 230 gpurun_out/r3h/bench_$pair.json | cut -c60-230)"
done
