mkdir -p gpurun_out/r3i
C=$PWD/demfi_amd/csrc
DEMFI_PAIR=3 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" 2>&1 | tail -1
T=gpurun_out/r3i/trace.txt
export DEMFI_HIP_LIB=$C/libdemfi_hip_trace.so
for knob in 0 1; do
echo "--- KNOB=$knob" >> $T
DEMFI_KNOB=$knob DEMFI_PAIR=3 PROBE_DATA=relu python tools/phase_trace.py c3x3 3 2>>gpurun_out/r3i/trace.err >> $T
DEMFI_KNOB=$knob DEMFI_PAIR=3 PROBE_DATA=relu python tools/phase_trace.py c3x3res 3 2>>gpurun_out/r3i/trace.err >> $T
done
cat $T; tail -3 gpurun_out/r3i/trace.err
unset DEMFI_HIP_LIB
P=gpurun_out/r3i/probe.txt
for cfg in "0 0" "3 0" "3 1"; do set -- $cfg
  echo "== PAIR=$1 KNOB=$2 batch 21 relu" >> $P
  DEMFI_KNOB=$2 DEMFI_PAIR=$1 PROBE_B=21 PROBE_DATA=relu python tools/conv_probe.py c3x3 10 2>/dev/null >> $P
  DEMFI_KNOB=$2 DEMFI_PAIR=$1 PROBE_B=21 PROBE_DATA=relu python tools/conv_probe.py c3x3res 10 2>/dev/null >> $P
done
cat $P
for cfg in "0 0" "3 0" "3 1"; do set -- $cfg
  DEMFI_KNOB=$2 DEMFI_PAIR=$1 python bench.py --steps 10 --warmup 3 --no-cpu-baseline > gpurun_out/r3i/bench_$1_$2.json 2> gpurun_out/r3i/bench_$1_$2.err
  echo "bench PAIR=$1 KNOB=$2: $(head -c 230 gpurun_out/r3i/bench_$1_$2.json | cut -c60-230)"
done
