mkdir -p gpurun_out/r3m
C=$PWD/demfi_amd/csrc
P=gpurun_out/r3m/probe.txt
for rep in 1 2; do for lib in prev cur; do
  if [ $lib = prev ]; then export DEMFI_HIP_LIB=$C/libdemfi_hip_prev.so; else export DEMFI_HIP_LIB=$C/libdemfi_hip.so; fi
  echo "== $lib" >> $P
  PROBE_DATA=relu python tools/conv_probe.py gru 30 2>/dev/null >> $P
  python tools/conv_probe.py narrow 30 2>/dev/null >> $P
  python bench.py --steps 20 --warmup 5 --no-cpu-baseline > gpurun_out/r3m/bench_$lib.json 2> gpurun_out/r3m/bench_$lib.err
  python -c "
import json
d=json.loads(open('gpurun_out/r3m/bench_$lib.json').read().strip().splitlines()[-1])
print('$lib', d['value'], d['ms_per_step'], d['breakdown_ms']['per_t'], d['breakdown_ms']['trunk_once_per_window'])" >> $P
done; done
cat $P
