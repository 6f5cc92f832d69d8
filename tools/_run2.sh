mkdir -p gpurun_out/r3b
./tools/microbench/overlap_matrix > gpurun_out/r3b/overlap.txt 2>&1
P=gpurun_out/r3b/probe.txt
run() { echo "== $*" >> $P; env "$@" python tools/conv_probe.py c3x3 40 2>/dev/null >> $P; env "$@" python tools/conv_probe.py c3x3res 40 2>/dev/null >> $P; }
for data in relu zero; do
  run PROBE_DATA=$data DEMFI_PAIR=0 DEMFI_KNOB=0
  run PROBE_DATA=$data DEMFI_PAIR=0 DEMFI_KNOB=1
  run PROBE_DATA=$data DEMFI_PAIR=1 DEMFI_KNOB=0
  run PROBE_DATA=$data DEMFI_PAIR=1 DEMFI_KNOB=1
  run PROBE_DATA=$data DEMFI_PAIR=1 DEMFI_KNOB=2
  run PROBE_DATA=$data DEMFI_PAIR=1 DEMFI_KNOB=3
  run PROBE_DATA=$data DEMFI_PAIR=1 DEMFI_KNOB=4
  run PROBE_DATA=$data DEMFI_PAIR=1 DEMFI_KNOB=5
  run PROBE_DATA=$data DEMFI_PAIR=2 DEMFI_KNOB=1
done
echo "== gru/narrow knob 0" >> $P; DEMFI_KNOB=0 PROBE_DATA=relu python tools/conv_probe.py gru 20 2>/dev/null >> $P; DEMFI_KNOB=0 python tools/conv_probe.py narrow 20 2>/dev/null >> $P
echo "== gru/narrow knob 1" >> $P; DEMFI_KNOB=1 PROBE_DATA=relu python tools/conv_probe.py gru 20 2>/dev/null >> $P; DEMFI_KNOB=1 python tools/conv_probe.py narrow 20 2>/dev/null >> $P
DEMFI_KNOB=1 python bench.py --steps 10 --warmup 3 > gpurun_out/r3b/bench_knob1.json 2> gpurun_out/r3b/bench_knob1.err
DEMFI_KNOB=0 python bench.py --steps 10 --warmup 3 > gpurun_out/r3b/bench_knob0.json 2> gpurun_out/r3b/bench_knob0.err
cat gpurun_out/r3b/overlap.txt; cat $P; head -c 300 gpurun_out/r3b/bench_knob0.json; echo; head -c 300 gpurun_out/r3b/bench_knob1.json
