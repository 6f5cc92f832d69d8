set -x
mkdir -p gpurun_out/r3a
./tools/microbench/overlap_matrix > gpurun_out/r3a/overlap.txt 2>&1
for pair in 0 1 2; do
  for data in relu zero rand; do
    echo "== PAIR=$pair DATA=$data" >> gpurun_out/r3a/probe.txt
    DEMFI_PAIR=$pair PROBE_DATA=$data python tools/conv_probe.py c3x3 40 >> gpurun_out/r3a/probe.txt 2>&1
    DEMFI_PAIR=$pair PROBE_DATA=$data python tools/conv_probe.py c3x3res 40 >> gpurun_out/r3a/probe.txt 2>&1
  done
  echo "== PAIR=$pair batch 21 relu" >> gpurun_out/r3a/probe.txt
  DEMFI_PAIR=$pair PROBE_B=21 PROBE_DATA=relu python tools/conv_probe.py c3x3 10 >> gpurun_out/r3a/probe.txt 2>&1
  DEMFI_PAIR=$pair PROBE_B=21 PROBE_DATA=relu python tools/conv_probe.py c3x3res 10 >> gpurun_out/r3a/probe.txt 2>&1
done
DEMFI_PAIR=1 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > gpurun_out/r3a/tests_pair1.txt 2>&1
DEMFI_PAIR=2 timeout 900 python -m pytest tests/test_gpu_kernels.py -x -q -k "conv" > gpurun_out/r3a/tests_pair2.txt 2>&1
for pair in 0 1; do
  DEMFI_PAIR=$pair python bench.py --steps 10 --warmup 3 > gpurun_out/r3a/bench_pair$pair.json 2> gpurun_out/r3a/bench_pair$pair.err
done
tail -3 gpurun_out/r3a/tests_pair1.txt gpurun_out/r3a/tests_pair2.txt
cat gpurun_out/r3a/overlap.txt
cat gpurun_out/r3a/probe.txt | grep -v "^+" 
head -c 600 gpurun_out/r3a/bench_pair0.json; echo; head -c 600 gpurun_out/r3a/bench_pair1.json
