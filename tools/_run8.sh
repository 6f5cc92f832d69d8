mkdir -p gpurun_out/r3j
timeout 2400 python -m pytest tests -m gpu -x -q > gpurun_out/r3j/tests_gpu.txt 2>&1
tail -5 gpurun_out/r3j/tests_gpu.txt
python bench.py --steps 20 --warmup 5 --profile-ops gpurun_out/r3j/per_launch.txt > gpurun_out/r3j/bench.json 2> gpurun_out/r3j/bench.err
head -c 3000 gpurun_out/r3j/bench.json; tail -3 gpurun_out/r3j/bench.err
python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3
