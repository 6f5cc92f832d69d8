cd $GRAFT_REPO_ROOT
mkdir -p gpurun_out/r05d
timeout 900 python -m pytest tests/test_gpu_e2e.py -m gpu -x -q -k "distinct_windows or same_window" 2>&1 | tail -3 | tee gpurun_out/r05d/tests.txt
DEMFI_BENCH_BACKEND=gloo timeout 900 python3 bench.py --gpus 2 --steps 4 --warmup 1 > gpurun_out/r05d/bench_gloo_2ranks_selflaunch.json 2> gpurun_out/r05d/bench_gloo.err; echo "rc=$?"
tail -c 1500 gpurun_out/r05d/bench_gloo_2ranks_selflaunch.json; tail -3 gpurun_out/r05d/bench_gloo.err
