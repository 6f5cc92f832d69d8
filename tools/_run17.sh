mkdir -p gpurun_out/r3x
# 8 gloo ranks sharing the one GPU of this box at 256x256: CPU-side soak of the N > 1 bench path (broadcast, barriers, max-over-ranks)
DEMFI_BENCH_BACKEND=gloo timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 8 --master-addr 127.0.0.1 --master-port 29541 \
  bench.py --gpus 8 --height 256 --width 256 --steps 6 --warmup 2 --no-cpu-baseline > gpurun_out/r3x/bench_gloo8_256.json 2> gpurun_out/r3x/bench_gloo8_256.err
tail -c 700 gpurun_out/r3x/bench_gloo8_256.json; echo; grep -v amdgpu gpurun_out/r3x/bench_gloo8_256.err | tail -3
# 2 gloo ranks at the headline size (as in round 2)
DEMFI_BENCH_BACKEND=gloo DEMFI_NTRUNK=2 timeout 900 python -m torch.distributed.run --nnodes=1 --nproc-per-node 2 --master-addr 127.0.0.1 --master-port 29542 \
  bench.py --gpus 2 --steps 8 --warmup 2 --no-cpu-baseline > gpurun_out/r3x/bench_gloo2_720p.json 2> gpurun_out/r3x/bench_gloo2_720p.err
head -c 400 gpurun_out/r3x/bench_gloo2_720p.json; echo
python tools/pmc_traffic.py $PWD/gpurun_out/r3x/pmc > gpurun_out/r3x/pmc_summary.txt 2>&1
cp gpurun_out/r3x/pmc/r03_pmc_traffic.json gpurun_out/r3x/ 2>/dev/null; rm -rf gpurun_out/r3x/pmc
tail -22 gpurun_out/r3x/pmc_summary.txt
bash tools/profile_bench.sh r03b > gpurun_out/r3x/profile_bench.log 2>&1
tail -6 gpurun_out/r3x/profile_bench.log
