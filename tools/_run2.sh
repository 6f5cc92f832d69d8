cd $GRAFT_REPO_ROOT
export TMPDIR=/tmp
mkdir -p gpurun_out
timeout 2400 python -m pytest tests -m gpu -x -q 2>&1 | tail -8 > gpurun_out/r05_gpu_tests.txt
cat gpurun_out/r05_gpu_tests.txt
bash tools/profile_bench.sh r05a > gpurun_out/prof_r05a.log 2>&1
tail -5 gpurun_out/prof_r05a.log
timeout 1500 python tools/pmc_traffic.py gpurun_out/pmc_r05a r05 > gpurun_out/pmc_r05a.log 2>&1
tail -30 gpurun_out/pmc_r05a.log
