#!/bin/bash
cd /root/repo
mkdir -p gpurun_out/fin
timeout 1500 python -m pytest tests -q -m gpu 2>&1 | tail -4 > gpurun_out/fin/pytest_gpu.txt; cat gpurun_out/fin/pytest_gpu.txt
timeout 300 python -c "import __graft_entry__ as g; g.smoke()" 2>&1 | tail -3 | tee gpurun_out/fin/smoke.txt
bash tools/profile_bench.sh r03c > gpurun_out/fin/profile_bench.log 2>&1
tail -6 gpurun_out/fin/profile_bench.log
