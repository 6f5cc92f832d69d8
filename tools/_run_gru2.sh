#!/bin/bash
cd /root/repo
timeout 600 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py -x -q -m gpu 2>&1 | tail -5
OPS="Booster_Module.GB.convzr1 Booster_Module.GB.convq1 Booster_Module.GB.convzr2 Booster_Module.GB.convq2"
timeout 300 python tools/op_time.py $OPS 2>&1 | grep -v Warning | tee gpurun_out/gru2.txt
timeout 300 python bench.py 2>&1 | tail -1 | tee gpurun_out/gru2_bench.json | python -c "import json,sys; d=json.loads(sys.stdin.read()); print(d['value'], d['ms_per_step'], d.get('verified'))"
