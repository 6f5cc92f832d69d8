mkdir -p gpurun_out/r3p
bash tools/power_trace.sh $PWD/gpurun_out/r3p/power_trace.txt > gpurun_out/r3p/power_summary.txt 2>&1
head -8 gpurun_out/r3p/power_trace.txt; cat gpurun_out/r3p/power_summary.txt; grep -c . gpurun_out/r3p/power_trace.txt
timeout 1800 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_configs.py -q -x 2>&1 | tail -4
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-ops gpurun_out/r3p/per_launch.txt > gpurun_out/r3p/bench.json 2> gpurun_out/r3p/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3p/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['verified']['mismatching_bytes'], d['roofline']['frac'], d['roofline_hbm']['frac'], d['breakdown_ms'])
PY
grep -E "dec3|enc1|cfr|warp|Dec_last2 " gpurun_out/r3p/per_launch.txt
