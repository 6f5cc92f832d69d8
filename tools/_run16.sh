mkdir -p gpurun_out/r3w
timeout 1500 python -m pytest tests/test_gpu_kernels.py tests/test_gpu_e2e.py tests/test_gpu_cabi.py -q -x 2>&1 | tail -3
for i in 1 2; do
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-ops gpurun_out/r3w/per_launch.txt > gpurun_out/r3w/bench.json 2> gpurun_out/r3w/bench.err
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3w/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['verified']['mismatching_bytes'], d['roofline']['frac'], d['breakdown_ms'])
PY
done
grep -E "Mixer|flow_occ|Dec_first_2" gpurun_out/r3w/per_launch.txt | head -14
