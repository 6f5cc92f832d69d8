mkdir -p gpurun_out/r3n
timeout 2700 python -m pytest tests -m gpu -q -x > gpurun_out/r3n/tests_gpu.txt 2>&1
tail -6 gpurun_out/r3n/tests_gpu.txt
python bench.py --steps 20 --warmup 5 --no-cpu-baseline --profile-ops gpurun_out/r3n/per_launch.txt > gpurun_out/r3n/bench.json 2> gpurun_out/r3n/bench.err
tail -2 gpurun_out/r3n/bench.err | grep -v amdgpu
python - <<'PY'
import json
d=json.loads(open('gpurun_out/r3n/bench.json').read().strip().splitlines()[-1])
print(d['value'], d['ms_per_step'], d['verified']['mismatching_bytes'], d['roofline']['frac'], d['roofline_hbm'], d['breakdown_ms'])
PY
grep -v " conv " gpurun_out/r3n/per_launch.txt
